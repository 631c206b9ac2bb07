// Block-dense aggregation on the fp32 matrix cores, operands straight from memory (round 4).
//
// A cluster batch's adjacency is block-dense (spmm_mfma.hip: 99 % of a batch row's neighbours lie in the row's
// own METIS part, a part's diagonal block is ~60 % full), so per part Y_p = C_p . X_p + (the few neighbours in
// other parts), C_p = the block's edge counts.  spmm_mfma.hip runs that product on the bf16 matrix cores with X
// cut into three bf16 pieces through LDS: the right trade at D >= 2048, but below it the kernel is its set-up,
// barriers and conversions (13-21 us for 2-8 MB), and the LDS-gather kernel (spmm_csr_lds2_kernel) is start-up
// bound as well (12-19 us per call, 25-35 % of a narrow step).
//
// Here the same product runs on v_mfma_f32_16x16x4_f32 with NO LDS, NO conversion and NO barrier: the counts
// come from the batch's prepared block structure (gist_spmm_blocks_prepare: bf16 counts <= 256, exact; widened to
// fp32 by a shift), X's rows are read as they lie in memory (4-byte loads, 64-byte segments: no alignment or
// width restriction, so the F = 602 input layer takes this path too), every count x value product is an exact
// fp32 FMA step -- the same arithmetic as adding the neighbours one by one, in another fixed order.  One WAVE owns
// (block, 16-column tile, group of 16-row tiles); the k permutation trick of classlayer.hip gives a lane its four
// k steps of a 16-k block from ONE 8-byte load of the counts image.  Neighbours outside the block (<= 8 per row,
// listed by the prepare kernel) are added per lane afterwards; a row the prepare kernel took out of the dense
// product (more than 8 outside neighbours, a count > 256) and the rows of a block beyond its first 128 are gathered
// in full.  Masks of the fused dropout (SpmmDrop modes 1 and 2) are applied to what is stored / read.
#include "common.h"

namespace gist {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int DB_ROWS = 128;                    // = MF_ROWS of spmm_mfma.hip (the prepared image's row count)
#ifndef MF_REM_N
#define MF_REM_N 8
#endif
constexpr int DB_REM = MF_REM_N;                // = MF_REM
constexpr int DB_PREP_STRIDE = 16 * DB_ROWS * 16 + DB_ROWS * 4 + DB_ROWS * DB_REM * 4 + 16;      // = MF_PREP_STRIDE

struct D32Args {
    const int32_t *rowptr, *col;
    const float *x; int64_t ldx;
    float *y; int64_t ldy;
    int n_rows, d;
    const float *out_scale, *src_scale;
    int accumulate;
    const int32_t *row_blocks;
    int n_blocks, n_col_tiles, groups, rt_per_group;
    const unsigned char *prep;
    SpmmDrop dr;
};

template <int MODE>
__device__ __forceinline__ float d32_src(const D32Args &a, int g_row, int col) {      // scaled, masked x[g_row][col]
    float v = a.x[(int64_t)g_row * a.ldx + col];
    if constexpr (MODE == 2)      // (the mask first, then the scale: as gist_dropout_f32 in front of the plain call)
        v *= drop_keep(a.dr.src_base + (uint64_t)g_row * (uint64_t)a.dr.ld + (uint64_t)col, a.dr.sm, a.dr.p, a.dr.scale);
    if (a.src_scale) v *= a.src_scale[g_row];
    return v;
}

// y[g_row][col] <- v (scaled by out_scale, masks, accumulation as the mode says)
template <int MODE>
__device__ __forceinline__ void d32_store(const D32Args &a, int g_row, int col, float v) {
    if (a.out_scale) v *= a.out_scale[g_row];
    float *o = a.y + (int64_t)g_row * a.ldy + col;
    const uint64_t yi = a.dr.y_base + (uint64_t)g_row * (uint64_t)a.dr.ld + (uint64_t)col;
    if constexpr (MODE == 1) v *= drop_keep(yi, a.dr.sm, a.dr.p, a.dr.scale);
    if (a.accumulate) {
        float old = *o;
        // (__fmul_rn: no contraction into an FMA with the add -- the separate dropout pass rounds the product)
        if constexpr (MODE == 2) old = __fmul_rn(old, drop_keep(yi, a.dr.sm, a.dr.p, a.dr.scale));
        v += old;
    }
    *o = v;
}

// every neighbour of one row, CSR order, four loads in flight (rows outside the dense product)
template <int MODE>
__device__ __forceinline__ float d32_gather_row(const D32Args &a, int g_row, int col) {
    const int e0 = a.rowptr[g_row], e1 = a.rowptr[g_row + 1];
    float v = 0.f;
    int e = e0;
    for (; e + 4 <= e1; e += 4) {
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = d32_src<MODE>(a, a.col[e + u], col);
#pragma unroll
        for (int u = 0; u < 4; ++u) v += t[u];
    }
    for (; e < e1; ++e) v += d32_src<MODE>(a, a.col[e], col);
    return v;
}

template <int MODE, int RT>
__global__ __launch_bounds__(256) void spmm_dense32_kernel(D32Args a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    // wave -> (block, row-tile group, column tile): the four waves of a workgroup take adjacent column tiles
    const int task = blockIdx.x * 4 + wave;
    const int per_block = a.groups * a.n_col_tiles;
    if (task >= a.n_blocks * per_block) return;
    const int rbk = task / per_block;
    const int rem_t = task - rbk * per_block;
    const int grp = rem_t / a.n_col_tiles;
    const int ct = rem_t - grp * a.n_col_tiles;
    int r0, r1;
    if (a.row_blocks) { r0 = a.row_blocks[rbk]; r1 = a.row_blocks[rbk + 1]; }
    else { r0 = rbk * DB_ROWS; r1 = r0 + DB_ROWS; }
    r1 = min(r1, a.n_rows);
    const int nrow = r1 - r0;
    if (nrow <= 0) return;
    const int nloc = min(nrow, DB_ROWS);
    const int n0 = ct * 16;
    const int col = n0 + r;
    const bool col_ok = col < a.d;
    const int colc = min(col, a.d - 1);
    const unsigned char *pb = a.prep + (int64_t)rbk * DB_PREP_STRIDE;
    const int32_t *rem_cnt = reinterpret_cast<const int32_t *>(pb + 16 * DB_ROWS * 16);
    const int32_t *rem_col = rem_cnt + DB_ROWS;
    const int n_rt = (nloc + 15) >> 4;                       // row tiles of the block = its 16-k blocks
    const int rt0 = grp * a.rt_per_group;
    const int rt1 = min(rt0 + a.rt_per_group, n_rt);
    if (rt0 < rt1) {
        f32x4 acc[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // counts of (row rt * 16 + r, k = 16 kb + 4 q + s): 8 bytes at [(2 kb + (q >> 1)) * 128 + row] * 16 + 8 (q & 1)
        const unsigned char *arow = pb + (int64_t)((q >> 1) * DB_ROWS + rt0 * 16 + r) * 16 + 8 * (q & 1);
        // A block has at most eight 16-k blocks: ALL operand loads of the task are issued before the first MFMA, in
        // two batches of four k blocks (the second lands under the first batch's MFMAs) -- a loop that loads one k
        // block, waits and multiplies is one memory latency per block, 5-8 us per task.
        constexpr int KB = 4;
        float bv[2][KB][4];
        uint2 cw[2][KB][RT];
        auto load = [&](float (&bb)[KB][4], uint2 (&cc)[KB][RT], int kb0) {
#pragma unroll
            for (int u = 0; u < KB; ++u) {
                const bool on = kb0 + u < n_rt;                          // wave-uniform
                const int kb = on ? kb0 + u : n_rt - 1;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int k = min(kb * 16 + 4 * q + s, nloc - 1);       // (rows >= nloc have zero counts)
                    bb[u][s] = d32_src<MODE>(a, r0 + k, colc);
                }
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const int tt = min(t, rt1 - rt0 - 1);                   // (a short last group re-reads its last tile)
                    const uint2 v = *reinterpret_cast<const uint2 *>(arow + (int64_t)(2 * kb * DB_ROWS + tt * 16) * 16);
                    cc[u][t] = on ? v : make_uint2(0u, 0u);
                }
            }
        };
        auto compute = [&](const float (&bb)[KB][4], const uint2 (&cc)[KB][RT]) {
#pragma unroll
            for (int u = 0; u < KB; ++u) {
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const float c0 = __uint_as_float(cc[u][t].x << 16), c1 = __uint_as_float(cc[u][t].x & 0xffff0000u);
                    const float c2 = __uint_as_float(cc[u][t].y << 16), c3 = __uint_as_float(cc[u][t].y & 0xffff0000u);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0, bb[u][0], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1, bb[u][1], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(c2, bb[u][2], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(c3, bb[u][3], acc[t], 0, 0, 0);
                }
            }
        };
        const bool two = n_rt > KB;
        load(bv[0], cw[0], 0);
        if (two) load(bv[1], cw[1], KB);
        compute(bv[0], cw[0]);
        if (two) compute(bv[1], cw[1]);
        // ---- epilogue: the lane holds rows rt * 16 + 4 q + i of column `col` ----
        // Every load of the epilogue is issued before the first store (a load behind a store that may alias it is
        // waited for store by store: sixteen round trips), and the outside neighbours are fetched list position by
        // list position for all sixteen outputs of the lane at once.
        int cnt[RT][4];
        float osc[RT][4], old[RT][4];
        bool ok[RT][4];
        int maxc = 0;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (rt0 + t) * 16 + 4 * q + i;
                ok[t][i] = rt0 + t < rt1 && row < nloc && col_ok;
                const int rowc = min(row, nloc - 1);
                // (>= 0x100: the row has edges in a pair image of its block, spmm_mfma.hip -- this kernel has no pair
                // product: such a row is gathered in full like the negative states)
                const int c = rem_cnt[rowc] >= 0x100 ? -1 : rem_cnt[rowc];
                cnt[t][i] = ok[t][i] ? c : 0;
                osc[t][i] = a.out_scale ? a.out_scale[r0 + rowc] : 1.f;
                old[t][i] = (a.accumulate && ok[t][i]) ? a.y[(int64_t)(r0 + rowc) * a.ldy + colc] : 0.f;
                maxc = max(maxc, cnt[t][i]);
            }
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (cnt[t][i] < 0) acc[t][i] = d32_gather_row<MODE>(a, r0 + (rt0 + t) * 16 + 4 * q + i, col);
        }
        for (int j = 0; j < DB_REM; ++j) {
            if (!__any(maxc > j)) break;
            int u[RT][4];
#pragma unroll
            for (int t = 0; t < RT; ++t) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = min((rt0 + t) * 16 + 4 * q + i, nloc - 1);
                    u[t][i] = cnt[t][i] > j ? rem_col[row * DB_REM + j] : -1;
                }
            }
            float xv[RT][4];
#pragma unroll
            for (int t = 0; t < RT; ++t) {
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[t][i] = u[t][i] >= 0 ? d32_src<MODE>(a, u[t][i], col) : 0.f;
            }
#pragma unroll
            for (int t = 0; t < RT; ++t) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][i] += xv[t][i];
            }
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!ok[t][i]) continue;
                const int g_row = r0 + (rt0 + t) * 16 + 4 * q + i;
                float v = acc[t][i] * osc[t][i];
                const uint64_t yi = a.dr.y_base + (uint64_t)g_row * (uint64_t)a.dr.ld + (uint64_t)col;
                if constexpr (MODE == 1) v *= drop_keep(yi, a.dr.sm, a.dr.p, a.dr.scale);
                if (a.accumulate) {
                    float o_ = old[t][i];
                    if constexpr (MODE == 2) o_ = __fmul_rn(o_, drop_keep(yi, a.dr.sm, a.dr.p, a.dr.scale));
                    v += o_;
                }
                a.y[(int64_t)g_row * a.ldy + col] = v;
            }
        }
    }
    // rows of an oversized block beyond the dense product: gathered in full by the first group's waves
    if (grp == 0 && nrow > DB_ROWS && col_ok) {
        for (int row = DB_ROWS + q; row < nrow; row += 4)
            d32_store<MODE>(a, r0 + row, col, d32_gather_row<MODE>(a, r0 + row, col));
    }
}

}  // namespace

// Which prepared aggregations run here.  Measured on the Reddit-like batch (rocprofv3, profiles/r04_spmm_dense32.txt):
// against the LDS-gather kernel 11.6 vs 10.5 us at D = 256, 13.6 vs 11.8 at 512, 19.5 vs 15.9 at 1024 (both kernels are
// their chain of four or five dependent memory round trips, not their arithmetic: 1 us of MFMA time at D = 256), against
// the bf16x3 matrix-core kernel 37 vs 19 at 2048 and 57 vs 27 at 4096 (the fp32 pipe is a sixteenth of the bf16 one) --
// but 15.7 against 22.0 us for the row-split kernel at D = 602, the one width of the step whose rows are not 16-byte
// aligned.  So: the widths the other blocked kernels do not take (tuning hook GIST_TUNE_SPMM_KERNEL = 3: every width).
bool spmm_dense32_takes(int64_t d, int64_t ldx, int64_t ldy) {
    const int want = (int)tune(GIST_TUNE_SPMM_KERNEL);
    if (want == 1 || want == 2) return false;
    if (!(d >= 16 && ldx >= d && ldy >= d)) return false;
    if (want == 3) return true;
    return d >= 128 && (d % 4 != 0 || ldx % 4 != 0 || ldy % 4 != 0);
}

int launch_spmm_dense32(const int32_t *rowptr, const int32_t *col, const float *x, int64_t ldx, float *y,
                        int64_t ldy, int64_t n_rows, int64_t d, const float *out_scale, const float *src_scale,
                        int accumulate, const int32_t *row_blocks, int64_t n_row_blocks, const void *prepared,
                        hipStream_t st, const SpmmDrop *dr) {
    GIST_REQUIRE(prepared != nullptr, "gist_spmm_csr_prepared_f32: null prepared blocks");
    GIST_REQUIRE(ldy < (1LL << 22) && ldx < (1LL << 22) && d < (1LL << 22),
                 "gist_spmm_csr_prepared_f32: row pitch of 2^22 floats or more");
    D32Args a{};
    a.rowptr = rowptr; a.col = col; a.x = x; a.ldx = ldx; a.y = y; a.ldy = ldy;
    a.n_rows = (int)n_rows; a.d = (int)d; a.out_scale = out_scale; a.src_scale = src_scale;
    a.accumulate = accumulate; a.row_blocks = row_blocks;
    a.prep = static_cast<const unsigned char *>(prepared);
    a.dr = dr ? *dr : SpmmDrop{};
    const int mode = dr ? dr->mode : 0;
    const int64_t nb = row_blocks ? n_row_blocks : ceil_div(n_rows, DB_ROWS);
    if (nb <= 0) return GIST_OK;
    a.n_blocks = (int)nb;
    a.n_col_tiles = (int)ceil_div(d, 16);
    // row-tile groups: enough wave tasks for the chip's 1024 SIMDs (a block has up to 8 row tiles)
    const int64_t col_tasks = nb * a.n_col_tiles;
    int rt = 2;                                                  // (measured: four groups of two row tiles beat two of four up to D = 1024)
    if (col_tasks * 4 > 8192) rt = 4;
    const int forced = (int)tune(GIST_TUNE_SPMM_SPLIT);          // 2, 4: row-tile groups per block
    if (forced == 2 || forced == 4) rt = 8 / forced;
    a.rt_per_group = rt;
    a.groups = 8 / rt;
    const int64_t tasks = col_tasks * a.groups;
    const int64_t grid = ceil_div(tasks, 4);
    if (grid > 0x7fffffffLL) { set_error("gist_spmm_csr_prepared_f32: grid too large"); return GIST_EINVAL; }
#define D32_GO(M, R) hipLaunchKernelGGL((spmm_dense32_kernel<M, R>), dim3((unsigned)grid), dim3(256), 0, st, a)
#define D32_MODE(M)                                          \
    do {                                                     \
        if (rt == 4) D32_GO(M, 4);                           \
        else D32_GO(M, 2);                                   \
    } while (0)
    if (mode == 1) D32_MODE(1);
    else if (mode == 2) D32_MODE(2);
    else D32_MODE(0);
#undef D32_MODE
#undef D32_GO
    return launch_status("gist_spmm_csr_prepared_f32");
}

}  // namespace gist

