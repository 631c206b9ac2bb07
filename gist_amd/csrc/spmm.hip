// CSR SpMM neighbour aggregation for gfx950 (MI355X).
//
// y[v,:] = (acc ? y[v,:] : 0) + out_scale[v] * sum_e src_scale[col[e]] * x[col[e],:]
//
// Replaces g.update_all(fn.copy_src, fn.sum) * norm  (reference
// cluster_gcn/modules.py:223-226) and its autograd (reverse graph, src_scale).
//
// Mapping: one 64-lane wavefront owns one (row, column tile).  The wave reads the
// row's column indices 64 at a time with one coalesced load, then walks them:
//   * wide rows (d/VEC >= 64 lanes): every lane holds VEC consecutive floats of the
//     feature tile; the neighbour id is broadcast with v_readlane (scalar address
//     math), 4 neighbour rows in flight per lane;
//   * narrow rows: the wave is split into G = 64/LPR lane groups, group g takes
//     neighbours g, g+G, ... and the groups are combined by a butterfly
//     (wavefront segmented reduction), so no lane idles on d = 41..128.
// HBM-bound: per launch the algorithmic traffic is rowptr + col + X once + Y once
// (DESIGN.md).  Column tiles are dealt to XCDs (blockIdx % 8 shares an L2) so that
// one tile of X (n_rows x 1 KiB) is gathered out of ONE L2 instead of eight.
#include <stdlib.h>

#include "common.h"

namespace gist {

template <int VEC> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<4> { using T = float4; };

template <int VEC>
__device__ __forceinline__ void vload(const float *p, float (&v)[VEC]) {
    using T = typename Vec<VEC>::T;
    T t = *reinterpret_cast<const T *>(p);
    if constexpr (VEC == 1) { v[0] = t; }
    if constexpr (VEC == 2) { v[0] = t.x; v[1] = t.y; }
    if constexpr (VEC == 4) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
}

template <int VEC>
__device__ __forceinline__ void vstore(float *p, const float (&v)[VEC]) {
    using T = typename Vec<VEC>::T;
    T t;
    if constexpr (VEC == 1) { t = v[0]; }
    if constexpr (VEC == 2) { t.x = v[0]; t.y = v[1]; }
    if constexpr (VEC == 4) { t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3]; }
    *reinterpret_cast<T *>(p) = t;
}

constexpr int kSpmmWavesPerBlock = 4;

// gist_dropout_f32's mask applied where a value is produced or consumed instead of by a pass of its own
// (gist_spmm_csr_drop_f32): element (row, c) of the operand has mask index base + row * ld + c.
//   mode 1 (forward):  what the aggregation stores to y is multiplied by y's mask;
//   mode 2 (backward): x is read through its mask (base = src_base) and the old y of the `y +=` through
//                      y's -- the aggregation of a gradient whose dropout pass has not run.
// (drop_keep / drop_vec / drop_f4: common.h -- the matrix-core kernel of spmm_mfma.hip applies the same masks)
template <int VEC, int LPR, bool DROP = false>
__global__ __launch_bounds__(256) void spmm_csr_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ x, int64_t ldx, float *__restrict__ y, int64_t ldy,
    int n_rows, int d, const float *__restrict__ out_scale,
    const float *__restrict__ src_scale, int accumulate, int n_row_blocks,
    int n_col_tiles, int xcd_tiles, SpmmDrop dr) {
    constexpr int G = kWave / LPR;
    // ---- block -> (row block, column tile) -------------------------------
    int rb, ct;
    const int b = blockIdx.x;
    if (xcd_tiles) {  // tiles dealt to XCDs: blocks b, b+8, b+16.. share an L2
        const int xcd = b % kXcds, i = b / kXcds;
        ct = xcd + kXcds * (i / n_row_blocks);
        rb = i % n_row_blocks;
        if (ct >= n_col_tiles) return;
    } else {
        rb = b / n_col_tiles;
        ct = b % n_col_tiles;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = rb * kSpmmWavesPerBlock + wave;
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const int c0 = (ct * LPR + sub) * VEC;
    const bool active = c0 < d;

    const int beg = __builtin_amdgcn_readfirstlane(rowptr[row]);
    const int end = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);

    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;

    const float *xc = x + c0;
    for (int base = beg; base < end; base += kWave) {
        const int e = base + lane;
        const int my = (e < end) ? col[e] : 0;
        float mys = 1.f;
        if (src_scale) mys = (e < end) ? src_scale[my] : 0.f;
        const int cnt = min(kWave, end - base);
        if constexpr (G == 1) {
            int j = 0;
            for (; j + 4 <= cnt; j += 4) {
                float v[4][VEC];
                float s[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int u = __builtin_amdgcn_readlane(my, j + t);
                    s[t] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                                                          __builtin_bit_cast(int, mys), j + t));
                    if (active) vload<VEC>(xc + (int64_t)u * ldx, v[t]);
                }
                if (active) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(s[t], v[t][k], acc[k]);
                }
            }
            for (; j < cnt; ++j) {
                const int u = __builtin_amdgcn_readlane(my, j);
                const float s = __builtin_bit_cast(
                    float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mys), j));
                if (active) {
                    float v[VEC];
                    vload<VEC>(xc + (int64_t)u * ldx, v);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = fmaf(s, v[k], acc[k]);
                }
            }
        } else {
            // group g handles neighbours g, g+G, ...; two in flight per lane.  The trip
            // count is wave-uniform so every lane is live at each cross-lane read.
            const int iters = (cnt + 2 * G - 1) / (2 * G);
            for (int t = 0; t < iters; ++t) {
                const int j = grp + t * 2 * G, j2 = j + G;
                const bool p0 = j < cnt, p1 = j2 < cnt;
                const int u0 = __shfl(my, p0 ? j : 0), u1 = __shfl(my, p1 ? j2 : 0);
                const float t0 = __shfl(mys, p0 ? j : 0), t1 = __shfl(mys, p1 ? j2 : 0);
                const float s0 = p0 ? t0 : 0.f, s1 = p1 ? t1 : 0.f;
                if (active) {
                    float v0[VEC], v1[VEC];
                    vload<VEC>(xc + (int64_t)u0 * ldx, v0);
                    vload<VEC>(xc + (int64_t)u1 * ldx, v1);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = fmaf(s0, v0[k], acc[k]);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = fmaf(s1, v1[k], acc[k]);
                }
            }
        }
    }
    if constexpr (G > 1) {  // butterfly across the lane groups
#pragma unroll
        for (int off = LPR; off < kWave; off <<= 1)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] += __shfl_xor(acc[k], off);
    }
    if (active && grp == 0) {
        const float os = out_scale ? out_scale[row] : 1.f;
        float *yp = y + (int64_t)row * ldy + c0;
        float o[VEC];
        if (accumulate) {
            vload<VEC>(yp, o);
#pragma unroll
            for (int k = 0; k < VEC; ++k) o[k] = fmaf(os, acc[k], o[k]);
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) o[k] = os * acc[k];
        }
        if constexpr (DROP) drop_vec<VEC>(o, dr.y_base + (uint64_t)row * (uint64_t)dr.ld + (uint64_t)c0, dr);
        vstore<VEC>(yp, o);
    }
}

// ---------------------------------------------------------------------------------------
// Wide rows (d/VEC >= 64 lanes): ONE WORKGROUP PER ROW.  A single wave walking a row is a
// chain of dependent L2 round trips (4 loads per trip): a 330-neighbour hub row of a
// power-law graph takes ~80 trips and sets the kernel's duration.  Here the row's
// neighbour list is dealt to the 4 waves in interleaved slices of 16 (wave w takes
// neighbours [16w, 16w+16) of every 64), 8 row reads in flight per wave (16 in flight cost
// 158 VGPRs and ran slower), and the four partial sums are combined through LDS in wave
// order (deterministic).  A 64-neighbour row costs 2 trips per wave instead of 16; a hub
// row 11 instead of 83.
//
// HALF (VEC = 4 only): d % 4 == 2 with 16-byte aligned source rows but an output that is only
// 8-byte aligned -- the layer-0 aggregation of F = 602 features inside Z_0 = [h | ah], whose right
// half starts 2408 bytes into a 4816-byte row.  Sources are still gathered 16 bytes per lane (the
// last lane of a row reads two floats past d, inside the row pitch, and never stores them);
// the result leaves as two 8-byte stores.
template <int VEC, bool HALF = false, bool DROP = false>
__global__ __launch_bounds__(256) void spmm_csr_rowsplit_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ x, int64_t ldx, float *__restrict__ y, int64_t ldy,
    int n_rows, int d, const float *__restrict__ out_scale,
    const float *__restrict__ src_scale, int accumulate, int n_col_tiles, int chunk_rows, SpmmDrop dr) {
    static_assert(!HALF || VEC == 4, "HALF is the 16-byte-load / 8-byte-store variant");
    __shared__ float part[3][kWave * VEC];
    // Row chunks (~ one METIS part: consecutive batch rows whose neighbours are mostly in the
    // same chunk) are dealt to XCDs -- workgroups b, b+8, b+16, ... share an L2 -- so the
    // rows an XCD gathers are the ones its own L2 already holds, instead of every L2
    // missing on all of X.  (Speed only: the result does not depend on placement.)
    const int b = blockIdx.x;
    const int xcd = b % kXcds, i = b / kXcds;
    const int per = chunk_rows * n_col_tiles;
    const int rem = i % per;
    const int row = (xcd + kXcds * (i / per)) * chunk_rows + rem / n_col_tiles;
    const int ct = rem % n_col_tiles;
    if (row >= n_rows) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int c0 = (ct * kWave + lane) * VEC;
    const bool active = c0 < d;
    const int beg = __builtin_amdgcn_readfirstlane(rowptr[row]);
    const int end = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);

    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    const float *xc = x + c0;
    // lanes 0..15 of wave w hold the ids of neighbours base + 16w + lane
    for (int base = beg + 16 * wave; base < end; base += kWave) {
        const int e = base + (lane & 15);
        const bool ok = (lane < 16) && (e < end);
        const int my = ok ? col[e] : 0;
        float mys = 1.f;
        if (src_scale) mys = ok ? src_scale[my] : 0.f;
        const int cnt = min(16, end - base);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (8 * h < cnt) {                            // wave-uniform
                float v[8][VEC];
                float s[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int jj = min(8 * h + t, cnt - 1);
                    const int u = __builtin_amdgcn_readlane(my, jj);
                    const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                                                                    __builtin_bit_cast(int, mys), jj));
                    s[t] = (8 * h + t < cnt) ? s0 : 0.f;   // clamped duplicates contribute 0
                    if (active) vload<VEC>(xc + (int64_t)u * ldx, v[t]);
                }
                if (active) {
#pragma unroll
                    for (int t = 0; t < 8; ++t)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(s[t], v[t][k], acc[k]);
                }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) part[wave - 1][lane * VEC + k] = acc[k];
    }
    __syncthreads();
    if (wave == 0 && active) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] += part[w][lane * VEC + k];
        const float os = out_scale ? out_scale[row] : 1.f;
        float *yp = y + (int64_t)row * ldy + c0;
        if constexpr (HALF) {
            const bool upper = c0 + 2 < d;                 // false only for the row's last lane
            float lo[2] = {0.f, 0.f}, hi[2] = {0.f, 0.f};
            if (accumulate) {
                vload<2>(yp, lo);
                if (upper) vload<2>(yp + 2, hi);
            }
            lo[0] = fmaf(os, acc[0], lo[0]); lo[1] = fmaf(os, acc[1], lo[1]);
            hi[0] = fmaf(os, acc[2], hi[0]); hi[1] = fmaf(os, acc[3], hi[1]);
            if constexpr (DROP) {
                const uint64_t i0 = dr.y_base + (uint64_t)row * (uint64_t)dr.ld + (uint64_t)c0;
                drop_vec<2>(lo, i0, dr);
                drop_vec<2>(hi, i0 + 2, dr);
            }
            vstore<2>(yp, lo);
            if (upper) vstore<2>(yp + 2, hi);
        } else {
            float o[VEC];
            if (accumulate) {
                vload<VEC>(yp, o);
#pragma unroll
                for (int k = 0; k < VEC; ++k) o[k] = fmaf(os, acc[k], o[k]);
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) o[k] = os * acc[k];
            }
            if constexpr (DROP) drop_vec<VEC>(o, dr.y_base + (uint64_t)row * (uint64_t)dr.ld + (uint64_t)c0, dr);
            vstore<VEC>(yp, o);
        }
    }
}

// ---------------------------------------------------------------------------------------
// LDS-staged kernel (the one the training step uses for wide rows): gist_spmm_csr_blocked_f32.
//
// Unit of work = (locality block of <= 128 consecutive rows -- a METIS part of the cluster
// batch --, 256-float column tile, row split).  One 1024-thread workgroup (16 waves, the whole
// CU: the 128 KiB tile allows one) per unit:
//   start-up, all global loads in flight together, ONE barrier
//     * the block's X tile [rows x 1 KiB] -> LDS, pre-scaled by src_scale (backward form), one
//       all-zero row behind it; the block's row pointers and out scales -> LDS;
//     * a 1-BYTE local row index per neighbour of the block's rows -> LDS (cross-block
//       neighbour: 128 = the zero row): it depends on the neighbour and the block only, so the
//       whole workgroup classifies the block's edge range in one flat coalesced pass and the
//       row loop reads no column ids from memory;
//   rows longer than 128 neighbours (hubs of a power-law graph) first, split over all 16 waves in
//   64-neighbour chunks, partial sums added in wave order through LDS;
//   every other row by one wave: per neighbour v_readlane + v_add + ds_read_b128 + 2 v_pk_add_f32,
//   groups of 8 reads software-pipelined and fully unrolled (constant lanes: address arithmetic on
//   the scalar unit, which issues one instruction per SIMD every 4 cycles, was the first
//   bottleneck); the next row's indices are read before the current row is gathered; cross-block
//   neighbours (a ballot mask) come from global memory, 4 rows in flight.
// X leaves L2/HBM once per unit instead of once per neighbour: at D = 4096 on a Reddit-like
// batch nnz*D*4 = 2.1 GB of L2 gather traffic becomes LDS reads.  Measured (scripts/
// spmm_probe.py, scripts/lds_gather_probe.hip): 82 -> 51 us at D = 4096;
// the gather loop itself runs at ~6 cycles per KiB per CU (LDS peak 4), the rest is the
// start-up of each unit, which one workgroup per CU cannot hide.  Tried and dropped: 512-thread
// workgroups that loop over column tiles with the next tile prefetched into registers (8 waves
// do not cover the LDS latency: 64 us), and 40-KiB workgroups (4 per CU) on 64-float tiles where
// one ds_read_b128 serves four neighbours with v_add_u32_dpp row_newbcast addressing (73 us).
// Results do not depend on the block boundaries, the row split or the placement.
constexpr int kL2Rows = 128;                 // rows of X staged per block
constexpr int kL2Threads = 1024;
constexpr int kL2Waves = kL2Threads / kWave;
constexpr int kL2Pre = kL2Rows / kL2Waves;   // rows of the tile each wave stages
constexpr int kL2Long = 128;                 // rows with more neighbours are shared by all waves
constexpr int kL2ZeroOff = kL2Rows * 1024;   // byte offset of the zero row
constexpr size_t kL2PartOff = (size_t)(kL2Rows + 1) * 1024;            // 16 x 1 KiB partial sums
constexpr size_t kL2RpOff = kL2PartOff + (size_t)kL2Waves * 1024;      // row pointers of the block
constexpr size_t kL2ScOff = kL2RpOff + (kL2Rows + 4) * 4;              // out_scale of the block's rows
constexpr size_t kL2MaskOff = kL2ScOff + kL2Rows * 4;                  // 2 x 64-bit: long rows
constexpr size_t kL2IdxOff = kL2MaskOff + 16;                          // 1-byte neighbour indices
constexpr int kL2IdxCap = 13312;
constexpr size_t kL2LdsBytes = kL2IdxOff + kL2IdxCap;
static_assert(kL2LdsBytes <= 160 * 1024, "LDS budget of one CU");

struct L2Args {
    const int32_t *rowptr, *col;
    const float *x; int64_t ldx;
    float *y; int64_t ldy;
    int n_rows, d;
    const float *out_scale, *src_scale;
    int accumulate;
    const int32_t *row_blocks;
    int n_blocks, n_col_tiles, row_split;
    SpmmDrop dr;
    SpmmLnBwd ln;                // LNB instantiation only
};

#ifdef L2_PROBE_NO_LDS
#define L2_LDS_READ(o) make_float4(__builtin_bit_cast(float, o), 0.f, 0.f, 0.f)
#else
#define L2_LDS_READ(o) (*reinterpret_cast<const float4 *>(tb + (o)))
#endif
#ifdef L2_PROBE_NO_ADD
#define L2_ADD(v) acc[0] += (v).x
#else
#define L2_ADD(v) acc[0] += (v).x; acc[1] += (v).y; acc[2] += (v).z; acc[3] += (v).w
#endif

// Sum the neighbours [base, base + cnt) of one row into acc: `off` holds, per lane, the LDS
// byte offset of neighbour base + lane (zero row if it is not in the tile).
__device__ __forceinline__ void l2_local(const unsigned char *tb, int cnt, int off, float (&acc)[4]) {
    // 8 groups of 8 neighbours, fully unrolled so that every v_readlane has a constant lane and
    // the loop costs no scalar instructions beyond one branch per group; group g + 1 is issued
    // before group g is summed (16 row reads in flight per wave, 128 per CU: the LDS pipe always
    // has work queued)
    float4 v[2][8];
#pragma unroll
    for (int t = 0; t < 8; ++t) v[0][t] = L2_LDS_READ(__builtin_amdgcn_readlane(off, t));
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const bool more = (g + 1) * 8 < cnt;                   // wave-uniform
        if (g < 7 && more) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
                v[(g + 1) & 1][t] = L2_LDS_READ(__builtin_amdgcn_readlane(off, ((g + 1) * 8 + t) & 63));
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) { L2_ADD(v[g & 1][t]); }
        if (!more) break;
    }
}

// Cross-block neighbours of a chunk: lanes in rmask hold the id `u` and scale `us` of one; 4 row
// reads from global memory in flight (the mask is wave-uniform).
template <bool SRC_DROP>
__device__ __forceinline__ void l2_remote(const float *xc, int64_t ldx, bool active,
                                          unsigned long long rmask, int u, float us, float (&acc)[4],
                                          const SpmmDrop &dr, int c0) {
    while (rmask) {
        float4 rv[4];
        float rs[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool on = rmask != 0;
            const int j = on ? __builtin_ctzll(rmask) : 0;
            if (on) rmask &= rmask - 1;
            const int uu = __builtin_amdgcn_readlane(u, j);
            const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                                                            __builtin_bit_cast(int, us), j));
            rs[t] = on ? s0 : 0.f;
            rv[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on && active) {
                rv[t] = *reinterpret_cast<const float4 *>(xc + (int64_t)uu * ldx);
                if constexpr (SRC_DROP)
                    drop_f4(rv[t], dr.src_base + (uint64_t)uu * (uint64_t)dr.ld + (uint64_t)c0, dr);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[0] = fmaf(rs[t], rv[t].x, acc[0]); acc[1] = fmaf(rs[t], rv[t].y, acc[1]);
            acc[2] = fmaf(rs[t], rv[t].z, acc[2]); acc[3] = fmaf(rs[t], rv[t].w, acc[3]);
        }
    }
}

// LNB (with DROP == 2, one column tile): the store is the LayerNorm + ReLU backward of the layer below (SpmmLnBwd)
template <int DROP, bool LNB = false>      // 0 none, 1 mask on what is stored (forward), 2 masks on x and on the old y (backward)
__global__ __launch_bounds__(kL2Threads) void spmm_csr_lds2_kernel(L2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    // ---- workgroup -> unit: units of one block stay on one XCD (blocks b, b + 8, .. share one)
    const int total = a.n_blocks * a.n_col_tiles * a.row_split;
    const int per_xcd = (total + kXcds - 1) / kXcds;
    const int unit = (int)(blockIdx.x % kXcds) * per_xcd + (int)(blockIdx.x / kXcds);
    if (unit >= total) return;
    const int rs = unit % a.row_split;
    const int ct = (unit / a.row_split) % a.n_col_tiles;
    const int rbk = unit / (a.row_split * a.n_col_tiles);
    int r0, r1;
    if (a.row_blocks) { r0 = a.row_blocks[rbk]; r1 = a.row_blocks[rbk + 1]; }
    else { r0 = rbk * kL2Rows; r1 = r0 + kL2Rows; }
    r1 = min(r1, a.n_rows);
    const int nrow = r1 - r0;
    if (nrow <= 0) {
        if constexpr (LNB)      // (an empty block still owns its rows of the partial sums)
            for (int c = threadIdx.x; c < a.d; c += kL2Threads) a.ln.col_partials[(int64_t)unit * a.d + c] = 0.f;
        return;
    }
    const int nloc = min(nrow, kL2Rows);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = ct * 256 + lane * 4;
    const bool active = c0 < a.d;
    const float *xc = a.x + c0;
    unsigned char *tb = smem_b + lane * 16;
    unsigned char *part = smem_b + kL2PartOff;
    int32_t *rp = reinterpret_cast<int32_t *>(smem_b + kL2RpOff);
    float *rsc = reinterpret_cast<float *>(smem_b + kL2ScOff);
    unsigned long long *lmask = reinterpret_cast<unsigned long long *>(smem_b + kL2MaskOff);
    unsigned char *idx8 = smem_b + kL2IdxOff;
    const int step = a.row_split * kL2Waves;

    // ---- start-up: row pointers, out scales, long rows; the tile (pre-scaled); neighbour indices --
    if (wave == 0) *reinterpret_cast<float4 *>(tb + kL2ZeroOff) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.x < kL2Rows) {                  // waves 0 and 1
        const int t = threadIdx.x;
        const int b0 = t < nloc ? a.rowptr[r0 + t] : 0;
        const int b1 = t < nloc ? a.rowptr[r0 + t + 1] : 0;
        if (t < nloc) {
            rp[t] = b0;
            rsc[t] = a.out_scale ? a.out_scale[r0 + t] : 1.f;
        }
        if (t == nloc - 1) rp[nloc] = b1;
        const bool lg = t < nloc && (b1 - b0) > kL2Long && (t % a.row_split) == rs;
        const unsigned long long m = __ballot(lg);
        if (lane == 0) lmask[wave] = m;
    }
#ifndef L2_PROBE_NO_STAGE
    {
        float4 v[kL2Pre];
        float sc[kL2Pre];
#pragma unroll
        for (int t = 0; t < kL2Pre; ++t) {
            const int r = min(wave + t * kL2Waves, nloc - 1);
            v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (active) v[t] = *reinterpret_cast<const float4 *>(xc + (int64_t)(r0 + r) * a.ldx);
            sc[t] = a.src_scale ? a.src_scale[r0 + r] : 1.f;
        }
#pragma unroll
        for (int t = 0; t < kL2Pre; ++t) {
            const int r = wave + t * kL2Waves;
            if (r < nloc) {
                if constexpr (DROP == 2)
                    if (active) drop_f4(v[t], a.dr.src_base + (uint64_t)(r0 + r) * (uint64_t)a.dr.ld + (uint64_t)c0, a.dr);
                v[t].x *= sc[t]; v[t].y *= sc[t]; v[t].z *= sc[t]; v[t].w *= sc[t];
                *reinterpret_cast<float4 *>(tb + (r << 10)) = v[t];
            }
        }
    }
#endif
    // 1-byte local index of EVERY neighbour of the block's rows (128: not in the tile): it depends
    // on the neighbour and the block only, so all threads classify the block's edge range in one
    // flat coalesced pass whose loads fly together with the tile's
    const int e0 = a.rowptr[r0];
    {
        const int e1 = min(a.rowptr[r0 + nloc], e0 + kL2IdxCap);
        for (int e = e0 + threadIdx.x; e < e1; e += kL2Threads) {
            const unsigned lu = (unsigned)(a.col[e] - r0);
            idx8[e - e0] = (unsigned char)(lu < (unsigned)nloc ? lu : kL2Rows);
        }
    }
    __syncthreads();
#ifdef L2_PROBE_NO_GATHER
    return;
#endif
    unsigned long long lm[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const unsigned long long m = lmask[h];
        lm[h] = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) |
                (unsigned)__builtin_amdgcn_readfirstlane((int)m);
    }

    // one 64-neighbour chunk [base, end) of a row: the lane's neighbour as an LDS byte offset, and
    // whether it is a cross-block row (then `u` is its id)
    auto classify = [&](int base, int end, bool staged, int &off, bool &rem, int &u) {
        const int e = base + lane;
        const bool in = e < end;
        u = r0;
        if (staged) {                                      // wave-uniform: indices in LDS
            const int li = in ? (int)idx8[e - e0] : kL2Rows;
            off = li << 10;
            rem = in && li == kL2Rows;
        } else {
            u = in ? a.col[e] : r0;
            const unsigned lu = (unsigned)(u - r0);
            const bool local = in && lu < (unsigned)nloc;
            off = local ? (int)(lu << 10) : kL2ZeroOff;
            rem = in && !local;
        }
    };
    auto remote = [&](int base, int end, bool staged, bool rem, int u, float (&acc)[4]) {
        const unsigned long long rmask = __ballot(rem);
        if (!rmask) return;
        if (staged) u = rem ? a.col[base + lane] : r0;
        const float us = (rem && a.src_scale) ? a.src_scale[u] : 1.f;
        l2_remote<DROP == 2>(xc, a.ldx, active, rmask, u, us, acc, a.dr, c0);
    };
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);      // LNB: column sums of the dy rows this wave stored
    auto yhat_of = [&](int lr) {                       // LNB: the row's normalised output, read early
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (LNB)
            if (active) v = *reinterpret_cast<const float4 *>(a.ln.yhat + (int64_t)(r0 + lr) * a.ln.ldy + c0);
        return v;
    };
    auto store = [&](int lr, float4 prev, float4 yv, float sx, float sy, float sz, float sw) {
        const int row = r0 + lr;
        if constexpr (LNB) {      // (every lane takes part in the row sums; lanes beyond d hold zeros)
            const float os = lr < nloc ? rsc[lr] : (a.out_scale ? a.out_scale[row] : 1.f);
            float4 g = make_float4(fmaf(os, sx, prev.x), fmaf(os, sy, prev.y), fmaf(os, sz, prev.z), fmaf(os, sw, prev.w));
            if (!active) g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.ln.relu) {
                g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
                g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
            }
            if (a.ln.rstd != nullptr) {
                float s1 = (g.x + g.y) + (g.z + g.w);
                float s2 = (g.x * yv.x + g.y * yv.y) + (g.z * yv.z + g.w * yv.w);
                s1 = wave_sum(s1);
                s2 = wave_sum(s2);
                const float m1 = s1 / (float)a.d, m2 = s2 / (float)a.d, rstd = a.ln.rstd[row];
                g.x = rstd * (g.x - m1 - yv.x * m2); g.y = rstd * (g.y - m1 - yv.y * m2);
                g.z = rstd * (g.z - m1 - yv.z * m2); g.w = rstd * (g.w - m1 - yv.w * m2);
            }
            if (!active) return;
            *reinterpret_cast<float4 *>(a.ln.dy + (int64_t)row * a.ln.lddy + c0) = g;
            cs.x += g.x; cs.y += g.y; cs.z += g.z; cs.w += g.w;
            return;
        }
        if (!active) return;
        const float os = lr < nloc ? rsc[lr] : (a.out_scale ? a.out_scale[row] : 1.f);
        float4 *yp = reinterpret_cast<float4 *>(a.y + (int64_t)row * a.ldy + c0);
        float4 o = make_float4(fmaf(os, sx, prev.x), fmaf(os, sy, prev.y), fmaf(os, sz, prev.z),
                               fmaf(os, sw, prev.w));
        if constexpr (DROP == 1) drop_f4(o, a.dr.y_base + (uint64_t)row * (uint64_t)a.dr.ld + (uint64_t)c0, a.dr);
        *yp = o;
    };
    auto previous = [&](int lr) {                          // y's old value, read early (accumulate)
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.accumulate && active) {
            p = *reinterpret_cast<const float4 *>(a.y + (int64_t)(r0 + lr) * a.ldy + c0);
            if constexpr (DROP == 2) drop_f4(p, a.dr.y_base + (uint64_t)(r0 + lr) * (uint64_t)a.dr.ld + (uint64_t)c0, a.dr);
        }
        return p;
    };

    // ---- long rows first (all waves in step): wave w takes the 64-neighbour chunks w, w + 16, ..;
    //      the partial sums meet in LDS and are added in wave order by wave (k mod 16) ---------------
#ifndef L2_PROBE_NO_LONG
    if (lm[0] | lm[1]) {
        int k = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned long long m = lm[h];
            while (m) {
                const int lr = 64 * h + __builtin_ctzll(m);
                m &= m - 1;
                const int beg = __builtin_amdgcn_readfirstlane(rp[lr]);
                const int end = __builtin_amdgcn_readfirstlane(rp[lr + 1]);
                const bool staged = end - e0 <= kL2IdxCap;
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                for (int base = beg + kWave * wave; base < end; base += kWave * kL2Waves) {
                    int off, u;
                    bool rem;
                    classify(base, end, staged, off, rem, u);
                    l2_local(tb, min(kWave, end - base), off, acc);
                    remote(base, end, staged, rem, u, acc);
                }
                if (k > 0) __syncthreads();               // previous long row's partials consumed
                *reinterpret_cast<float4 *>(part + wave * 1024 + lane * 16) =
                    make_float4(acc[0], acc[1], acc[2], acc[3]);
                __syncthreads();
                if (wave == (k & (kL2Waves - 1))) {
                    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int w = 0; w < kL2Waves; ++w) {
                        const float4 p = *reinterpret_cast<const float4 *>(part + w * 1024 + lane * 16);
                        sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w;
                    }
                    store(lr, previous(lr), yhat_of(lr), sum.x, sum.y, sum.z, sum.w);
                }
                ++k;
            }
        }
    }
#endif

    // ---- the other rows of this unit, one wave per row; the next row's first chunk is classified
    //      (its LDS index read issued) before the current row is gathered ----------------------------
    {
        int lr = rs + a.row_split * wave;
        int beg = 0, end = 0, off = kL2ZeroOff, u = r0;
        bool rem = false, staged = true;
        auto open_row = [&](int r) {                       // first chunk of the next short row >= r
            while (r < nloc) {
                beg = __builtin_amdgcn_readfirstlane(rp[r]);
                end = __builtin_amdgcn_readfirstlane(rp[r + 1]);
                if (end - beg <= kL2Long) break;           // long rows were done above
                r += step;
            }
            if (r < nloc) {
                staged = end - e0 <= kL2IdxCap;
                classify(beg, end, staged, off, rem, u);
            }
            return r;
        };
        lr = open_row(lr);
        while (lr < nloc) {
            const int cur = lr, cbeg = beg, cend = end, coff = off, cu = u;
            const bool crem = rem, cstaged = staged;
            const float4 prev = previous(cur);
            const float4 yv = yhat_of(cur);
            lr = open_row(cur + step);                      // next row's indices on their way
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            l2_local(tb, min(kWave, cend - cbeg), coff, acc);
            remote(cbeg, cend, cstaged, crem, cu, acc);
            for (int base = cbeg + kWave; base < cend; base += kWave) {   // rows of 65..128 neighbours
                int o2, u2;
                bool r2;
                classify(base, cend, cstaged, o2, r2, u2);
                l2_local(tb, min(kWave, cend - base), o2, acc);
                remote(base, cend, cstaged, r2, u2, acc);
            }
            store(cur, prev, yv, acc[0], acc[1], acc[2], acc[3]);
        }
    }
    // rows of an oversize block beyond the staged ones (their row pointers are not in LDS)
    for (int lr = kL2Rows + rs + a.row_split * wave; lr < nrow; lr += step) {
        const int row = r0 + lr;
        const int beg = __builtin_amdgcn_readfirstlane(a.rowptr[row]);
        const int end = __builtin_amdgcn_readfirstlane(a.rowptr[row + 1]);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int base = beg; base < end; base += kWave) {
            int off, u;
            bool rem;
            classify(base, end, false, off, rem, u);
            l2_local(tb, min(kWave, end - base), off, acc);
            remote(base, end, false, rem, u, acc);
        }
        store(lr, previous(lr), yhat_of(lr), acc[0], acc[1], acc[2], acc[3]);
    }
    if constexpr (LNB) {      // the workgroup's column sums: the waves' sums meet in LDS and are added in wave order
        __syncthreads();      // (the long rows' partial sums are consumed)
        *reinterpret_cast<float4 *>(part + wave * 1024 + lane * 16) = cs;
        __syncthreads();
        if (wave == 0 && active) {
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int w = 0; w < kL2Waves; ++w) {
                const float4 q = *reinterpret_cast<const float4 *>(part + w * 1024 + lane * 16);
                sum.x += q.x; sum.y += q.y; sum.z += q.z; sum.w += q.w;
            }
            *reinterpret_cast<float4 *>(a.ln.col_partials + (int64_t)unit * a.d + c0) = sum;
        }
    }
}

__global__ void in_degree_norm_kernel(const int32_t *__restrict__ rowptr, int64_t n,
                                      float *__restrict__ norm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int deg = rowptr[i + 1] - rowptr[i];
        norm[i] = deg > 0 ? 1.f / (float)deg : 0.f;
    }
}

template <int VEC, bool HALF = false>
static int launch_spmm(const int32_t *rowptr, const int32_t *col, const float *x, int64_t ldx,
                       float *y, int64_t ldy, int64_t n_rows, int64_t d, const float *out_scale,
                       const float *src_scale, int accumulate, hipStream_t st,
                       const SpmmDrop *drop = nullptr) {
    const bool dropping = drop != nullptr && drop->mode == 1 && drop->p > 0.f;
    SpmmDrop dr{};
    if (dropping) dr = *drop;
    const int lanes = (int)ceil_div(d, VEC);
    int lpr = 8;
    while (lpr < 64 && lpr < lanes) lpr <<= 1;
    const int n_col_tiles = (int)ceil_div(lanes, lpr);
    const int n_row_blocks = (int)ceil_div(n_rows, kSpmmWavesPerBlock);
    const int xcd_tiles = n_col_tiles >= 8 ? 1 : 0;
    if (lpr == 64) {   // wide rows: one workgroup per (row, column tile), waves share the row
        const int chunk_rows = tune(GIST_TUNE_SPMM_CHUNK) > 0.0 ? (int)tune(GIST_TUNE_SPMM_CHUNK) : 128;
        const int64_t n_chunks = ceil_div(n_rows, chunk_rows);
        const int64_t g2 = kXcds * ceil_div(n_chunks, kXcds) * chunk_rows * n_col_tiles;
        if (g2 > 0x7fffffffLL) {
            set_error("gist_spmm_csr_f32: grid too large");
            return GIST_EINVAL;
        }
        if (dropping)
            hipLaunchKernelGGL((spmm_csr_rowsplit_kernel<VEC, HALF, true>), dim3((unsigned)g2), dim3(256), 0,
                               st, rowptr, col, x, ldx, y, ldy, (int)n_rows, (int)d, out_scale,
                               src_scale, accumulate, n_col_tiles, chunk_rows, dr);
        else
            hipLaunchKernelGGL((spmm_csr_rowsplit_kernel<VEC, HALF, false>), dim3((unsigned)g2), dim3(256), 0,
                               st, rowptr, col, x, ldx, y, ldy, (int)n_rows, (int)d, out_scale,
                               src_scale, accumulate, n_col_tiles, chunk_rows, dr);
        return launch_status("gist_spmm_csr_f32");
    }
    const int64_t grid = xcd_tiles ? (int64_t)kXcds * ceil_div(n_col_tiles, kXcds) * n_row_blocks
                                   : (int64_t)n_row_blocks * n_col_tiles;
    if (grid > 0x7fffffffLL) {
        set_error("gist_spmm_csr_f32: grid too large");
        return GIST_EINVAL;
    }
#define GIST_SPMM_LAUNCH(L)                                                                       \
    do {                                                                                          \
        if (dropping)                                                                             \
            hipLaunchKernelGGL((spmm_csr_kernel<VEC, L, true>), dim3((unsigned)grid), dim3(256), 0, st, \
                               rowptr, col, x, ldx, y, ldy, (int)n_rows, (int)d, out_scale,       \
                               src_scale, accumulate, n_row_blocks, n_col_tiles, xcd_tiles, dr);  \
        else                                                                                      \
            hipLaunchKernelGGL((spmm_csr_kernel<VEC, L, false>), dim3((unsigned)grid), dim3(256), 0, st, \
                               rowptr, col, x, ldx, y, ldy, (int)n_rows, (int)d, out_scale,       \
                               src_scale, accumulate, n_row_blocks, n_col_tiles, xcd_tiles, dr);  \
    } while (0)
    switch (lpr) {
        case 8: GIST_SPMM_LAUNCH(8); break;
        case 16: GIST_SPMM_LAUNCH(16); break;
        case 32: GIST_SPMM_LAUNCH(32); break;
        default: GIST_SPMM_LAUNCH(64); break;
    }
#undef GIST_SPMM_LAUNCH
    return launch_status("gist_spmm_csr_f32");
}

// Row split R of the second LDS design: units = blocks x column tiles x R on 256 CUs, one unit
// per CU at a time; every split re-stages the tile (cost ~1) and gathers 1/R of the block's rows
// (cost ~2.5 / R), fitted to scripts/spmm_probe.py on a Reddit-like batch.  (R up to 16 since round 5: 20 blocks of <= 256
// columns take R = 12 = 240 units in one round -- config 2 0.2994 -> 0.2918 ms/step against R = 8, h = 256 0.1782 -> 0.1741.)
static int l2_row_split(int64_t nb, int n_col_tiles) {
    int best = 1;
    double best_cost = 1e30;
    for (int r = 1; r <= 16; ++r) {
        const double rounds = (double)ceil_div(nb * n_col_tiles * r, 256);
        const double cost = rounds * (1.0 + 3.0 / r);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = r; }
    }
    return best;
}

static int l2_split_for(int64_t nb, int n_col_tiles) {
    const int r = tune(GIST_TUNE_SPMM_SPLIT) > 0.0 ? (int)tune(GIST_TUNE_SPMM_SPLIT) : l2_row_split(nb, n_col_tiles);
    return r < 1 ? 1 : (r > 64 ? 64 : r);
}

static int launch_spmm_lds2(const int32_t *rowptr, const int32_t *col, const float *x, int64_t ldx,
                            float *y, int64_t ldy, int64_t n_rows, int64_t d, const float *out_scale,
                            const float *src_scale, int accumulate, const int32_t *row_blocks,
                            int64_t n_row_blocks, hipStream_t st, const SpmmDrop *drop = nullptr,
                            const SpmmLnBwd *ln = nullptr) {
    L2Args a;
    a.dr = SpmmDrop{};
    a.ln = ln ? *ln : SpmmLnBwd{};
    const int dmode = (drop != nullptr && drop->p > 0.f) ? drop->mode : 0;
    if (dmode) a.dr = *drop;
    a.rowptr = rowptr; a.col = col; a.x = x; a.ldx = ldx; a.y = y; a.ldy = ldy;
    a.n_rows = (int)n_rows; a.d = (int)d; a.out_scale = out_scale; a.src_scale = src_scale;
    a.accumulate = accumulate; a.row_blocks = row_blocks;
    const int64_t nb = row_blocks ? n_row_blocks : ceil_div(n_rows, kL2Rows);
    a.n_blocks = (int)nb;
    a.n_col_tiles = (int)ceil_div(d, 256);
    a.row_split = l2_split_for(nb, a.n_col_tiles);
    const int64_t total = nb * a.n_col_tiles * a.row_split;
    const int64_t grid = kXcds * ceil_div(total, kXcds);
    if (grid > 0x7fffffffLL) { set_error("gist_spmm_csr_blocked_f32: grid too large"); return GIST_EINVAL; }
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        hipError_t e = hipSuccess;
        for (const void *fn : {reinterpret_cast<const void *>(&spmm_csr_lds2_kernel<0>),
                               reinterpret_cast<const void *>(&spmm_csr_lds2_kernel<1>),
                               reinterpret_cast<const void *>(&spmm_csr_lds2_kernel<2>),
                               reinterpret_cast<const void *>(&spmm_csr_lds2_kernel<2, true>)})
            if (e == hipSuccess)
                e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kL2LdsBytes);
        if (e != hipSuccess) {
            set_error("gist_spmm_csr_blocked_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return GIST_ELAUNCH;
        }
        once.done(dev);
    }
    if (ln != nullptr) {
        if (dmode != 2 || a.n_col_tiles != 1 || row_blocks == nullptr) {
            set_error("gist_spmm_csr_drop_f32: the LayerNorm-backward store needs mode 2, d <= 256 and row blocks");
            return GIST_EINVAL;
        }
        hipLaunchKernelGGL((spmm_csr_lds2_kernel<2, true>), dim3((unsigned)grid), dim3(kL2Threads), kL2LdsBytes, st, a);
    } else if (dmode == 1)
        hipLaunchKernelGGL(spmm_csr_lds2_kernel<1>, dim3((unsigned)grid), dim3(kL2Threads), kL2LdsBytes, st, a);
    else if (dmode == 2)
        hipLaunchKernelGGL(spmm_csr_lds2_kernel<2>, dim3((unsigned)grid), dim3(kL2Threads), kL2LdsBytes, st, a);
    else
        hipLaunchKernelGGL(spmm_csr_lds2_kernel<0>, dim3((unsigned)grid), dim3(kL2Threads), kL2LdsBytes, st, a);
    return launch_status("gist_spmm_csr_blocked_f32");
}

// Calls the prepared matrix-core kernel takes (everything else: gist_spmm_csr_blocked_f32's choice).
// Measured with the block structure prepared (rocprofv3, Reddit-like batch): D = 4096 30.5 us (38 when
// every workgroup builds its own, 57 on the LDS gather kernel), 2048 21 (29, 33), 1024 13.5 (21.5, 16),
// 512 13.4 (20, 12); the preparation is one 13 us launch per batch, so it pays from D = 2048 on.
bool spmm_prepared_takes(int64_t d, int64_t ldx, int64_t ldy, const float *x, const float *y) {
    return d >= 1536 && d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && aligned16(x) && aligned16(y) &&
           (int)tune(GIST_TUNE_SPMM_KERNEL) != 1;
}

}  // namespace gist

extern "C" int gist_in_degree_norm_f32(const int32_t *rowptr, int64_t n_rows, float *norm,
                                       gist_stream_t stream) {
    GIST_REQUIRE(n_rows >= 0, "gist_in_degree_norm_f32: n_rows < 0");
    if (n_rows == 0) return GIST_OK;
    GIST_REQUIRE(rowptr && norm, "gist_in_degree_norm_f32: null pointer");
    hipLaunchKernelGGL(gist::in_degree_norm_kernel, dim3((unsigned)gist::ceil_div(n_rows, 256)),
                       dim3(256), 0, gist::as_stream(stream), rowptr, n_rows, norm);
    return gist::launch_status("gist_in_degree_norm_f32");
}

namespace gist {
static int spmm_generic(const int32_t *rowptr, const int32_t *col, const float *x, int64_t ldx, float *y,
                        int64_t ldy, int64_t n_rows, int64_t d, const float *out_scale,
                        const float *src_scale, int accumulate, hipStream_t st, const SpmmDrop *drop) {
    if (d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && aligned16(x) && aligned16(y))
        return launch_spmm<4>(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale,
                              accumulate, st, drop);
    // d = 4q + 2 wide enough for the workgroup-per-row kernel, 16-byte source rows with room
    // for the last lane's over-read, 8-byte destination: gather 16 bytes per lane anyway
    if (d % 4 == 2 && d >= 4 * kWave && ldx % 4 == 0 && ldx >= d + 2 && aligned16(x) && ldy % 2 == 0 &&
        aligned8(y))
        return launch_spmm<4, true>(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale,
                                    accumulate, st, drop);
    if (d % 2 == 0 && ldx % 2 == 0 && ldy % 2 == 0 && aligned8(x) && aligned8(y))
        return launch_spmm<2>(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale,
                              accumulate, st, drop);
    return launch_spmm<1>(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale,
                          accumulate, st, drop);
}

static bool lds2_takes(int64_t d, int64_t ldx, int64_t ldy, const float *x, const float *y) {
    return d >= 128 && d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && aligned16(x) && aligned16(y);
}

// Which calls can carry the dropout mask themselves: forward (mode 1) every kernel; backward (mode 2) the
// LDS-staged kernel only (it reads every source element of a block once; the row-split kernels would hash every
// gathered element, and on the matrix-core kernel the hashes in the tile conversion cost more than the pass:
// spmm_mfma.hip).
static bool mfma_takes(int64_t d, int64_t ldx, int64_t ldy, const float *x, const float *y,
                       const int32_t *row_blocks) {
    const int forced = (int)tune(GIST_TUNE_SPMM_KERNEL);
    return row_blocks != nullptr && lds2_takes(d, ldx, ldy, x, y) && forced != 1 && (d >= 1536 || forced == 2);
}
bool spmm_drop_takes(int mode, int64_t d, int64_t ldx, int64_t ldy, const float *x, const float *y,
                     const int32_t *row_blocks) {
    if (mode == 1) return true;
    if (mode == 2) return row_blocks != nullptr && lds2_takes(d, ldx, ldy, x, y) && !mfma_takes(d, ldx, ldy, x, y, row_blocks);
    return false;
}

bool spmm_lnb_takes(int64_t d, int64_t ldx, int64_t ldy, const float *x, const float *y, const int32_t *row_blocks,
                    const void *prepared) {
    if (d > 256 || !spmm_drop_takes(2, d, ldx, ldy, x, y, row_blocks)) return false;
    return !(prepared != nullptr && aligned16(prepared) && spmm_dense32_takes(d, ldx, ldy));
}
int64_t spmm_lnb_units(int64_t n_row_blocks) { return n_row_blocks * l2_split_for(n_row_blocks, 1); }

int spmm_drop(const int32_t *rowptr, const int32_t *col, const float *x, int64_t ldx, float *y, int64_t ldy,
              int64_t n_rows, int64_t d, const float *out_scale, const float *src_scale, int accumulate,
              const int32_t *row_blocks, int64_t n_row_blocks, const SpmmDrop &dr, hipStream_t st,
              const void *prepared, const SpmmLnBwd *ln) {
    GIST_REQUIRE(n_rows >= 0 && d >= 0, "gist_spmm_csr_drop_f32: negative size");
    if (n_rows == 0 || d == 0) return GIST_OK;
    GIST_REQUIRE(rowptr && x && y, "gist_spmm_csr_drop_f32: null pointer");
    GIST_REQUIRE(ldx >= d && ldy >= d, "gist_spmm_csr_drop_f32: leading dimension < d");
    GIST_REQUIRE(n_rows < (1LL << 31) && d < (1LL << 31), "gist_spmm_csr_drop_f32: size >= 2^31");
    GIST_REQUIRE(dr.mode == 1 || dr.mode == 2, "gist_spmm_csr_drop_f32: mode must be 1 or 2");
    GIST_REQUIRE(dr.p >= 0.f && dr.p < 1.f && dr.ld >= d, "gist_spmm_csr_drop_f32: bad mask description");
    GIST_REQUIRE(spmm_drop_takes(dr.mode, d, ldx, ldy, x, y, row_blocks),
                 "gist_spmm_csr_drop_f32: this shape cannot carry the mask (use gist_dropout_f32)");
    if (ln != nullptr) {
        GIST_REQUIRE(dr.mode == 2 && dr.p > 0.f && spmm_lnb_takes(d, ldx, ldy, x, y, row_blocks, prepared) && ln->yhat &&
                         ln->dy && ln->col_partials && ln->ldy >= d && ln->lddy >= d && ln->ldy % 4 == 0 &&
                         ln->lddy % 4 == 0 && aligned16(ln->yhat) && aligned16(ln->dy) && aligned16(ln->col_partials),
                     "gist_spmm_csr_drop_f32: this call cannot carry the LayerNorm backward");
        return launch_spmm_lds2(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale, accumulate, row_blocks,
                                n_row_blocks, st, &dr, ln);
    }
    if (prepared != nullptr && aligned16(prepared) && spmm_dense32_takes(d, ldx, ldy))
        return launch_spmm_dense32(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale, accumulate,
                                   row_blocks, n_row_blocks, prepared, st, &dr);
    if (mfma_takes(d, ldx, ldy, x, y, row_blocks))
        return launch_spmm_mfma(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale, accumulate,
                                row_blocks, n_row_blocks, prepared, st, &dr);
    if (row_blocks != nullptr && lds2_takes(d, ldx, ldy, x, y))
        return launch_spmm_lds2(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale, accumulate,
                                row_blocks, n_row_blocks, st, &dr);
    return spmm_generic(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale, accumulate, st, &dr);
}
}  // namespace gist

extern "C" int gist_spmm_csr_f32(const int32_t *rowptr, const int32_t *col, const float *x,
                                 int64_t ldx, float *y, int64_t ldy, int64_t n_rows, int64_t d,
                                 const float *out_scale, const float *src_scale, int accumulate,
                                 gist_stream_t stream) {
    using namespace gist;
    GIST_REQUIRE(n_rows >= 0 && d >= 0, "gist_spmm_csr_f32: negative size");
    if (n_rows == 0 || d == 0) return GIST_OK;
    GIST_REQUIRE(rowptr && x && y, "gist_spmm_csr_f32: null pointer");
    GIST_REQUIRE(ldx >= d && ldy >= d, "gist_spmm_csr_f32: leading dimension < d");
    GIST_REQUIRE(n_rows < (1LL << 31) && d < (1LL << 31), "gist_spmm_csr_f32: size >= 2^31");
    return spmm_generic(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale, accumulate,
                        as_stream(stream), nullptr);
}

/* Aggregation with gist_dropout_f32's mask folded in (see SpmmDrop); row_blocks = NULL: no locality blocks. */
extern "C" int gist_spmm_csr_drop_f32(const int32_t *rowptr, const int32_t *col, const float *x, int64_t ldx,
                                      float *y, int64_t ldy, int64_t n_rows, int64_t d, const float *out_scale,
                                      const float *src_scale, int accumulate, const int32_t *row_blocks,
                                      int64_t n_row_blocks, int mode, float p, uint64_t seed, uint64_t y_offset,
                                      uint64_t src_offset, int64_t mask_ld, gist_stream_t stream) {
    gist::SpmmDrop dr{};
    dr.mode = mode; dr.p = p; dr.scale = (p > 0.f && p < 1.f) ? 1.0f / (1.0f - p) : 1.f;
    dr.sm = seed * 0x9E3779B97F4A7C15ULL; dr.y_base = y_offset; dr.src_base = src_offset; dr.ld = mask_ld;
    return gist::spmm_drop(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale, accumulate, row_blocks,
                           n_row_blocks, dr, gist::as_stream(stream));
}

extern "C" int gist_spmm_csr_drop_lnbwd_f32(const int32_t *rowptr, const int32_t *col, const float *x, int64_t ldx,
                                            const float *y, int64_t ldy, int64_t n_rows, int64_t d, const float *src_scale,
                                            const int32_t *row_blocks, int64_t n_row_blocks, float p, uint64_t seed,
                                            uint64_t y_offset, uint64_t src_offset, int64_t mask_ld, const float *yhat,
                                            int64_t ldyh, const float *rstd, float *dy, int64_t lddy, float *col_partials,
                                            int64_t partial_rows, int relu, gist_stream_t stream) {
    using namespace gist;
    GIST_REQUIRE(row_blocks != nullptr && n_row_blocks > 0 && partial_rows >= spmm_lnb_units(n_row_blocks),
                 "gist_spmm_csr_drop_lnbwd_f32: col_partials has fewer than gist_spmm_lnb_units(n_row_blocks) rows");
    SpmmDrop dr{};
    dr.mode = 2; dr.p = p; dr.scale = (p > 0.f && p < 1.f) ? 1.0f / (1.0f - p) : 1.f;
    dr.sm = seed * 0x9E3779B97F4A7C15ULL; dr.y_base = y_offset; dr.src_base = src_offset; dr.ld = mask_ld;
    SpmmLnBwd ln{};
    ln.yhat = yhat; ln.ldy = ldyh; ln.rstd = rstd; ln.dy = dy; ln.lddy = lddy; ln.col_partials = col_partials; ln.relu = relu;
    return spmm_drop(rowptr, col, x, ldx, const_cast<float *>(y), ldy, n_rows, d, nullptr, src_scale, 1, row_blocks,
                     n_row_blocks, dr, as_stream(stream), nullptr, &ln);
}

extern "C" int64_t gist_spmm_lnb_units(int64_t n_row_blocks) { return n_row_blocks > 0 ? gist::spmm_lnb_units(n_row_blocks) : 0; }

extern "C" int gist_spmm_csr_drop_prepared_f32(const int32_t *rowptr, const int32_t *col, const float *x, int64_t ldx,
                                               float *y, int64_t ldy, int64_t n_rows, int64_t d,
                                               const float *out_scale, const float *src_scale, int accumulate,
                                               const int32_t *row_blocks, int64_t n_row_blocks, int mode, float p,
                                               uint64_t seed, uint64_t y_offset, uint64_t src_offset,
                                               int64_t mask_ld, const void *prepared, gist_stream_t stream) {
    gist::SpmmDrop dr{};
    dr.mode = mode; dr.p = p; dr.scale = (p > 0.f && p < 1.f) ? 1.0f / (1.0f - p) : 1.f;
    dr.sm = seed * 0x9E3779B97F4A7C15ULL; dr.y_base = y_offset; dr.src_base = src_offset; dr.ld = mask_ld;
    return gist::spmm_drop(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale, accumulate, row_blocks,
                           n_row_blocks, dr, gist::as_stream(stream), prepared);
}

extern "C" int gist_spmm_prepared_useful(int64_t d, int64_t ldx, int64_t ldy, const float *x, const float *y) {
    // (the fp32 block-dense kernel rides along when the structure exists; alone it does not pay for the 11-us
    // prepare launch)
    return gist::spmm_prepared_takes(d, ldx, ldy, x, y) ? 1 : 0;
}

extern "C" int gist_spmm_drop_takes(int mode, int64_t d, int64_t ldx, int64_t ldy, const float *x, const float *y,
                                    int has_row_blocks) {
    static const int32_t dummy = 0;
    return gist::spmm_drop_takes(mode, d, ldx, ldy, x, y, has_row_blocks ? &dummy : nullptr) ? 1 : 0;
}

extern "C" int gist_spmm_csr_blocked_f32(const int32_t *rowptr, const int32_t *col, const float *x,
                                         int64_t ldx, float *y, int64_t ldy, int64_t n_rows,
                                         int64_t d, const float *out_scale,
                                         const float *src_scale, int accumulate,
                                         const int32_t *row_blocks, int64_t n_row_blocks,
                                         gist_stream_t stream) {
    using namespace gist;
    GIST_REQUIRE(n_rows >= 0 && d >= 0, "gist_spmm_csr_blocked_f32: negative size");
    if (n_rows == 0 || d == 0) return GIST_OK;
    GIST_REQUIRE(rowptr && x && y, "gist_spmm_csr_blocked_f32: null pointer");
    GIST_REQUIRE(ldx >= d && ldy >= d, "gist_spmm_csr_blocked_f32: leading dimension < d");
    GIST_REQUIRE(n_rows < (1LL << 31) && d < (1LL << 31), "gist_spmm_csr_blocked_f32: size >= 2^31");
    GIST_REQUIRE(row_blocks == nullptr || n_row_blocks > 0,
                 "gist_spmm_csr_blocked_f32: row_blocks given but n_row_blocks <= 0");
    // narrow rows keep every lane busy only in the lane-group kernel
    if (d < 128)
        return gist_spmm_csr_f32(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale,
                                 accumulate, stream);
    hipStream_t st = as_stream(stream);
    if (d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && aligned16(x) && aligned16(y)) {
        // wide rows: the block-dense matrix-core kernel (spmm_mfma.hip), whose per-workgroup set-up
        // (the block's counts, ~11 us) pays from two 128-column tiles per workgroup on -- measured on
        // the Reddit-like batch: D = 4096 38 vs 57 us, 2048 29 vs 33, 1024 21 vs 16, 512 20 vs 12.
        // Tuning hook: 1 = always the LDS gather kernel, 2 = always the matrix-core kernel.
        const int want = (int)tune(GIST_TUNE_SPMM_KERNEL);
        // (uniform 128-row blocks -- row_blocks = NULL -- cut across the parts: too many neighbours fall
        // outside a block for the dense product, the gather kernel degrades more gracefully there)
        if (want == 2 || (want != 1 && d >= 1536 && row_blocks != nullptr))
            return launch_spmm_mfma(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale,
                                    accumulate, row_blocks, n_row_blocks, nullptr, st);
        return launch_spmm_lds2(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale,
                                accumulate, row_blocks, n_row_blocks, st);
    }
    // other widths / alignments (the layer-0 aggregation of F = 602 features): the row-split kernel
    return gist_spmm_csr_f32(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale,
                             accumulate, stream);
}

extern "C" int64_t gist_spmm_blocks_bytes(int64_t n_row_blocks) { return gist::spmm_blocks_bytes(n_row_blocks); }

extern "C" int gist_spmm_blocks_prepare(const int32_t *rowptr, const int32_t *col, int64_t n_rows,
                                        const int32_t *row_blocks, int64_t n_row_blocks, void *prepared,
                                        int64_t prepared_bytes, gist_stream_t stream) {
    using namespace gist;
    GIST_REQUIRE(n_rows >= 0, "gist_spmm_blocks_prepare: negative size");
    if (n_rows == 0) return GIST_OK;
    GIST_REQUIRE(rowptr && prepared, "gist_spmm_blocks_prepare: null pointer");
    GIST_REQUIRE(n_rows < (1LL << 31), "gist_spmm_blocks_prepare: size >= 2^31");
    GIST_REQUIRE(row_blocks == nullptr || n_row_blocks > 0,
                 "gist_spmm_blocks_prepare: row_blocks given but n_row_blocks <= 0");
    const int64_t nb = row_blocks ? n_row_blocks : ceil_div(n_rows, 128);
    GIST_REQUIRE(aligned16(prepared) && prepared_bytes >= spmm_blocks_bytes(nb),
                 "gist_spmm_blocks_prepare: buffer too small or not 16-byte aligned");
    return launch_spmm_blocks_prepare(rowptr, col, nullptr, nullptr, n_rows, row_blocks, n_row_blocks, prepared,
                                      nullptr, as_stream(stream));
}

extern "C" int gist_spmm_csr_prepared_f32(const int32_t *rowptr, const int32_t *col, const float *x,
                                          int64_t ldx, float *y, int64_t ldy, int64_t n_rows, int64_t d,
                                          const float *out_scale, const float *src_scale, int accumulate,
                                          const int32_t *row_blocks, int64_t n_row_blocks,
                                          const void *prepared, gist_stream_t stream) {
    using namespace gist;
    GIST_REQUIRE(n_rows >= 0 && d >= 0, "gist_spmm_csr_prepared_f32: negative size");
    if (n_rows == 0 || d == 0) return GIST_OK;
    GIST_REQUIRE(rowptr && x && y && prepared, "gist_spmm_csr_prepared_f32: null pointer");
    GIST_REQUIRE(ldx >= d && ldy >= d, "gist_spmm_csr_prepared_f32: leading dimension < d");
    GIST_REQUIRE(n_rows < (1LL << 31) && d < (1LL << 31), "gist_spmm_csr_prepared_f32: size >= 2^31");
    GIST_REQUIRE(row_blocks == nullptr || n_row_blocks > 0,
                 "gist_spmm_csr_prepared_f32: row_blocks given but n_row_blocks <= 0");
    // below the bf16x3 matrix-core kernel's widths (and for unaligned rows): the fp32 block-dense kernel
    if (aligned16(prepared) && spmm_dense32_takes(d, ldx, ldy))
        return launch_spmm_dense32(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale, accumulate,
                                   row_blocks, n_row_blocks, prepared, as_stream(stream));
    // widths / alignments the matrix-core kernel does not take: the unprepared entry point decides
    if (!spmm_prepared_takes(d, ldx, ldy, x, y) || !aligned16(prepared))
        return gist_spmm_csr_blocked_f32(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale,
                                         accumulate, row_blocks, n_row_blocks, stream);
    return launch_spmm_mfma(rowptr, col, x, ldx, y, ldy, n_rows, d, out_scale, src_scale, accumulate,
                            row_blocks, n_row_blocks, prepared, as_stream(stream));
}
