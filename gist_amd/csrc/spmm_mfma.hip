// Block-dense aggregation on the matrix cores: gist_spmm_csr_blocked_f32 for wide rows.
//
// A cluster batch is a union of METIS parts, and inside a part the adjacency is close to DENSE: on the
// Reddit-like batch of the metric (2046 rows = 20 parts of ~102 rows, 63 in-batch neighbours per row)
// 99.1 % of the edges stay inside the row's own part, i.e. the part's diagonal block of A holds
// 63 / 102 = 62 % non-zeros.  Gathering those neighbours one by one (the LDS-staged kernel in
// spmm.hip) moves nnz x D x 4 bytes through the LDS read port and is bound by its latency (51 us at
// D = 4096, 0.14 of the HBM roofline).  Here the diagonal block is what it looks like, a small dense
// matrix:
//     Y_p[rows, D] = C_p[rows, rows] . X_p[rows, D]  +  (the few neighbours outside the part)
// with C_p the block's edge COUNTS (a multigraph: an edge listed twice counts twice), on
// v_mfma_f32_16x16x32_bf16.  It is exact fp32 aggregation, not a reduced-precision one: a count
// <= 256 is exact in bf16, every fp32 x is carried as three bf16 pieces x = x1 + x2 + x3 (3 x 8 =
// all 24 significant bits, fp32's exponent range), each count x piece product is exact, and the
// accumulation is fp32 -- the same arithmetic as summing the neighbours one after the other, in
// another (fixed) order.  Results do not depend on block boundaries or placement.
//
// One 1024-thread workgroup per (part, group of 128-column tiles):
//   once:   the block's counts -> LDS, [k chunk][row][8 k] bf16: one flat pass over the block's edges
//           (all loads of a thread in flight, row of an edge by binary search in the row pointers,
//           LDS atomics on 16-bit counters), converted in place; per row the ids of its neighbours
//           outside the block (up to 8, restored to CSR order; a row with more of them walks its edge list for
//           them in the epilogue, its in-block edges stay in the dense product; a row with a count > 256 or
//           beyond the 128 rows a block may stage is gathered from memory in full instead: oversized blocks)
//   tile:   X_p tile [rows x 128] -> registers (4 x 16-byte loads per thread, issued one tile ahead,
//           branch-free) -> x src_scale -> three bf16 pieces -> LDS as X^T, [k chunk][column & 3]
//           [column >> 2][8 k]: both MFMA operands are then one conflict-free ds_read_b128 per lane
//           8 x 8 MFMA tiles x 4 k steps x 3 pieces, 4 output tiles per wave
//           accumulators -> LDS (fp32, over the X^T image) -> registers, two rows per wave pass ->
//           barrier -> + outside neighbours (global loads, CSR order) -> x out_scale (+ y) -> 512-byte
//           row stores, which run under the next tile's conversion
// The barriers wait for LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads() also
// waits for vmcnt(0), i.e. for the prefetched tile.
//
// Measured (scripts/spmm_mf_probe.py under rocprofv3, scripts/spmm_mf_phases.py; Reddit-like batch,
// 20 blocks x 12 column groups = 240 workgroups at D = 4096): 38 us against 57 us for the LDS gather
// kernel (29 vs 33 at D = 2048; below that the set-up does not pay and the dispatcher keeps the gather
// kernel).  Phases of a workgroup: set-up 11 us (row pointers 1, counts 7, conversion 3), then per tile
// ~8 us = conversion 2.5 (VALU-bound: 16 waves x ~250 instructions) + MFMA 1 + barriers 2 + result
// tile 0.5 + stores 2 (the chip's 256 workgroups store in step: 9 TB/s bursts).  One workgroup per CU
// (136 KiB of LDS) cannot overlap those phases; the HBM floor of the call is 15 us.
#include <type_traits>

#include "common.h"

namespace gist {

typedef __bf16 mf_bf16x8 __attribute__((ext_vector_type(8)));
typedef float mf_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 mf_bf16x2 __attribute__((ext_vector_type(2)));
typedef float mf_f32x2 __attribute__((ext_vector_type(2)));

constexpr int MF_ROWS = 128;                       // rows of a block staged = k extent of the product
constexpr int MF_CT = 128;                         // columns per tile
constexpr int MF_THREADS = 1024;
constexpr int MF_WAVES = MF_THREADS / 64;
// one bf16 piece of X^T: [k chunk 16][column & 3][column >> 2 (32, padded to 36)][8 k] -- a thread
// stages four ADJACENT columns (one 16-byte load per row), and with this order the 32 lanes of a
// half wave write consecutive 16-byte slots while the 16 lanes of a fragment read (columns n .. n+15
// -> slots 36 (n & 3) + (n >> 2)) still hit 16 different bank quads
constexpr int MF_CHUNK_SLOTS = 4 * 36;
constexpr int MF_PIECE = 16 * MF_CHUNK_SLOTS * 16;
constexpr int MF_YT_PITCH = 132;                   // floats; the fp32 result tile aliases the X^T image
#ifndef MF_REM_N      // (dev A/B: -DMF_REM_N=...)
#define MF_REM_N 8
#endif
constexpr int MF_REM = MF_REM_N;                   // outside neighbours listed per row
constexpr int MF_A_OFF = 3 * MF_PIECE;             // counts, [k chunk 16][row 128][8 k] bf16
constexpr int MF_RP_OFF = MF_A_OFF + 16 * MF_ROWS * 16;          // int rowptr[132]
constexpr int MF_REMC_OFF = MF_RP_OFF + 132 * 4;                 // int rem_cnt[128]  (-1: gather the whole row)
constexpr int MF_REM_OFF = MF_REMC_OFF + MF_ROWS * 4;            // int rem_col[128][8]
constexpr int MF_REME_OFF = MF_REM_OFF + MF_ROWS * MF_REM * 4;   // their edge indices (to restore CSR order)
constexpr int MF_SC_OFF = MF_REME_OFF + MF_ROWS * MF_REM * 4;    // float out_scale[128]
constexpr int MF_BIG_OFF = MF_SC_OFF + MF_ROWS * 4;              // int big_row[128]: bit 0 a count > 256, bit 1 edges in a pair image
constexpr int MF_PINFO_OFF = MF_BIG_OFF + MF_ROWS * 4;           // int pair[2][2] = (first source row, source rows) of the block's pairs
constexpr int MF_LDS_BYTES = MF_PINFO_OFF + 16;
static_assert(MF_ROWS * MF_YT_PITCH * 4 <= 3 * MF_PIECE, "result tile fits the X^T image");
static_assert(MF_LDS_BYTES <= 160 * 1024, "LDS budget of one CU");
// Sibling parts (round 6).  Two parts of ONE community in a batch: every row of either has 30-80 neighbours in the other,
// far more than the per-row list holds, and walking those rows' edge lists cost 125-140 us per launch
// (profiles/r05_unplanted_graph.json).  The off-diagonal block (rows of p, sources of q) is dense like a diagonal one, so
// it is multiplied like one: the prepare kernel counts a block's outside edges per OTHER block of the batch, and up to
// MF_PAIRS blocks with >= MF_PAIR_MIN of them get their own count image; the aggregation kernel runs them as further k
// steps of the block's product (the accumulators stay in registers; per pair and tile one more X tile conversion).  Their
// edges are in no per-row list.
constexpr int MF_PAIRS = 2;
constexpr int MF_IMG_BYTES = 16 * MF_ROWS * 16;                  // one count image
#ifndef MF_PAIR_MIN_N      // (dev A/B: unplanted H = 4096 step, aggregation launch average: 256 49.9 us, 128 50.0, 48 53.0; 128 for the tail: fewer 70-136-us launches)
#define MF_PAIR_MIN_N 128
#endif
constexpr int MF_PAIR_MIN = MF_PAIR_MIN_N;                       // edges into the other block (gist_spmm_pair_min_edges)
#ifndef MF_FINE_TILES      // column tiles per workgroup of a block with pairs (dev A/B: 2 tiles 53.5 us against 50.0)
#define MF_FINE_TILES 1
#endif
constexpr int MF_PAIR_BLOCKS = 256;                              // batches of up to this many blocks look for pairs
// (prepare kernel only, in the idle X^T region: the pair images, the outside-edge histogram, the row blocks)
constexpr int MF_PHIST_OFF = MF_PAIRS * MF_IMG_BYTES;
constexpr int MF_PRB_OFF = MF_PHIST_OFF + MF_PAIR_BLOCKS * 4;
static_assert(MF_PRB_OFF + (MF_PAIR_BLOCKS + 1) * 4 <= 3 * MF_PIECE, "prepare-time scratch fits the X^T region");
// a prepared block in memory: the counts image, then rem_cnt[128], then rem_col[128][8] (as in LDS), then pair[2][2].
// rem_cnt[r] >= 0: bits 0-7 = listed outside neighbours, bit 8 = the row has edges in a pair image (a consumer without
// the pair images gathers such a row in full); -1: gather the row in full; -2: walk the edge list for the neighbours
// outside the block and its pairs.  The pair images of all blocks follow the block records (spmm_blocks_bytes).
constexpr int MF_PREP_STRIDE = MF_IMG_BYTES + MF_ROWS * 4 + MF_ROWS * MF_REM * 4 + 16;
constexpr int MF_PREP_PINFO = MF_IMG_BYTES + MF_ROWS * 4 + MF_ROWS * MF_REM * 4;
static_assert(MF_REM_OFF == MF_REMC_OFF + MF_ROWS * 4 && MF_PREP_STRIDE % 16 == 0, "rem_cnt and rem_col are contiguous");

struct MfArgs {
    const int32_t *rowptr, *col;
    const float *x; int64_t ldx;
    float *y; int64_t ldy;
    int n_rows, d;
    const float *out_scale, *src_scale;
    int accumulate;
    const int32_t *row_blocks;
    int n_blocks, n_col_tiles, groups;
    const unsigned char *prep;       // prepared blocks (spmm_blocks_prepare_kernel) or NULL
    const int32_t *units;            // NULL, or per workgroup unit (r0, r1, xs0, xs1): output rows [r0, r1), SOURCE rows
                                     // [xs0, xs1) of x -- an off-diagonal block pair of a part-ordered graph (prepared
                                     // counts, no outside neighbours: gist_spmm_block_units_f32)
    SpmmDrop dr;                     // dropout masks folded in (kernel template DROP = dr.mode)
    int pairs;                       // prepared blocks: the structure may hold pairs (prepare: look for them)
};

__device__ __forceinline__ uint32_t mf_pack(__bf16 lo, __bf16 hi) {
    return (uint32_t)__builtin_bit_cast(unsigned short, lo) |
           ((uint32_t)__builtin_bit_cast(unsigned short, hi) << 16);
}

// Workgroup barrier that waits for this wave's LDS traffic only: the global loads of the next tile stay
// in flight across it (__syncthreads() waits for vmcnt(0) as well, i.e. for every prefetch).
__device__ __forceinline__ void mf_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One 32-KiB count image, memory -> LDS without passing through registers: 16 waves x 2 LDS-DMA instructions of 1 KiB
// (the caller waits for vmcnt(0) before the barrier that publishes it)
__device__ __forceinline__ void mf_image_to_lds(const unsigned char *img, unsigned char *dst, int wave, int lane) {
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(img), 0, MF_IMG_BYTES, 0x00020000);
#pragma unroll
    for (int h = 0; h < 2; ++h)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(dst + (2 * wave + h) * 1024),
                                                 16, (uint32_t)((2 * wave + h) * 1024 + lane * 16), 0, 0, 0);
}

// v += sum over the listed source rows (ids in lanes 0 .. cnt-1 of `ids`) of scale . x[id][gc .. gc+3],
// in list order, for the lanes with `mine`; 4 row reads in flight
__device__ __forceinline__ void mf_gather(const float *x, int64_t ldx, const float *src_scale, int gc,
                                          bool mine, int ids, int cnt, float4 &v) {
    for (int j0 = 0; j0 < cnt; j0 += 4) {
        float4 rv[4];
        float rs[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool on = j0 + t < cnt;                          // wave-uniform
            const int g = __builtin_amdgcn_readlane(ids, (j0 + t) & 63);
            rs[t] = 0.f;
            rv[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on) {
                rs[t] = src_scale ? src_scale[g] : 1.f;
                if (mine) rv[t] = *reinterpret_cast<const float4 *>(x + (int64_t)g * ldx + gc);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            v.x = fmaf(rs[t], rv[t].x, v.x); v.y = fmaf(rs[t], rv[t].y, v.y);
            v.z = fmaf(rs[t], rv[t].z, v.z); v.w = fmaf(rs[t], rv[t].w, v.w);
        }
    }
}

// every neighbour of one row, CSR order
__device__ __forceinline__ void mf_gather_row(const MfArgs &a, int e0, int e1, int lane, int gc, bool mine,
                                              float4 &v) {
    for (int base = e0; base < e1; base += 64) {
        const int ids = base + lane < e1 ? a.col[base + lane] : 0;
        mf_gather(a.x, a.ldx, a.src_scale, gc, mine, ids, min(64, e1 - base), v);
    }
}

// the neighbours of one row OUTSIDE the block [r0, r0 + nloc) and its pairs' source rows, CSR order (a row whose outside
// neighbours do not fit the per-row list: its in-block and in-pair edges stay in the dense products)
template <bool PP>
__device__ __forceinline__ void mf_gather_row_outside(const MfArgs &a, int e0, int e1, int r0, int nloc, int4 pairs,
                                                      int lane, int gc, bool mine, float4 &v) {
    for (int base = e0; base < e1; base += 64) {
        const bool in = base + lane < e1;
        const int ids = in ? a.col[base + lane] : 0;
        // (unsigned compares: inside [s, s + n) <=> (unsigned)(id - s) < n; an absent pair has n = 0)
        unsigned long long m = PP ? __ballot(in && (unsigned)(ids - r0) >= (unsigned)nloc &&
                                             (unsigned)(ids - pairs.x) >= (unsigned)pairs.y &&
                                             (unsigned)(ids - pairs.z) >= (unsigned)pairs.w)
                                  : __ballot(in && (ids < r0 || ids >= r0 + nloc));
        while (m) {                                        // four row reads in flight, lowest lanes (= edge order) first
            float4 rv[4];
            float rs[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool on = m != 0ULL;                 // wave-uniform
                const int l = on ? __builtin_ctzll(m) : 0;
                if (on) m &= m - 1ULL;
                const int g = __builtin_amdgcn_readlane(ids, l);
                rs[t] = 0.f;
                rv[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (on) {
                    rs[t] = a.src_scale ? a.src_scale[g] : 1.f;
                    if (mine) rv[t] = *reinterpret_cast<const float4 *>(a.x + (int64_t)g * a.ldx + gc);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                v.x = fmaf(rs[t], rv[t].x, v.x); v.y = fmaf(rs[t], rv[t].y, v.y);
                v.z = fmaf(rs[t], rv[t].z, v.z); v.w = fmaf(rs[t], rv[t].w, v.w);
            }
        }
    }
}

#ifdef MF_PROBE      // dev build (scripts/spmm_mf_phases.py): s_memrealtime (100 MHz) stamps of workgroup 0, wave 0
__device__ unsigned long long g_mf_probe[64];
#ifdef MF_PROBE_PAIRS   // ... of the first workgroup of the blocks-with-pairs class instead (block 0 must have a pair)
#define MF_STAMP_B(i) do { } while (0)
#define MF_STAMP(i) do { if (PAIRK && bid == 0 && threadIdx.x == 0) g_mf_probe[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MF_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_mf_probe[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define MF_STAMP_B(i) MF_STAMP(i)
#endif
#else
#define MF_STAMP(i) do { } while (0)
#define MF_STAMP_B(i) do { } while (0)
#endif

// Edge counts of block [r0, r0 + nloc) as bf16 in `ab`, [k chunk][row][8 k], and per row its neighbours
// outside the block (rem_cnt: how many, -2 = more than the list holds: the epilogue walks the row's edges for them,
// -1 = the row leaves the dense product and is gathered in full;
// rem_col: their ids in CSR order).  All 1024 threads; `after_ids` is called once, right after the first
// batch of id loads has been issued.
// PAIRS (the prepare kernel, batches of <= MF_PAIR_BLOCKS blocks): block `rbk`'s outside edges are counted per other
// block first; the (up to MF_PAIRS) blocks holding >= MF_PAIR_MIN of them become pairs -- (first source row, rows) at
// MF_PINFO_OFF, their counts as images at the start of the X^T region, their edges in no per-row list.
template <bool PAIRS, typename F>
__device__ __forceinline__ void mf_build_block(const MfArgs &a, int rbk, int r0, int nloc, unsigned char *mf_smem,
                                               F after_ids) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *ab = mf_smem + MF_A_OFF;
    uint32_t *a32 = reinterpret_cast<uint32_t *>(ab);
    int32_t *rp = reinterpret_cast<int32_t *>(mf_smem + MF_RP_OFF);
    int32_t *rem_cnt = reinterpret_cast<int32_t *>(mf_smem + MF_REMC_OFF);
    int32_t *rem_col = reinterpret_cast<int32_t *>(mf_smem + MF_REM_OFF);
    int32_t *rem_e = reinterpret_cast<int32_t *>(mf_smem + MF_REME_OFF);
    float *sc = reinterpret_cast<float *>(mf_smem + MF_SC_OFF);
    int32_t *big_row = reinterpret_cast<int32_t *>(mf_smem + MF_BIG_OFF);
    int32_t *pinfo = reinterpret_cast<int32_t *>(mf_smem + MF_PINFO_OFF);
    uint32_t *p32 = reinterpret_cast<uint32_t *>(mf_smem);                       // pair images (PAIRS)
    int32_t *hist = reinterpret_cast<int32_t *>(mf_smem + MF_PHIST_OFF);
    int32_t *rbl = reinterpret_cast<int32_t *>(mf_smem + MF_PRB_OFF);
    // ---- once per workgroup: the block's edge counts and its outside neighbours ----
    {
        uint4 *z = reinterpret_cast<uint4 *>(ab + tid * 32);
        z[0] = make_uint4(0, 0, 0, 0);
        z[1] = make_uint4(0, 0, 0, 0);
        if (tid <= nloc) rp[tid] = a.rowptr[r0 + tid];
        if (tid < MF_ROWS) {
            sc[tid] = (tid < nloc && a.out_scale) ? a.out_scale[r0 + tid] : 1.f;
            rem_cnt[tid] = 0;
            big_row[tid] = 0;
        }
        if (tid < 4) pinfo[tid] = 0;
        if constexpr (PAIRS) {
            uint4 *zp = reinterpret_cast<uint4 *>(mf_smem + tid * (MF_PAIRS * MF_IMG_BYTES / MF_THREADS));
#pragma unroll
            for (int i = 0; i < MF_PAIRS * MF_IMG_BYTES / MF_THREADS / 16; ++i) zp[i] = make_uint4(0, 0, 0, 0);
            if (tid < a.n_blocks) hist[tid] = 0;
            if (tid <= a.n_blocks) rbl[tid] = min(a.row_blocks[tid], a.n_rows);
        }
    }
    mf_barrier();
    MF_STAMP_B(1);
    if constexpr (PAIRS) {
        // outside edges per other block of the batch (the block of a source row by binary search in the row blocks)
        const int E0 = rp[0], E1 = rp[nloc];
        for (int base = E0; base < E1; base += 8 * MF_THREADS) {
            int c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) c[u] = a.col[min(base + u * MF_THREADS + tid, E1 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (base + u * MF_THREADS + tid >= E1 || (unsigned)(c[u] - r0) < (unsigned)nloc) continue;
                int lo = 0, hi = a.n_blocks;                // rbl[lo] <= c < rbl[hi]
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (rbl[mid] <= c[u]) lo = mid; else hi = mid;
                }
                atomicAdd(&hist[lo], 1);
            }
        }
        mf_barrier();
        if (wave == 0) {
            // the MF_PAIRS fullest blocks (ties: the lower block), each with >= MF_PAIR_MIN edges and no more rows than
            // an image has k
            int taken = -1;
            for (int j = 0; j < MF_PAIRS; ++j) {
                int best_c = MF_PAIR_MIN - 1, best_b = -1;
                for (int b = lane; b < a.n_blocks; b += 64) {
                    const int cb = hist[b];
                    if (b != rbk && b != taken && rbl[b + 1] - rbl[b] <= MF_ROWS && cb > best_c) { best_c = cb; best_b = b; }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const int oc = __shfl_xor(best_c, off), ob = __shfl_xor(best_b, off);
                    if (oc > best_c || (oc == best_c && ob >= 0 && (best_b < 0 || ob < best_b))) { best_c = oc; best_b = ob; }
                }
                if (best_b < 0) break;                       // (wave-uniform after the butterfly)
                if (lane == 0) { pinfo[2 * j] = rbl[best_b]; pinfo[2 * j + 1] = rbl[best_b + 1] - rbl[best_b]; }
                taken = best_b;
            }
        }
        mf_barrier();
    }

    {   // every edge of the block, 1024 at a time, the loads of a thread in flight together (and BEFORE
        // the first X tile's: loads return in order); the row of an edge by binary search in the block's
        // row pointers (LDS)
        const int E0 = rp[0], E1 = rp[nloc];
        bool first = true;
        for (int base = E0; base < E1 || first; base += 8 * MF_THREADS) {
            int c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) c[u] = -1;
            if (E1 > E0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) c[u] = a.col[min(base + u * MF_THREADS + tid, E1 - 1)];
            }
            if (first) {
                first = false;
                after_ids();        // the caller's own loads go out behind the ids (loads return in order)
            }
            // the rows of the thread's eight edges: LDS only, done while the ids are in flight
            int er[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * MF_THREADS + tid;
                int lo = 0, hi = nloc;                     // rp[lo] <= e < rp[hi]
#pragma unroll
                for (int it = 0; it < 7; ++it) {
                    const int mid = (lo + hi) >> 1;
                    const bool up = mid > lo && rp[mid] <= e;
                    hi = (!up && mid > lo) ? mid : hi;
                    lo = up ? mid : lo;
                }
                er[u] = lo;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * MF_THREADS + tid;
                if (e >= E1) continue;
                const int r = er[u];
                const int k = c[u] - r0;
                if (k >= 0 && k < nloc) {
                    atomicAdd(&a32[((k >> 3) * MF_ROWS + r) * 4 + ((k & 7) >> 1)], (k & 1) ? 0x10000u : 1u);
                    continue;
                }
                if constexpr (PAIRS) {
                    bool paired = false;
#pragma unroll
                    for (int j = 0; j < MF_PAIRS; ++j) {
                        const int kq = c[u] - pinfo[2 * j];
                        if ((unsigned)kq < (unsigned)pinfo[2 * j + 1]) {
                            atomicAdd(&p32[j * (MF_IMG_BYTES / 4) + ((kq >> 3) * MF_ROWS + r) * 4 + ((kq & 7) >> 1)],
                                      (kq & 1) ? 0x10000u : 1u);
                            paired = true;
                        }
                    }
                    if (paired) { atomicOr(&big_row[r], 2); continue; }
                }
                const int slot = atomicAdd(&rem_cnt[r], 1);
                if (slot < MF_REM) { rem_e[r * MF_REM + slot] = e; rem_col[r * MF_REM + slot] = c[u]; }
            }
        }
    }
    mf_barrier();
    MF_STAMP_B(2);
    // outside neighbours of a row back into CSR order (the slots were taken in arrival order); a row
    // with more than fit is gathered in full below
    if (tid < nloc) {
        const int cnt = rem_cnt[tid];
        if (rp[tid + 1] - rp[tid] > 65535) {       // (65536 copies of an edge would wrap a counter)
            rem_cnt[tid] = -1;
        } else if (cnt > MF_REM) {
            // (round 5) more outside neighbours than the list holds -- a row of a community cut into two parts of one batch
            // has 30-80 of them: its in-block edges STAY in the dense product, the epilogue walks its edge list for the
            // outside ones only (-1, the whole row gathered instead, cost 130-270 us per launch on every eighth batch of
            // the power-law community graph: profiles/r05_unplanted_graph.json)
            rem_cnt[tid] = -2;
        } else {
            for (int i = 1; i < cnt; ++i) {
                const int e = rem_e[tid * MF_REM + i], c = rem_col[tid * MF_REM + i];
                int j = i - 1;
                while (j >= 0 && rem_e[tid * MF_REM + j] > e) {
                    rem_e[tid * MF_REM + j + 1] = rem_e[tid * MF_REM + j];
                    rem_col[tid * MF_REM + j + 1] = rem_col[tid * MF_REM + j];
                    --j;
                }
                rem_e[tid * MF_REM + j + 1] = e;
                rem_col[tid * MF_REM + j + 1] = c;
            }
        }
    }
    constexpr int RW = MF_ROWS / MF_WAVES;                 // rows a wave converts: wave + 16 i
    // counts -> bf16 in place; a row with a count > 256 (not exact in bf16) leaves the dense product
    // as well (its part of the result tile is ignored) and is gathered in full
    {
        uint32_t v[RW];
#pragma unroll
        for (int i = 0; i < RW; ++i) v[i] = a32[((lane >> 2) * MF_ROWS + wave + MF_WAVES * i) * 4 + (lane & 3)];
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const uint32_t c0 = v[i] & 0xffffu, c1 = v[i] >> 16;
            const bool big = __ballot(c0 > 256u || c1 > 256u) != 0ULL;
            if (big && lane == 0) atomicOr(&big_row[wave + MF_WAVES * i], 1);
            a32[((lane >> 2) * MF_ROWS + wave + MF_WAVES * i) * 4 + (lane & 3)] =
                mf_pack((__bf16)(float)c0, (__bf16)(float)c1);
        }
    }
    if constexpr (PAIRS) {
        for (int j = 0; j < MF_PAIRS; ++j) {
            if (pinfo[2 * j + 1] <= 0) break;
            uint32_t *q32 = p32 + j * (MF_IMG_BYTES / 4);
            uint32_t v[RW];
#pragma unroll
            for (int i = 0; i < RW; ++i) v[i] = q32[((lane >> 2) * MF_ROWS + wave + MF_WAVES * i) * 4 + (lane & 3)];
#pragma unroll
            for (int i = 0; i < RW; ++i) {
                const uint32_t c0 = v[i] & 0xffffu, c1 = v[i] >> 16;
                const bool big = __ballot(c0 > 256u || c1 > 256u) != 0ULL;
                if (big && lane == 0) atomicOr(&big_row[wave + MF_WAVES * i], 1);
                q32[((lane >> 2) * MF_ROWS + wave + MF_WAVES * i) * 4 + (lane & 3)] =
                    mf_pack((__bf16)(float)c0, (__bf16)(float)c1);
            }
        }
    }
    mf_barrier();
    // final state of a row in one place: -1 = gathered in full; bit 8 = edges in a pair image
    if (tid < nloc) {
        if (big_row[tid] & 1) rem_cnt[tid] = -1;
        else if (rem_cnt[tid] >= 0 && (big_row[tid] & 2)) rem_cnt[tid] |= 0x100;
    }
    mf_barrier();
}

// DROP = 1 (gist_spmm_csr_drop_f32, forward form): what is stored is multiplied by y's dropout mask (+4 us per
// D = 4096 launch for the 22-us pass it replaces).  The backward form (x read through its mask in the tile
// conversion, the old y through y's) was built and measured on this kernel: bit-identical, 103 us against
// 35 + 22 -- eight more 64-bit hashes per thread in the vector-bound conversion and 144 B per lane of spills at
// 128 registers; it stays a separate pass in front of this kernel.
// PAIRK = false: spmm_csr_mfma_kernel, every block that has no pairs; PAIRK = true: spmm_csr_mfma_pairs_kernel, launched
// behind it over the same grid when the batch's structure was prepared with pairs allowed -- the blocks WITH pairs (both
// kernels read the block's pair descriptor first; a block belongs to exactly one of them).  Two kernels because the pair
// units cost registers this code does not have (128 at 1024 threads): in one kernel, even behind a branch that is never
// taken, they moved the register allocation of the common path -- 49 us per D = 4096 launch on a batch WITHOUT pairs
// against 36 (88 spilled registers against 3).
// PP = false (spmm_csr_mfma_kernel: structures prepared WITHOUT pairs, or not prepared): no pair descriptor is read, no
// rem_cnt carries bit 8, the edge-list walk knows no pair ranges -- the code every batch without sibling parts runs is the
// code it ran before pairs existed (this kernel answers epilogue edits with another register allocation: NEGATIVES.md).
template <bool PREP, int DROP, bool PAIRK, bool PP>
__device__ __forceinline__ void mf_body(const MfArgs &a, unsigned char *mf_smem, int bid, int groups) {
    MF_STAMP(0);
    // ---- workgroup -> (block, column group); the groups of one block stay on one XCD ----
    const int total = a.n_blocks * groups;
    const int per_xcd = (total + kXcds - 1) / kXcds;
    const int unit = (int)(bid % kXcds) * per_xcd + (int)(bid / kXcds);
    if (unit >= total) return;
    const int grp = unit % groups;
    const int rbk = unit / groups;
    // the block's pairs (prepared batches only): (first source row, source rows) x 2, uniform
    int4 pin = make_int4(0, 0, 0, 0);
    if constexpr (PREP && PP) {
        if (a.pairs && a.units == nullptr && a.row_blocks != nullptr && a.n_blocks <= MF_PAIR_BLOCKS) {
            const int4 t = *reinterpret_cast<const int4 *>(a.prep + (int64_t)rbk * MF_PREP_STRIDE + MF_PREP_PINFO);
            pin = make_int4(__builtin_amdgcn_readfirstlane(t.x), __builtin_amdgcn_readfirstlane(min(t.y, MF_ROWS)),
                            __builtin_amdgcn_readfirstlane(t.z), __builtin_amdgcn_readfirstlane(min(t.w, MF_ROWS)));
        }
    }
    const int n_pairs = pin.y > 0 ? (pin.w > 0 ? 2 : 1) : 0;
    // (routing the blocks with a row on a slow path -- walked or gathered in full -- to the per-tile workgroups as well
    // was measured: no gain, 32 set-ups instead of 12 cost what the finer split wins)
    if ((n_pairs > 0) != PAIRK) return;
    int r0, r1, xs0 = -1, xs1 = 0;
    if (a.units) { r0 = a.units[4 * rbk]; r1 = a.units[4 * rbk + 1]; xs0 = a.units[4 * rbk + 2]; xs1 = a.units[4 * rbk + 3]; }
    else if (a.row_blocks) { r0 = a.row_blocks[rbk]; r1 = a.row_blocks[rbk + 1]; }
    else { r0 = rbk * MF_ROWS; r1 = r0 + MF_ROWS; }
    r1 = min(r1, a.n_rows);
    const int nrow = a.units ? min(r1 - r0, MF_ROWS) : r1 - r0;
    if (nrow <= 0) return;
    const int nloc = min(nrow, MF_ROWS);
    if (xs0 < 0) { xs0 = r0; xs1 = r0 + nloc; }
    const int nx = min(xs1 - xs0, MF_ROWS);          // source rows staged = k extent of the product
    if (nx <= 0) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, cq = lane & 31;

    unsigned char *xt = mf_smem;
    float *yt = reinterpret_cast<float *>(mf_smem);
    unsigned char *ab = mf_smem + MF_A_OFF;
    int32_t *rp = reinterpret_cast<int32_t *>(mf_smem + MF_RP_OFF);
    int32_t *rem_cnt = reinterpret_cast<int32_t *>(mf_smem + MF_REMC_OFF);
    int32_t *rem_col = reinterpret_cast<int32_t *>(mf_smem + MF_REM_OFF);
    float *sc = reinterpret_cast<float *>(mf_smem + MF_SC_OFF);

    // staging role of this thread: rows 8 wave + 4 sh + i, columns 4 scq .. 4 scq + 3 of the tile, with (scq, sh) =
    // (lane >> 1, lane & 1): the 32 lanes of a ds_write_b64 group then write 16 slots x both 8-byte halves = 256
    // contiguous bytes (with (lane & 31, lane >> 5) a group wrote one half of 32 slots: 16-byte stride, two lanes per
    // bank pair and cycle -- SQ_LDS_BANK_CONFLICT 0.17 of the kernel's LDS cycles)
    const int sh = lane & 1, scq = lane >> 1;
    const int srow = 8 * wave + 4 * sh;
    float ss[4];
    float4 xv[4];
    int ct = grp;
    // (branch-free: a predicated load compiles to an exec-mask branch and a conservative s_waitcnt
    // vmcnt(0) at its join, which serialises the loads -- clamp the address, select when the value is used)
    auto load_tile = [&](int t) {
        const bool cok = t * MF_CT + 4 * scq < a.d;
        const float *px = a.x + (int64_t)xs0 * a.ldx + (cok ? t * MF_CT + 4 * scq : 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            xv[i] = *reinterpret_cast<const float4 *>(px + (int64_t)min(srow + i, nx - 1) * a.ldx);
    };
    auto first_loads = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) ss[i] = a.src_scale ? a.src_scale[xs0 + min(srow + i, nx - 1)] : 1.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) ss[i] = srow + i < nx ? ss[i] : 0.f;
        if (ct < a.n_col_tiles) load_tile(ct);
    };
    if constexpr (PREP) {
        // ---- the block's counts and outside neighbours were built once for the batch: copy them ----
        const unsigned char *src = a.prep + (int64_t)rbk * MF_PREP_STRIDE;
        const uint4 c0 = reinterpret_cast<const uint4 *>(src)[2 * tid], c1 = reinterpret_cast<const uint4 *>(src)[2 * tid + 1];
        const int4 rm = tid < (MF_ROWS * (1 + MF_REM)) / 4 ? reinterpret_cast<const int4 *>(src + MF_IMG_BYTES)[tid]
                                                           : make_int4(0, 0, 0, 0);
        const int rpv = (tid <= nloc && a.units == nullptr) ? a.rowptr[r0 + tid] : 0;      // (units: no gathered rows)
        const float scv = (tid < nloc && a.out_scale) ? a.out_scale[r0 + tid] : 1.f;
        first_loads();
        reinterpret_cast<uint4 *>(ab)[2 * tid] = c0;
        reinterpret_cast<uint4 *>(ab)[2 * tid + 1] = c1;
        if (tid < (MF_ROWS * (1 + MF_REM)) / 4) reinterpret_cast<int4 *>(mf_smem + MF_REMC_OFF)[tid] = rm;
        if (tid <= nloc) rp[tid] = rpv;
        if (tid < MF_ROWS) sc[tid] = scv;
        mf_barrier();
    } else {
        mf_build_block<false>(a, rbk, r0, nloc, mf_smem, first_loads);
    }
    MF_STAMP(3);

    const int rr = lane & 15, kg = lane >> 4;
    const int mt0 = (wave >> 2) * 2, nt0 = (wave & 3) * 2;
    const int n_ks = (nx + 31) >> 5;
    const bool m_on = mt0 * 16 < nloc, m_two = (mt0 + 1) * 16 < nloc;
    // B fragment slots of this lane's two output column tiles
    int bslot[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int n = (nt0 + ni) * 16 + rr;
        bslot[ni] = ((n & 3) * 36 + (n >> 2)) * 16;
    }
    // the rows this wave finishes: pairs (one per half wave), row = wave + 16 (2 p + half)
    // (their outside-neighbour counts and out scales are re-read from LDS in every tile's epilogue.  Kept as
    // two 8-element arrays from the set-up on, the scale of a lane's row was `half ? rsc[2 p + 1] : rsc[2 p]`,
    // which the compiler turned into an indexed load from a PRIVATE array: 40 B per lane stored to scratch at
    // set-up and re-loaded from memory in every epilogue -- the 9.8 MB by which WRITE_SIZE exceeded the
    // 33.5 MB output in the round-2 profile)
    constexpr int RW = MF_ROWS / MF_WAVES;
    // mask index of an element = a uniform 64-bit block base + a 32-bit in-block offset (DROP only)
    const uint64_t dyb = a.dr.y_base + (uint64_t)r0 * (uint64_t)a.dr.ld;
    const int dld = (int)a.dr.ld;
    // ---- the tile loop of a block with pairs (the pairs kernel); every other block's follows it ----
    if constexpr (PAIRK) {
        {
    // X tile in xv (x ss, rows >= nsrc zero) -> three bf16 pieces -> the X^T image
    auto convert = [&](int t, int nsrc) {
        // rows (0, 1) and (2, 3) of a column are converted in pairs: one v_cvt_pk_bf16_f32 per
        // piece gives the packed word the image wants, its two halves shifted / masked back to fp32
        // give the residuals
        const bool cok = t * MF_CT + 4 * scq < a.d;
        float xs[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = cok && srow + i < nsrc;                        // (clamped loads: select here)
            xs[i][0] = ok ? xv[i].x * ss[i] : 0.f; xs[i][1] = ok ? xv[i].y * ss[i] : 0.f;
            xs[i][2] = ok ? xv[i].z * ss[i] : 0.f; xs[i][3] = ok ? xv[i].w * ss[i] : 0.f;
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            uint32_t w[3][2];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                float x0 = xs[2 * pr][jj], x1 = xs[2 * pr + 1][jj];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const mf_bf16x2 pk = __builtin_convertvector(mf_f32x2{x0, x1}, mf_bf16x2);
                    const uint32_t u = __builtin_bit_cast(uint32_t, pk);
                    w[q][pr] = u;
                    x0 -= __builtin_bit_cast(float, u << 16);
                    x1 -= __builtin_bit_cast(float, u & 0xffff0000u);
                }
            }
            unsigned char *dst = xt + (wave * MF_CHUNK_SLOTS + jj * 36 + scq) * 16 + sh * 8;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                *reinterpret_cast<uint2 *>(dst + q * MF_PIECE) = make_uint2(w[q][0], w[q][1]);
        }
    };
    mf_f32x4 acc[2][2];
    // acc += counts image . X^T image over ksteps k steps of 32: 2 x 2 output tiles per wave
    auto mma = [&](int ksteps) {
        if (!m_on) return;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < ksteps) {
                mf_bf16x8 av[2], bv[3][2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    av[mi] = *reinterpret_cast<const mf_bf16x8 *>(
                        ab + ((ks * 4 + kg) * MF_ROWS + (mt0 + mi) * 16 + rr) * 16);
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        bv[q][ni] = *reinterpret_cast<const mf_bf16x8 *>(
                            xt + q * MF_PIECE + (ks * 4 + kg) * (MF_CHUNK_SLOTS * 16) + bslot[ni]);
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        acc[0][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], bv[q][ni], acc[0][ni], 0, 0, 0);
                        if (m_two)
                            acc[1][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], bv[q][ni], acc[1][ni], 0, 0, 0);
                    }
            }
        }
    };
    const int n_units_extra = n_pairs;
    int mf_it = 0;
    for (; ct < a.n_col_tiles; ct += groups, ++mf_it) {
        MF_STAMP(8 + 8 * mf_it);
        const int gc = ct * MF_CT + 4 * cq;
        const bool colok = gc < a.d;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mi][ni][e] = 0.f;
        // Unit 0 = the block itself; units 1 .. n_units_extra (prepared batches) = its pairs: rows of this block x the source
        // rows of a sibling block -- the same product with the pair's count image and X tile, into the same accumulators.
        int nsrc = nx;
        for (int u = 0;; ++u) {
            // ---- X tile -> x src_scale -> three bf16 pieces -> X^T image ----
            convert(ct, nsrc);
            MF_STAMP(9 + 8 * mf_it);
            if (n_units_extra > 0) __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): this unit's count image has landed
            // what comes next goes out now and lands under the MFMAs: the next pair's tile, or the next tile of the block
            if (u < n_units_extra) {
                const int qs0 = u == 0 ? pin.x : pin.z, qn = u == 0 ? pin.y : pin.w;
                const bool cok = ct * MF_CT + 4 * scq < a.d;
                const float *px = a.x + (int64_t)qs0 * a.ldx + (cok ? ct * MF_CT + 4 * scq : 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rq = min(srow + i, qn - 1);
                    xv[i] = *reinterpret_cast<const float4 *>(px + (int64_t)rq * a.ldx);
                    ss[i] = a.src_scale ? a.src_scale[qs0 + rq] : 1.f;
                }
            } else if (ct + groups < a.n_col_tiles) {
                if (n_units_extra > 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) ss[i] = a.src_scale ? a.src_scale[xs0 + min(srow + i, nx - 1)] : 1.f;
                }
                load_tile(ct + groups);
            }
            mf_barrier();
            MF_STAMP(10 + 8 * mf_it);
            mma((nsrc + 31) >> 5);
            MF_STAMP(11 + 8 * mf_it);
            mf_barrier();                                      // every wave is done reading both images
            if (u == n_units_extra) break;
            nsrc = u == 0 ? pin.y : pin.w;
            mf_image_to_lds(a.prep + (int64_t)a.n_blocks * MF_PREP_STRIDE + ((int64_t)rbk * MF_PAIRS + u) * MF_IMG_BYTES,
                            ab, wave, lane);
        }
        // (the block's own image comes back for the next tile: its conversion waits for it)
        if (n_units_extra > 0) mf_image_to_lds(a.prep + (int64_t)rbk * MF_PREP_STRIDE, ab, wave, lane);
        MF_STAMP(12 + 8 * mf_it);
        // ---- accumulators -> fp32 result tile (C/D of 16x16x32: col = lane & 15, row = 4 (lane >> 4) + e) ----
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    yt[((mt0 + mi) * 16 + 4 * kg + e) * MF_YT_PITCH + (nt0 + ni) * 16 + rr] = acc[mi][ni][e];
        mf_barrier();
        MF_STAMP(13 + 8 * mf_it);

        // ---- per row: + neighbours outside the block, x out_scale (+ y), store; two rows per pass ----
        // (row offsets inside the block in 32 bits, launcher-checked: as 64-bit products hoisted out of the tile
        // loop they were spilled to scratch)
        float *yblk = a.y + (int64_t)r0 * a.ldy;
        const int ldy32 = (int)a.ldy;
        int rcnt[RW];
        float rsc[RW / 2];                                 // this lane's rows: wave + 16 (2 p + half)
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int r = wave + MF_WAVES * i;
            const int rc = r < nloc ? __builtin_amdgcn_readfirstlane(rem_cnt[r]) : 0;
            rcnt[i] = (PP && rc >= 0) ? (rc & 0xff) : rc;
        }
#pragma unroll
        for (int pp = 0; pp < RW / 2; ++pp) rsc[pp] = sc[min(wave + MF_WAVES * (2 * pp + half), MF_ROWS - 1)];
        float4 v[RW / 2], yold[RW / 2];
#pragma unroll
        for (int pp = 0; pp < RW / 2; ++pp) yold[pp] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.accumulate) {                                // (uniform; addresses clamped, all four in flight)
#pragma unroll
            for (int pp = 0; pp < RW / 2; ++pp)
                yold[pp] = *reinterpret_cast<const float4 *>(
                    yblk + (uint32_t)(min(wave + MF_WAVES * (2 * pp + half), nloc - 1) * ldy32 + (colok ? gc : 0)));
        }
#pragma unroll
        for (int pp = 0; pp < RW / 2; ++pp)
            v[pp] = *reinterpret_cast<const float4 *>(
                yt + min(wave + MF_WAVES * (2 * pp + half), MF_ROWS - 1) * MF_YT_PITCH + 4 * cq);
        // the result tile (which aliases the next X^T image) is in registers: the stores below run
        // under the next tile's conversion
        mf_barrier();
        MF_STAMP(14 + 8 * mf_it);
#pragma unroll
        for (int pp = 0; pp < RW / 2; ++pp) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * pp + h;
                const int r = wave + MF_WAVES * i;
                if (r >= nloc || rcnt[i] == 0) continue;                   // wave-uniform
                const bool mine = colok && half == h;
                if (rcnt[i] > 0) {
                    const int ids = lane < MF_REM ? rem_col[r * MF_REM + lane] : 0;
                    mf_gather(a.x, a.ldx, a.src_scale, gc, mine, ids, rcnt[i], v[pp]);
                } else if (rcnt[i] == -2) {
                    mf_gather_row_outside<PP>(a, rp[r], rp[r + 1], r0, nloc, pin, lane, gc, mine, v[pp]);
                } else {
                    if (mine) v[pp] = make_float4(0.f, 0.f, 0.f, 0.f);
                    mf_gather_row(a, rp[r], rp[r + 1], lane, gc, mine, v[pp]);
                }
            }
            const int r = wave + MF_WAVES * (2 * pp + half);
            const float s = rsc[pp];
            if (colok && r < nloc) {
                float4 o = make_float4(fmaf(s, v[pp].x, yold[pp].x), fmaf(s, v[pp].y, yold[pp].y),
                                       fmaf(s, v[pp].z, yold[pp].z), fmaf(s, v[pp].w, yold[pp].w));
                if constexpr (DROP == 1) {
                    int rv_ = r;                       // (opaque: hoisted out of the tile loop, the four row
                    asm volatile("" : "+v"(rv_));      // products r * dld were spilled to scratch)
                    drop_f4(o, dyb + (uint32_t)(rv_ * dld + gc), a.dr);
                }
                *reinterpret_cast<float4 *>(yblk + (uint32_t)(r * ldy32 + gc)) = o;
            }
        }
        // rows of an oversized block beyond the 128 staged ones: gathered in full, two rows per pass
        for (int rb = MF_ROWS + 2 * wave; rb < nrow; rb += 2 * MF_WAVES) {
            float4 vo = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (rb + h >= nrow) continue;                              // wave-uniform
                mf_gather_row(a, a.rowptr[r0 + rb + h], a.rowptr[r0 + rb + h + 1], lane, gc, colok && half == h, vo);
            }
            const int r = rb + half;
            if (colok && r < nrow) {
                float *yp = a.y + (int64_t)(r0 + r) * a.ldy + gc;
                const uint64_t yi = a.dr.y_base + (uint64_t)(r0 + r) * (uint64_t)a.dr.ld + (uint64_t)gc;
                float4 yo = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.accumulate) yo = *reinterpret_cast<const float4 *>(yp);
                const float s = a.out_scale ? a.out_scale[r0 + r] : 1.f;
                float4 o = make_float4(fmaf(s, vo.x, yo.x), fmaf(s, vo.y, yo.y), fmaf(s, vo.z, yo.z), fmaf(s, vo.w, yo.w));
                if constexpr (DROP == 1) drop_f4(o, yi, a.dr);
                *reinterpret_cast<float4 *>(yp) = o;
            }
        }
        MF_STAMP(15 + 8 * mf_it);
    }
            return;
        }
    }
    int mf_it = 0;
    for (; ct < a.n_col_tiles; ct += groups, ++mf_it) {
        MF_STAMP(8 + 8 * mf_it);
        // ---- X tile -> x src_scale -> three bf16 pieces -> X^T image ----
        {   // rows (0, 1) and (2, 3) of a column are converted in pairs: one v_cvt_pk_bf16_f32 per
            // piece gives the packed word the image wants, its two halves shifted / masked back to fp32
            // give the residuals
            const bool cok = ct * MF_CT + 4 * scq < a.d;
            float xs[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = cok && srow + i < nx;                      // (clamped loads: select here)
                xs[i][0] = ok ? xv[i].x * ss[i] : 0.f; xs[i][1] = ok ? xv[i].y * ss[i] : 0.f;
                xs[i][2] = ok ? xv[i].z * ss[i] : 0.f; xs[i][3] = ok ? xv[i].w * ss[i] : 0.f;
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                uint32_t w[3][2];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    float x0 = xs[2 * pr][jj], x1 = xs[2 * pr + 1][jj];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const mf_bf16x2 pk = __builtin_convertvector(mf_f32x2{x0, x1}, mf_bf16x2);
                        const uint32_t u = __builtin_bit_cast(uint32_t, pk);
                        w[q][pr] = u;
                        x0 -= __builtin_bit_cast(float, u << 16);
                        x1 -= __builtin_bit_cast(float, u & 0xffff0000u);
                    }
                }
                unsigned char *dst = xt + (wave * MF_CHUNK_SLOTS + jj * 36 + scq) * 16 + sh * 8;
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    *reinterpret_cast<uint2 *>(dst + q * MF_PIECE) = make_uint2(w[q][0], w[q][1]);
            }
        }
        MF_STAMP(9 + 8 * mf_it);
        if (ct + groups < a.n_col_tiles) load_tile(ct + groups);      // next tile, in flight under the MFMAs
        const int gc = ct * MF_CT + 4 * cq;
        const bool colok = gc < a.d;
        mf_barrier();
        MF_STAMP(10 + 8 * mf_it);

        // ---- counts . X^T: 2 x 2 output tiles per wave ----
        mf_f32x4 acc[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mi][ni][e] = 0.f;
        if (m_on) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < n_ks) {
                    mf_bf16x8 av[2], bv[3][2];
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
                        av[mi] = *reinterpret_cast<const mf_bf16x8 *>(
                            ab + ((ks * 4 + kg) * MF_ROWS + (mt0 + mi) * 16 + rr) * 16);
#pragma unroll
                    for (int q = 0; q < 3; ++q)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            bv[q][ni] = *reinterpret_cast<const mf_bf16x8 *>(
                                xt + q * MF_PIECE + (ks * 4 + kg) * (MF_CHUNK_SLOTS * 16) + bslot[ni]);
#pragma unroll
                    for (int q = 0; q < 3; ++q)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            acc[0][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], bv[q][ni], acc[0][ni], 0, 0, 0);
                            if (m_two)
                                acc[1][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], bv[q][ni], acc[1][ni], 0, 0, 0);
                        }
                }
            }
        }
        MF_STAMP(11 + 8 * mf_it);
        mf_barrier();                                      // every wave is done reading the X^T image
        MF_STAMP(12 + 8 * mf_it);
        // ---- accumulators -> fp32 result tile (C/D of 16x16x32: col = lane & 15, row = 4 (lane >> 4) + e) ----
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    yt[((mt0 + mi) * 16 + 4 * kg + e) * MF_YT_PITCH + (nt0 + ni) * 16 + rr] = acc[mi][ni][e];
        mf_barrier();
        MF_STAMP(13 + 8 * mf_it);

        // ---- per row: + neighbours outside the block, x out_scale (+ y), store; two rows per pass ----
        // (row offsets inside the block in 32 bits, launcher-checked: as 64-bit products hoisted out of the tile
        // loop they were spilled to scratch)
        float *yblk = a.y + (int64_t)r0 * a.ldy;
        const int ldy32 = (int)a.ldy;
        int rcnt[RW];
        float rsc[RW / 2];                                 // this lane's rows: wave + 16 (2 p + half)
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int r = wave + MF_WAVES * i;
            const int rc = r < nloc ? __builtin_amdgcn_readfirstlane(rem_cnt[r]) : 0;
            rcnt[i] = (PP && rc >= 0) ? (rc & 0xff) : rc;
        }
#pragma unroll
        for (int pp = 0; pp < RW / 2; ++pp) rsc[pp] = sc[min(wave + MF_WAVES * (2 * pp + half), MF_ROWS - 1)];
        float4 v[RW / 2], yold[RW / 2];
#pragma unroll
        for (int pp = 0; pp < RW / 2; ++pp) yold[pp] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.accumulate) {                                // (uniform; addresses clamped, all four in flight)
#pragma unroll
            for (int pp = 0; pp < RW / 2; ++pp)
                yold[pp] = *reinterpret_cast<const float4 *>(
                    yblk + (uint32_t)(min(wave + MF_WAVES * (2 * pp + half), nloc - 1) * ldy32 + (colok ? gc : 0)));
        }
#pragma unroll
        for (int pp = 0; pp < RW / 2; ++pp)
            v[pp] = *reinterpret_cast<const float4 *>(
                yt + min(wave + MF_WAVES * (2 * pp + half), MF_ROWS - 1) * MF_YT_PITCH + 4 * cq);
        // the result tile (which aliases the next X^T image) is in registers: the stores below run
        // under the next tile's conversion
        mf_barrier();
        MF_STAMP(14 + 8 * mf_it);
#pragma unroll
        for (int pp = 0; pp < RW / 2; ++pp) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * pp + h;
                const int r = wave + MF_WAVES * i;
                if (r >= nloc || rcnt[i] == 0) continue;                   // wave-uniform
                const bool mine = colok && half == h;
                if (rcnt[i] > 0) {
                    const int ids = lane < MF_REM ? rem_col[r * MF_REM + lane] : 0;
                    mf_gather(a.x, a.ldx, a.src_scale, gc, mine, ids, rcnt[i], v[pp]);
                } else if (rcnt[i] == -2) {
                    mf_gather_row_outside<PP>(a, rp[r], rp[r + 1], r0, nloc, pin, lane, gc, mine, v[pp]);
                } else {
                    if (mine) v[pp] = make_float4(0.f, 0.f, 0.f, 0.f);
                    mf_gather_row(a, rp[r], rp[r + 1], lane, gc, mine, v[pp]);
                }
            }
            const int r = wave + MF_WAVES * (2 * pp + half);
            const float s = rsc[pp];
            if (colok && r < nloc) {
                float4 o = make_float4(fmaf(s, v[pp].x, yold[pp].x), fmaf(s, v[pp].y, yold[pp].y),
                                       fmaf(s, v[pp].z, yold[pp].z), fmaf(s, v[pp].w, yold[pp].w));
                if constexpr (DROP == 1) {
                    int rv_ = r;                       // (opaque: hoisted out of the tile loop, the four row
                    asm volatile("" : "+v"(rv_));      // products r * dld were spilled to scratch)
                    drop_f4(o, dyb + (uint32_t)(rv_ * dld + gc), a.dr);
                }
                *reinterpret_cast<float4 *>(yblk + (uint32_t)(r * ldy32 + gc)) = o;
            }
        }
        // rows of an oversized block beyond the 128 staged ones: gathered in full, two rows per pass
        for (int rb = MF_ROWS + 2 * wave; rb < nrow; rb += 2 * MF_WAVES) {
            float4 vo = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (rb + h >= nrow) continue;                              // wave-uniform
                mf_gather_row(a, a.rowptr[r0 + rb + h], a.rowptr[r0 + rb + h + 1], lane, gc, colok && half == h, vo);
            }
            const int r = rb + half;
            if (colok && r < nrow) {
                float *yp = a.y + (int64_t)(r0 + r) * a.ldy + gc;
                const uint64_t yi = a.dr.y_base + (uint64_t)(r0 + r) * (uint64_t)a.dr.ld + (uint64_t)gc;
                float4 yo = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.accumulate) yo = *reinterpret_cast<const float4 *>(yp);
                const float s = a.out_scale ? a.out_scale[r0 + r] : 1.f;
                float4 o = make_float4(fmaf(s, vo.x, yo.x), fmaf(s, vo.y, yo.y), fmaf(s, vo.z, yo.z), fmaf(s, vo.w, yo.w));
                if constexpr (DROP == 1) drop_f4(o, yi, a.dr);
                *reinterpret_cast<float4 *>(yp) = o;
            }
        }
        MF_STAMP(15 + 8 * mf_it);
    }
}

template <bool PREP, int DROP = 0>
__global__ __launch_bounds__(MF_THREADS) void spmm_csr_mfma_kernel(MfArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mf_smem[];
    mf_body<PREP, DROP, false, false>(a, mf_smem, (int)blockIdx.x, a.groups);
}
// Prepared structure WITH pairs: ONE launch whose first grid_main workgroups are spmm_csr_mfma_kernel's (a block with
// pairs: exit) and whose others take the blocks with pairs, one workgroup per (block, column tile) -- they start on the
// CUs the first ones leave free (their own blocks' workgroups exit at once, 240 of 256 CUs are used anyway) and run beside
// them.  The two bodies share no value: the choice is made on blockIdx alone, before anything else.
template <int DROP>
__global__ __launch_bounds__(MF_THREADS) void spmm_csr_mfma_pairs_kernel(MfArgs a, int grid_main) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mf_smem[];
    if ((int)blockIdx.x < grid_main) mf_body<true, DROP, false, true>(a, mf_smem, (int)blockIdx.x, a.groups);
    else mf_body<true, DROP, true, true>(a, mf_smem, (int)blockIdx.x - grid_main, (a.n_col_tiles + MF_FINE_TILES - 1) / MF_FINE_TILES);
}

// One workgroup per block: its counts image and outside-neighbour lists -> memory, for every aggregation
// over the same graph and blocks (gist_spmm_blocks_prepare).
__global__ __launch_bounds__(MF_THREADS) void spmm_blocks_prepare_kernel(MfArgs a, const int32_t *rowptr2,
                                                                         const int32_t *col2, unsigned char *out,
                                                                         unsigned char *out2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mf_smem[];
    const int rbk = blockIdx.x;
    if (blockIdx.y == 1) { a.rowptr = rowptr2; a.col = col2; out = out2; }      // second graph, same blocks
    int r0, r1;
    if (a.row_blocks) { r0 = a.row_blocks[rbk]; r1 = a.row_blocks[rbk + 1]; }
    else { r0 = rbk * MF_ROWS; r1 = r0 + MF_ROWS; }
    r1 = min(r1, a.n_rows);
    const int nloc = min(max(r1 - r0, 0), MF_ROWS);
    const int tid = threadIdx.x;
    unsigned char *dst = out + (int64_t)rbk * MF_PREP_STRIDE;
    const bool pairs = a.pairs && a.row_blocks != nullptr && a.n_blocks <= MF_PAIR_BLOCKS;
    if (nloc > 0) {
        if (pairs) mf_build_block<true>(a, rbk, r0, nloc, mf_smem, [] {});
        else mf_build_block<false>(a, rbk, r0, nloc, mf_smem, [] {});
        reinterpret_cast<uint4 *>(dst)[2 * tid] = reinterpret_cast<const uint4 *>(mf_smem + MF_A_OFF)[2 * tid];
        reinterpret_cast<uint4 *>(dst)[2 * tid + 1] = reinterpret_cast<const uint4 *>(mf_smem + MF_A_OFF)[2 * tid + 1];
        if (tid < (MF_ROWS * (1 + MF_REM)) / 4)
            reinterpret_cast<int4 *>(dst + MF_IMG_BYTES)[tid] = reinterpret_cast<const int4 *>(mf_smem + MF_REMC_OFF)[tid];
        const int4 pin = *reinterpret_cast<const int4 *>(mf_smem + MF_PINFO_OFF);
        if (tid == 0) *reinterpret_cast<int4 *>(dst + MF_PREP_PINFO) = pin;
        if (pairs && pin.y > 0) {      // the block's pair images (the second one only if there is a second pair)
            unsigned char *pd = out + (int64_t)a.n_blocks * MF_PREP_STRIDE + (int64_t)rbk * (MF_PAIRS * MF_IMG_BYTES);
            const int n16 = (pin.w > 0 ? 2 : 1) * (MF_IMG_BYTES / 16);
            for (int i = tid; i < n16; i += MF_THREADS)
                reinterpret_cast<uint4 *>(pd)[i] = reinterpret_cast<const uint4 *>(mf_smem)[i];
        }
    } else if (tid == 0) {
        *reinterpret_cast<int4 *>(dst + MF_PREP_PINFO) = make_int4(0, 0, 0, 0);
    }
}

static int mf_set_lds(const void *kernel, const char *name) {
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MF_LDS_BYTES);
    if (e != hipSuccess) {
        set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
        return GIST_ELAUNCH;
    }
    return GIST_OK;
}

// the block records, then -- batches of <= MF_PAIR_BLOCKS blocks -- MF_PAIRS pair images per block
thread_local bool tl_spmm_pairs = true;

int64_t spmm_blocks_bytes(int64_t n_blocks) {
    if (n_blocks <= 0) return 0;
    return n_blocks * (int64_t)MF_PREP_STRIDE + (n_blocks <= MF_PAIR_BLOCKS ? n_blocks * (int64_t)(MF_PAIRS * MF_IMG_BYTES) : 0);
}

// rowptr2 / col2 / prepared2: optionally a second graph over the same rows and blocks (the reversed
// CSR of the backward aggregation), prepared by the same launch
int launch_spmm_blocks_prepare(const int32_t *rowptr, const int32_t *col, const int32_t *rowptr2,
                               const int32_t *col2, int64_t n_rows, const int32_t *row_blocks,
                               int64_t n_row_blocks, void *prepared, void *prepared2, hipStream_t st, bool pairs) {
    MfArgs a{};
    a.rowptr = rowptr; a.col = col; a.n_rows = (int)n_rows; a.row_blocks = row_blocks;
    a.pairs = (pairs && tl_spmm_pairs) ? 1 : 0;
    const int64_t nb = row_blocks ? n_row_blocks : ceil_div(n_rows, MF_ROWS);
    a.n_blocks = (int)nb;
    if (nb <= 0) return GIST_OK;
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        const int rc = mf_set_lds(reinterpret_cast<const void *>(&spmm_blocks_prepare_kernel), "gist_spmm_blocks_prepare");
        if (rc != GIST_OK) return rc;
        once.done(dev);
    }
    const bool two = rowptr2 != nullptr && col2 != nullptr && prepared2 != nullptr;
    hipLaunchKernelGGL(spmm_blocks_prepare_kernel, dim3((unsigned)nb, two ? 2u : 1u), dim3(MF_THREADS), MF_LDS_BYTES,
                       st, a, rowptr2, col2, static_cast<unsigned char *>(prepared),
                       static_cast<unsigned char *>(prepared2));
    return launch_status("gist_spmm_blocks_prepare");
}

int launch_spmm_mfma(const int32_t *rowptr, const int32_t *col, const float *x, int64_t ldx, float *y,
                     int64_t ldy, int64_t n_rows, int64_t d, const float *out_scale, const float *src_scale,
                     int accumulate, const int32_t *row_blocks, int64_t n_row_blocks, const void *prepared,
                     hipStream_t st, const SpmmDrop *dr, bool pairs) {
    GIST_REQUIRE(ldy < (1LL << 22) && ldx < (1LL << 22) && d < (1LL << 22),
                 "gist_spmm_csr_blocked_f32: row pitch of 2^22 floats or more");       // 32-bit offsets inside a block
    MfArgs a{};
    a.rowptr = rowptr; a.col = col; a.x = x; a.ldx = ldx; a.y = y; a.ldy = ldy;
    a.n_rows = (int)n_rows; a.d = (int)d; a.out_scale = out_scale; a.src_scale = src_scale;
    a.accumulate = accumulate; a.row_blocks = row_blocks;
    a.prep = static_cast<const unsigned char *>(prepared);
    a.pairs = (pairs && tl_spmm_pairs && prepared != nullptr && row_blocks != nullptr && n_row_blocks <= MF_PAIR_BLOCKS) ? 1 : 0;
    a.dr = dr ? *dr : SpmmDrop{};
    GIST_REQUIRE(a.dr.ld < (1LL << 22), "gist_spmm_csr_drop_f32: mask pitch of 2^22 elements or more");
    const int mode = dr ? dr->mode : 0;
    GIST_REQUIRE(mode == 0 || mode == 1, "gist_spmm_csr_drop_f32: the matrix-core kernel carries the forward mask only");
    const int64_t nb = row_blocks ? n_row_blocks : ceil_div(n_rows, MF_ROWS);
    a.n_blocks = (int)nb;
    a.n_col_tiles = (int)ceil_div(d, MF_CT);
    // one workgroup per CU at a time: as many column groups per block as fill the chip once
    int64_t groups = nb > 0 ? 256 / nb : 1;
    if (groups < 1) groups = 1;
    if (groups > a.n_col_tiles) groups = a.n_col_tiles;
    a.groups = (int)groups;
    const int64_t total = nb * groups;
    const int64_t grid = kXcds * ceil_div(total, kXcds);
    if (grid > 0x7fffffffLL) { set_error("gist_spmm_csr_blocked_f32: grid too large"); return GIST_EINVAL; }
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        const void *ks[6] = {reinterpret_cast<const void *>(&spmm_csr_mfma_kernel<false, 0>),
                             reinterpret_cast<const void *>(&spmm_csr_mfma_kernel<true, 0>),
                             reinterpret_cast<const void *>(&spmm_csr_mfma_kernel<false, 1>),
                             reinterpret_cast<const void *>(&spmm_csr_mfma_kernel<true, 1>),
                             reinterpret_cast<const void *>(&spmm_csr_mfma_pairs_kernel<0>),
                             reinterpret_cast<const void *>(&spmm_csr_mfma_pairs_kernel<1>)};
        for (const void *k : ks) {
            const int rc = mf_set_lds(k, "gist_spmm_csr_blocked_f32");
            if (rc != GIST_OK) return rc;
        }
        once.done(dev);
    }
#define MF_GO(P, D)                                                                                             \
    hipLaunchKernelGGL((spmm_csr_mfma_kernel<P, D>), dim3((unsigned)grid), dim3(MF_THREADS), MF_LDS_BYTES, st, a)
    if (prepared && a.pairs) {
        // + the blocks with pairs, one workgroup per (block, column tile), behind the others in the same grid
        const int64_t grid_p = kXcds * ceil_div(nb * ((a.n_col_tiles + MF_FINE_TILES - 1) / MF_FINE_TILES), kXcds);
        if (grid + grid_p > 0x7fffffffLL) { set_error("gist_spmm_csr_blocked_f32: grid too large"); return GIST_EINVAL; }
        if (mode == 1)
            hipLaunchKernelGGL((spmm_csr_mfma_pairs_kernel<1>), dim3((unsigned)(grid + grid_p)), dim3(MF_THREADS), MF_LDS_BYTES,
                               st, a, (int)grid);
        else
            hipLaunchKernelGGL((spmm_csr_mfma_pairs_kernel<0>), dim3((unsigned)(grid + grid_p)), dim3(MF_THREADS), MF_LDS_BYTES,
                               st, a, (int)grid);
    } else if (prepared) {
        if (mode == 1) MF_GO(true, 1); else MF_GO(true, 0);
    } else {
        if (mode == 1) MF_GO(false, 1); else MF_GO(false, 0);
    }
#undef MF_GO
    return launch_status("gist_spmm_csr_blocked_f32");
}

// Off-diagonal block pairs of a part-ordered graph (full-graph evaluation): unit u = (r0, r1, xs0, xs1) computes
// y[r0 .. r1) (+)= out_scale . C_u . x[xs0 .. xs1) with C_u = the u-th prepared counts image (no outside neighbours).
// Units of ONE launch must have disjoint output rows (the evaluator launches the j-th pair of every row block together).
int launch_spmm_mfma_units(const int32_t *units, int64_t n_units, const void *prepared, const float *x, int64_t ldx,
                           float *y, int64_t ldy, int64_t n_rows_y, int64_t d, const float *out_scale, int accumulate,
                           hipStream_t st) {
    GIST_REQUIRE(ldy < (1LL << 22) && ldx < (1LL << 22) && d < (1LL << 22),
                 "gist_spmm_block_units_f32: row pitch of 2^22 floats or more");
    MfArgs a{};
    a.x = x; a.ldx = ldx; a.y = y; a.ldy = ldy; a.n_rows = (int)n_rows_y; a.d = (int)d;
    a.out_scale = out_scale; a.accumulate = accumulate; a.units = units;
    a.prep = static_cast<const unsigned char *>(prepared);
    a.n_blocks = (int)n_units;
    a.n_col_tiles = (int)ceil_div(d, MF_CT);
    int64_t groups = n_units > 0 ? 256 / n_units : 1;
    if (groups < 1) groups = 1;
    if (groups > a.n_col_tiles) groups = a.n_col_tiles;
    a.groups = (int)groups;
    const int64_t grid = kXcds * ceil_div(n_units * groups, kXcds);
    if (grid > 0x7fffffffLL) { set_error("gist_spmm_block_units_f32: grid too large"); return GIST_EINVAL; }
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        const int rc = mf_set_lds(reinterpret_cast<const void *>(&spmm_csr_mfma_kernel<true, 0>), "gist_spmm_block_units_f32");
        if (rc != GIST_OK) return rc;
        once.done(dev);
    }
    hipLaunchKernelGGL((spmm_csr_mfma_kernel<true, 0>), dim3((unsigned)grid), dim3(MF_THREADS), MF_LDS_BYTES, st, a);
    return launch_status("gist_spmm_block_units_f32");
}

// ---- chains of block pairs (full-graph evaluation, round 5) ---------------------------------------------------------
// gist_spmm_block_units_f32 launches the j-th pair of every row block together and accumulates into y: per (pair, 128-column
// tile) 64 KB of X read + 64 KB of y read + 64 KB of y written = 7.5 us at a CU's share of the memory system, 17 ms for the
// 18 224 pairs of the Reddit-sized graph at D = 4096 (profiles/NEGATIVES.md).  Here ONE workgroup owns a row block's tile across
// ALL units that write it (its diagonal block and every dense off-diagonal pair: a chain): the accumulators stay in registers
// from the chain's first unit to its last, per unit the 32-KiB count image and the 64-KB X tile are loaded (both prefetched
// into registers under the previous unit's MFMAs), y is written once.
struct MfChainArgs {
    const int32_t *chain_ptr;        // [n_chains + 1]: units of chain c = [chain_ptr[c], chain_ptr[c + 1])
    const int32_t *units;            // [n_units][4] = (r0, r1, xs0, xs1); r0 / r1 equal within a chain
    const unsigned char *images;     // unit u's count image at images + u * MF_PREP_STRIDE
    const float *x; int64_t ldx;
    float *y; int64_t ldy;
    int n_rows, d;
    const float *out_scale;
    int accumulate, n_chains, n_col_tiles, groups;
};

__global__ __launch_bounds__(MF_THREADS) void spmm_chain_mfma_kernel(MfChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mf_smem[];
    const int total = a.n_chains * a.groups;
    const int per_xcd = (total + kXcds - 1) / kXcds;
    const int unit = (int)(blockIdx.x % kXcds) * per_xcd + (int)(blockIdx.x / kXcds);
    if (unit >= total) return;
    const int grp = unit % a.groups;
    const int chain = unit / a.groups;
    const int u0 = a.chain_ptr[chain], u1 = a.chain_ptr[chain + 1];
    if (u1 <= u0) return;
    const int r0 = a.units[4 * u0];
    const int r1 = min(a.units[4 * u0 + 1], a.n_rows);
    const int nloc = min(r1 - r0, MF_ROWS);
    if (nloc <= 0) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, cq = lane & 31;
    unsigned char *xt = mf_smem;
    float *yt = reinterpret_cast<float *>(mf_smem);
    unsigned char *ab = mf_smem + MF_A_OFF;
    float *sc = reinterpret_cast<float *>(mf_smem + MF_SC_OFF);
    const int sh = lane & 1, scq = lane >> 1;
    const int srow = 8 * wave + 4 * sh;
    const int rr = lane & 15, kg = lane >> 4;
    const int mt0 = (wave >> 2) * 2, nt0 = (wave & 3) * 2;
    const bool m_on = mt0 * 16 < nloc, m_two = (mt0 + 1) * 16 < nloc;
    int bslot[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int n = (nt0 + ni) * 16 + rr;
        bslot[ni] = ((n & 3) * 36 + (n >> 2)) * 16;
    }
    if (tid < MF_ROWS) sc[tid] = (tid < nloc && a.out_scale) ? a.out_scale[r0 + tid] : 1.f;
    // what is in flight for the NEXT (unit, tile): its X tile rows and its count image
    float4 xv[4];
    uint4 cv0, cv1;
    int nx_next = 0;
    auto prefetch = [&](int u, int t) {          // (clamped addresses, selected when used: branch-free loads)
        const int xs0 = a.units[4 * u + 2];
        nx_next = min(a.units[4 * u + 3] - xs0, MF_ROWS);
        const bool cok = t * MF_CT + 4 * scq < a.d;
        const float *px = a.x + (int64_t)xs0 * a.ldx + (cok ? t * MF_CT + 4 * scq : 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            xv[i] = *reinterpret_cast<const float4 *>(px + (int64_t)min(srow + i, max(nx_next, 1) - 1) * a.ldx);
        const unsigned char *src = a.images + (int64_t)u * MF_PREP_STRIDE;
        cv0 = reinterpret_cast<const uint4 *>(src)[2 * tid];
        cv1 = reinterpret_cast<const uint4 *>(src)[2 * tid + 1];
    };
    constexpr int RW = MF_ROWS / MF_WAVES;
    int ct = grp;
    if (ct < a.n_col_tiles) prefetch(u0, ct);
    for (; ct < a.n_col_tiles; ct += a.groups) {
        mf_f32x4 acc[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mi][ni][e] = 0.f;
        for (int u = u0; u < u1; ++u) {
            const int nx = nx_next;
            // ---- this unit's X tile -> three bf16 pieces -> X^T image; its counts -> the A image ----
            {
                const bool cok = ct * MF_CT + 4 * scq < a.d;
                float xs[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = cok && srow + i < nx;
                    xs[i][0] = ok ? xv[i].x : 0.f; xs[i][1] = ok ? xv[i].y : 0.f;
                    xs[i][2] = ok ? xv[i].z : 0.f; xs[i][3] = ok ? xv[i].w : 0.f;
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    uint32_t w[3][2];
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        float x0 = xs[2 * pr][jj], x1 = xs[2 * pr + 1][jj];
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            const mf_bf16x2 pk = __builtin_convertvector(mf_f32x2{x0, x1}, mf_bf16x2);
                            const uint32_t uu = __builtin_bit_cast(uint32_t, pk);
                            w[q][pr] = uu;
                            x0 -= __builtin_bit_cast(float, uu << 16);
                            x1 -= __builtin_bit_cast(float, uu & 0xffff0000u);
                        }
                    }
                    unsigned char *dst = xt + (wave * MF_CHUNK_SLOTS + jj * 36 + scq) * 16 + sh * 8;
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        *reinterpret_cast<uint2 *>(dst + q * MF_PIECE) = make_uint2(w[q][0], w[q][1]);
                }
                reinterpret_cast<uint4 *>(ab)[2 * tid] = cv0;
                reinterpret_cast<uint4 *>(ab)[2 * tid + 1] = cv1;
            }
            // the next (unit, tile) in flight under this unit's MFMAs
            if (u + 1 < u1) prefetch(u + 1, ct);
            else if (ct + a.groups < a.n_col_tiles) prefetch(u0, ct + a.groups);
            mf_barrier();
            const int n_ks = (nx + 31) >> 5;
            if (m_on) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks < n_ks) {
                        mf_bf16x8 av[2], bv[3][2];
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi)
                            av[mi] = *reinterpret_cast<const mf_bf16x8 *>(
                                ab + ((ks * 4 + kg) * MF_ROWS + (mt0 + mi) * 16 + rr) * 16);
#pragma unroll
                        for (int q = 0; q < 3; ++q)
#pragma unroll
                            for (int ni = 0; ni < 2; ++ni)
                                bv[q][ni] = *reinterpret_cast<const mf_bf16x8 *>(
                                    xt + q * MF_PIECE + (ks * 4 + kg) * (MF_CHUNK_SLOTS * 16) + bslot[ni]);
#pragma unroll
                        for (int q = 0; q < 3; ++q)
#pragma unroll
                            for (int ni = 0; ni < 2; ++ni) {
                                acc[0][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], bv[q][ni], acc[0][ni], 0, 0, 0);
                                if (m_two)
                                    acc[1][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], bv[q][ni], acc[1][ni], 0, 0, 0);
                            }
                    }
                }
            }
            mf_barrier();                                  // every wave is done reading both images
        }
        // ---- accumulators -> fp32 result tile -> rows: x out_scale (+ y), store ----
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    yt[((mt0 + mi) * 16 + 4 * kg + e) * MF_YT_PITCH + (nt0 + ni) * 16 + rr] = acc[mi][ni][e];
        mf_barrier();
        const int gc = ct * MF_CT + 4 * cq;
        const bool colok = gc < a.d;
        float *yblk = a.y + (int64_t)r0 * a.ldy;
        const int ldy32 = (int)a.ldy;
        float4 v[RW / 2], yold[RW / 2];
#pragma unroll
        for (int pp = 0; pp < RW / 2; ++pp) yold[pp] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.accumulate) {
#pragma unroll
            for (int pp = 0; pp < RW / 2; ++pp)
                yold[pp] = *reinterpret_cast<const float4 *>(
                    yblk + (uint32_t)(min(wave + MF_WAVES * (2 * pp + half), nloc - 1) * ldy32 + (colok ? gc : 0)));
        }
#pragma unroll
        for (int pp = 0; pp < RW / 2; ++pp)
            v[pp] = *reinterpret_cast<const float4 *>(
                yt + min(wave + MF_WAVES * (2 * pp + half), MF_ROWS - 1) * MF_YT_PITCH + 4 * cq);
        mf_barrier();                                      // (the result tile aliases the next X^T image)
#pragma unroll
        for (int pp = 0; pp < RW / 2; ++pp) {
            const int r = wave + MF_WAVES * (2 * pp + half);
            const float s = sc[min(r, MF_ROWS - 1)];
            if (colok && r < nloc) {
                const float4 o = make_float4(fmaf(s, v[pp].x, yold[pp].x), fmaf(s, v[pp].y, yold[pp].y),
                                             fmaf(s, v[pp].z, yold[pp].z), fmaf(s, v[pp].w, yold[pp].w));
                *reinterpret_cast<float4 *>(yblk + (uint32_t)(r * ldy32 + gc)) = o;
            }
        }
    }
}

int launch_spmm_mfma_chains(const int32_t *chain_ptr, int64_t n_chains, const int32_t *units, const void *images,
                            const float *x, int64_t ldx, float *y, int64_t ldy, int64_t n_rows_y, int64_t d,
                            const float *out_scale, int accumulate, hipStream_t st) {
    GIST_REQUIRE(ldy < (1LL << 22) && ldx < (1LL << 22) && d < (1LL << 22),
                 "gist_spmm_block_chains_f32: row pitch of 2^22 floats or more");
    MfChainArgs a{};
    a.chain_ptr = chain_ptr; a.units = units; a.images = static_cast<const unsigned char *>(images);
    a.x = x; a.ldx = ldx; a.y = y; a.ldy = ldy; a.n_rows = (int)n_rows_y; a.d = (int)d;
    a.out_scale = out_scale; a.accumulate = accumulate; a.n_chains = (int)n_chains;
    a.n_col_tiles = (int)ceil_div(d, MF_CT);
    int64_t groups = n_chains > 0 ? 512 / n_chains : 1;      // (two rounds of workgroups per CU at least)
    if (groups < 1) groups = 1;
    if (groups > a.n_col_tiles) groups = a.n_col_tiles;
    a.groups = (int)groups;
    const int64_t grid = kXcds * ceil_div(n_chains * groups, kXcds);
    if (grid > 0x7fffffffLL) { set_error("gist_spmm_block_chains_f32: grid too large"); return GIST_EINVAL; }
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        const int rc = mf_set_lds(reinterpret_cast<const void *>(&spmm_chain_mfma_kernel), "gist_spmm_block_chains_f32");
        if (rc != GIST_OK) return rc;
        once.done(dev);
    }
    hipLaunchKernelGGL(spmm_chain_mfma_kernel, dim3((unsigned)grid), dim3(MF_THREADS), MF_LDS_BYTES, st, a);
    return launch_status("gist_spmm_block_chains_f32");
}

}  // namespace gist

extern "C" int64_t gist_spmm_block_image_bytes(void) { return gist::MF_PREP_STRIDE; }
extern "C" int32_t gist_spmm_pair_min_edges(void) { return gist::MF_PAIR_MIN; }

extern "C" int gist_spmm_block_chains_f32(const int32_t *chain_ptr, int64_t n_chains, const int32_t *units, const void *images,
                                          const float *x, int64_t ldx, float *y, int64_t ldy, int64_t n_rows_y, int64_t d,
                                          const float *out_scale, int accumulate, gist_stream_t stream) {
    using namespace gist;
    GIST_REQUIRE(n_chains >= 0 && d >= 0 && n_rows_y >= 0, "gist_spmm_block_chains_f32: negative size");
    if (n_chains == 0 || d == 0) return GIST_OK;
    GIST_REQUIRE(chain_ptr && units && images && x && y, "gist_spmm_block_chains_f32: null pointer");
    GIST_REQUIRE(d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= d && ldy >= d && aligned16(x) && aligned16(y) &&
                     aligned16(images),
                 "gist_spmm_block_chains_f32: rows must be 16-byte aligned multiples of 4 floats");
    return launch_spmm_mfma_chains(chain_ptr, n_chains, units, images, x, ldx, y, ldy, n_rows_y, d, out_scale, accumulate,
                                   as_stream(stream));
}

extern "C" int gist_spmm_block_units_f32(const int32_t *units, int64_t n_units, const void *images, const float *x,
                                         int64_t ldx, float *y, int64_t ldy, int64_t n_rows_y, int64_t d,
                                         const float *out_scale, int accumulate, gist_stream_t stream) {
    using namespace gist;
    GIST_REQUIRE(n_units >= 0 && d >= 0 && n_rows_y >= 0, "gist_spmm_block_units_f32: negative size");
    if (n_units == 0 || d == 0) return GIST_OK;
    GIST_REQUIRE(units && images && x && y, "gist_spmm_block_units_f32: null pointer");
    GIST_REQUIRE(d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= d && ldy >= d && aligned16(x) && aligned16(y) &&
                     aligned16(images),
                 "gist_spmm_block_units_f32: rows must be 16-byte aligned multiples of 4 floats");
    return launch_spmm_mfma_units(units, n_units, images, x, ldx, y, ldy, n_rows_y, d, out_scale, accumulate,
                                  as_stream(stream));
}

#ifdef MF_PROBE
extern "C" int gist_mf_probe_read(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(gist::g_mf_probe), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
