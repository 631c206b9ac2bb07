// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, one
// rounding per product -- the 1e-4 parity bar rules out bf16/fp8 inputs).
//
// Three operand layouts, all row-major with explicit leading dimensions:
//   NT  Y[m,n] = A[m,k] . W[n,k]^T + bias   forward of nn.Linear   (modules.py:233)
//   NN  Z[m,n] = G[m,k] . W[k,n]            dZ = dY . W
//   TN  D[m,n] = G[k,m]^T . A[k,n]          dW = dY^T . Z
//
// Block tile 128x128x32 (or 64x64x64 when the output has too few 128x128 tiles to fill and
// balance 256 CUs), 256 threads = 4 waves in 2x2, each wave 64x64 = 2x2 MFMA tiles of
// 32x32 (64 accumulator VGPRs; one tile for the 64x64 block).  Two LDS buffers, one barrier
// per k tile.  The main loop holds nothing but MFMAs, fragment reads, 8 LDS-DMA instructions
// and scalar ops: next to an fp32 MFMA stream a VALU op of the wave costs ~8.5 cycles of
// matrix-pipe time and a ds_write_b128 26-45 (profiles/r01_gemm_phase_trace.md), so
//   * full k tiles go global -> LDS by DMA (buffer_load_dwordx4 ... lds): the buffer
//     resource's base (SGPRs) carries tile origin + k advance, the thread's part is a 32-bit
//     offset computed once; no staging VGPRs, no ds_write;
//   * the k loop is unrolled by two, so the LDS buffer index is a compile-time constant and
//     every fragment address is an invariant VGPR + immediate;
//   * only the ragged last k tile of a split (and operands that are not 16-B aligned) goes
//     through registers (clamped loads, k mask, ds_write);
//   * the epilogue is raw buffer stores: row advance in the scalar offset, ragged columns
//     dropped by the hardware range check.
//
// LDS images (MI355X_MICROARCH.md section LDS):
//   k-contiguous operand  : [128 rows][32 k] (or [64][64]), no padding; the 16-byte chunk c of row r sits in
//                           slot c ^ (r & (chunks per row - 1)) of its row (a DMA instruction's 1 KiB lands
//                           contiguously; 8 consecutive rows put the same k chunk into 8
//                           different bank groups).  A lane reads ONE ds_read_b128 = 4
//                           consecutive k of its row.  Lanes 0-31 take k = 8q..8q+3, lanes 32-63
//                           k = 8q+4..8q+7, so MFMA step j of block q multiplies
//                           k = 8q + 4*(lane>>5) + j -- a permutation of k applied to BOTH
//                           operands, which a dot product allows.
//   m/n-contiguous operand: [32 k][128]; a lane reads its 4 k values as two
//                           ds_read2st64_b32; 32 consecutive lanes hit 32 consecutive banks.
//
// Shapes with few output tiles (n = 41 logits, dW of the last layer) are split along
// k across blockIdx.z into a workspace and reduced deterministically.
// Blocks are dealt to XCDs in 8x8 super-tiles so that the 64 blocks sharing an L2
// touch 8 A panels + 8 B panels.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace gist {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Block tiles: 128x128 (each wave 64x64 = 2x2 MFMA tiles) and 64x64 (each wave one 32x32
// MFMA tile) for outputs whose 128x128 tile count cannot fill / balance 256 CUs.
// k depth of a tile: 32 for the 128x128 block, 64 for the 64x64 block (half the barriers
// per flop where a SIMD holds a single wave and nothing hides them).
// XOR applied to the 16-byte chunk index of row r of a k-contiguous image.  16 consecutive rows (the rows of a
// ds_read_b128's lane groups) must land in 16 different 16-byte bank groups of the 256-byte bank space:
//   BK = 64 (256-byte rows: every row covers the whole space)  : r & 15 over the row's 16 chunks;
//   BK = 32 (128-byte rows: rows alternate between two halves) : (r >> 1) & 7 over its 8 chunks -- rows r and r + 1
//                                                               share the XOR and differ in the half.
// (r & 7 for both, as first written, put rows r and r + 8 on the same banks: SQ_LDS_BANK_CONFLICT = half of the LDS
// cycles on the 64-tile kernel, 6 % of all cycles on the 128-tile one.)
template <int BK> __device__ __forceinline__ constexpr int kc_swz(int r) { return BK == 64 ? (r & 15) : ((r >> 1) & 7); }

template <int R, int BK> struct Img {   // R = rows of a k-contiguous image = columns of an m/n image
    // k-contiguous image: R rows of BK floats, no padding; the 16-byte chunk c of row r sits in
    // slot c ^ (r & 7) of the row (XOR swizzle: 8 consecutive rows put the same k chunk into 8
    // different bank groups, and a row is a contiguous 4*BK bytes so LDS-DMA can fill it)
    static constexpr int KC = R * BK;       // floats
    static constexpr int MC = BK * R;
    static constexpr int ITERS = R * BK / 1024;   // float4 per thread per k tile (256 threads)
};

struct GemmArgs {
    const float *a; int64_t lda;
    const float *b; int64_t ldb;
    const float *bias;
    float *c; int64_t ldc;
    int m, n, k;
    int tiles_m, tiles_n;  // filled by the launcher for the chosen tile edge
    int k_per_split;       // multiple of 64 (both k depths divide it)
    int64_t split_stride;  // elements between split slabs (0 = write C directly)
    int setprio;
};

// ---- global -> register staging --------------------------------------------
// Two loaders per image.  The ALIGNED one (16-B aligned base, ld % 4 == 0) is branch
// free: hipcc turns a per-load `if (in range) load` into a branch plus an
// s_waitcnt per load, which serialises the eight loads of a k tile.  Instead
//   * the NON-reduction coordinate (row of a k-contiguous image, column of an
//     m/n-contiguous image) is CLAMPED to a readable address: whatever is loaded
//     there only ever reaches output rows/columns >= m/n, which are never stored;
//   * the REDUCTION coordinate k is masked with selects (0 beyond k_end), so nothing
//     out of range enters a dot product.
// A 16-B chunk that straddles the end of a row is still inside the row's pitch
// because ld >= round_up(extent, 4) whenever ld % 4 == 0.
// The generic loader (any alignment) keeps per-element guards.

// k-contiguous operand: element (r, kk) at p[r*ld + kk].  256 threads move
// 128 rows x 32 k = 1024 float4: thread t -> rows t/8 + 32 i, k chunk t%8.
template <int IT, int BK>
__device__ __forceinline__ void load_kc_aligned(const float *__restrict__ p, int64_t ld, int rows,
                                                int kdim, int row0, int k0, float4 (&st)[IT]) {
    constexpr int CPR = BK / 4;                    // float4 chunks per row
    const int t = threadIdx.x;
    const int kk = k0 + (t % CPR) * 4;
    const int kc = min(kk, (int)ld - 4);           // stays inside the row pitch
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int r = min(row0 + t / CPR + (256 / CPR) * i, rows - 1);
        st[i] = *reinterpret_cast<const float4 *>(p + (int64_t)r * ld + kc);
    }
}

// Reduction-dimension mask of the aligned loaders, applied when the registers are
// written to LDS -- i.e. AFTER the MFMA block -- so that the loads stay in flight
// under the MFMAs instead of being waited for right after issue.
template <int IT, int BK>
__device__ __forceinline__ void mask_kc(float4 (&st)[IT], int kdim, int k0) {
    const int kk = k0 + (threadIdx.x % (BK / 4)) * 4;
    if (kk + 3 < kdim) return;
    const bool k0ok = kk + 0 < kdim, k1ok = kk + 1 < kdim, k2ok = kk + 2 < kdim;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        st[i].x = k0ok ? st[i].x : 0.f;
        st[i].y = k1ok ? st[i].y : 0.f;
        st[i].z = k2ok ? st[i].z : 0.f;
        st[i].w = 0.f;
    }
}

template <int IT, int R>
__device__ __forceinline__ void mask_mc(float4 (&st)[IT], int kdim, int k0) {
    constexpr int CPR = R / 4;                      // float4 chunks per k row (cols / 4)
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int kk = k0 + threadIdx.x / CPR + (256 / CPR) * i;
        if (kk >= kdim) st[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int IT, int BK>
__device__ __forceinline__ void load_kc_generic(const float *__restrict__ p, int64_t ld, int rows,
                                                int kdim, int row0, int k0, float4 (&st)[IT]) {
    constexpr int CPR = BK / 4;
    const int t = threadIdx.x;
    const int kk = k0 + (t % CPR) * 4;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int r = row0 + t / CPR + (256 / CPR) * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows) {
            const float *src = p + (int64_t)r * ld + kk;
            if (kk + 0 < kdim) v.x = src[0];
            if (kk + 1 < kdim) v.y = src[1];
            if (kk + 2 < kdim) v.z = src[2];
            if (kk + 3 < kdim) v.w = src[3];
        }
        st[i] = v;
    }
}

template <bool ALIGNED, int IT, int BK>
__device__ __forceinline__ void load_kc(const float *__restrict__ p, int64_t ld, int rows, int kdim,
                                        int row0, int k0, float4 (&st)[IT]) {
    if constexpr (ALIGNED) load_kc_aligned<IT, BK>(p, ld, rows, kdim, row0, k0, st);
    else load_kc_generic<IT, BK>(p, ld, rows, kdim, row0, k0, st);
}

template <int IT, int BK>
__device__ __forceinline__ void store_kc(float *__restrict__ s, const float4 (&st)[IT]) {
    constexpr int CPR = BK / 4;
    const int t = threadIdx.x;
    const int kq = t % CPR;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int r = t / CPR + (256 / CPR) * i;
        *reinterpret_cast<float4 *>(s + r * BK + ((kq ^ kc_swz<BK>(r)) << 2)) = st[i];
    }
}

// m/n-contiguous operand: element (kk, c) at p[kk*ld + c].  BK k rows x R cols:
// thread t -> k rows t/CPR + (256/CPR) i, column chunk t % CPR, CPR = R/4.
template <int IT, int R>
__device__ __forceinline__ void load_mc_aligned(const float *__restrict__ p, int64_t ld, int cols,
                                                int kdim, int col0, int k0, float4 (&st)[IT]) {
    constexpr int CPR = R / 4;
    const int t = threadIdx.x;
    const int cq = min(col0 + (t % CPR) * 4, (int)ld - 4);   // clamp inside the row pitch
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int kk = k0 + t / CPR + (256 / CPR) * i;
        st[i] = *reinterpret_cast<const float4 *>(p + (int64_t)min(kk, kdim - 1) * ld + cq);
    }
}

template <int IT, int R>
__device__ __forceinline__ void load_mc_generic(const float *__restrict__ p, int64_t ld, int cols,
                                                int kdim, int col0, int k0, float4 (&st)[IT]) {
    constexpr int CPR = R / 4;
    const int t = threadIdx.x;
    const int cq = col0 + (t % CPR) * 4;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int kk = k0 + t / CPR + (256 / CPR) * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kk < kdim) {
            const float *src = p + (int64_t)kk * ld + cq;
            if (cq + 0 < cols) v.x = src[0];
            if (cq + 1 < cols) v.y = src[1];
            if (cq + 2 < cols) v.z = src[2];
            if (cq + 3 < cols) v.w = src[3];
        }
        st[i] = v;
    }
}

template <bool ALIGNED, int IT, int R>
__device__ __forceinline__ void load_mc(const float *__restrict__ p, int64_t ld, int cols, int kdim,
                                        int col0, int k0, float4 (&st)[IT]) {
    if constexpr (ALIGNED) load_mc_aligned<IT, R>(p, ld, cols, kdim, col0, k0, st);
    else load_mc_generic<IT, R>(p, ld, cols, kdim, col0, k0, st);
}

template <int IT, int R>
__device__ __forceinline__ void store_mc(float *__restrict__ s, const float4 (&st)[IT]) {
    constexpr int CPR = R / 4, C = R;
    const int t = threadIdx.x;
    const int cq = (t % CPR) * 4;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int kk = t / CPR + (256 / CPR) * i;
        *reinterpret_cast<float4 *>(s + kk * C + cq) = st[i];
    }
}

// ---- LDS-DMA staging for full interior k tiles (ALIGNED operands) ------------------------
// Next to an MFMA stream every VALU instruction of the same wave costs ~8.5 cycles of matrix
// pipe time and a ds_write_b128 26-45, SALU costs nothing (profiles/r01_gemm_phase_trace.md).
// So the per-thread part of a staging address is a 32-bit byte offset RELATIVE to the
// workgroup's tile origin, computed once, and the k advance lives in the wave-uniform base of
// a buffer resource (SGPRs, rebuilt per k tile by SALU).  Relative offsets stay below
// 128 rows x ld x 4 B, so they fit 32 bits for any ld < 2^22 (checked by the launcher).
// LDS-DMA (buffer_load_dwordx4 ... lds): one wave instruction moves 64 x 16 B from per-lane
// global addresses to 1 KiB of LDS at M0 + lane*16, without passing through VGPRs and without a
// ds_write (a ds_write_b128 next to MFMAs costs 26-45 cycles of matrix-pipe time).  A 16 KiB
// image is 16 such instructions, 4 per wave: instruction j = 4*wave + jj fills LDS bytes
// [1024 j, 1024 (j+1)) of the image, so the thread's source is whatever belongs there:
//   k-contiguous image : row (64j + lane) / CPR, slot (64j + lane) % CPR, k chunk slot ^ (row&7)
//   m/n-contiguous     : k row (64j + lane) / (R/4), column chunk (64j + lane) % (R/4)
template <int IT, int BK>
__device__ __forceinline__ void dma_offsets_kc(int64_t ld, int rows, int row0, int wave, int lane,
                                               uint32_t (&off)[IT]) {
    constexpr int CPR = BK / 4;
#pragma unroll
    for (int jj = 0; jj < IT; ++jj) {
        const int p = 64 * (IT * wave + jj) + lane;
        const int r = p / CPR, slot = p % CPR;
        const int dr = min(r, rows - 1 - row0);
        off[jj] = (uint32_t)(((int64_t)dr * ld + ((slot ^ kc_swz<BK>(r)) << 2)) * 4);
    }
}

template <int IT, int R>
__device__ __forceinline__ void dma_offsets_mc(int64_t ld, int col0, int wave, int lane,
                                               uint32_t (&off)[IT]) {
    constexpr int CPR = R / 4;
#pragma unroll
    for (int jj = 0; jj < IT; ++jj) {
        const int p = 64 * (IT * wave + jj) + lane;
        const int kk = p / CPR;
        const int dc = min(col0 + (p % CPR) * 4, (int)ld - 4) - col0;
        off[jj] = (uint32_t)(((int64_t)kk * ld + dc) * 4);
    }
}

template <int IT>
__device__ __forceinline__ void dma_image(const float *ubase, const uint32_t *off, float *image,
                                          int wave) {
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(ubase), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int jj = 0; jj < IT; ++jj)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrc, (__attribute__((address_space(3))) void *)(image + (IT * wave + jj) * 256), 16,
            off[jj], 0, 0, 0);
}

// ---- LDS -> fragment ---------------------------------------------------------
// The 4 values a lane feeds to MFMA steps j = 0..3 of k block q (k = 8q + 4hh .. +3) of the
// 32-row (or 32-column) slab whose row for this lane is `row`.
//   k-contiguous image: one ds_read_b128 at row*BK + ((2q+hh) ^ (row&7))*4; `kc_off[q&3]` holds
//   that offset for q = 0..3 (computed once: the XOR is lane dependent), q >= 4 (BK = 64) adds 32;
//   m/n-contiguous image [k][C]: 4 scalars C apart (paired along k into ds_read2st64_b32).
template <bool KC, int C, int BK>
__device__ __forceinline__ float4 read_frag(const float *__restrict__ s, int row, int q, int hh,
                                            const int (&kc_off)[4]) {
    if constexpr (KC) {
        // BK = 64: the XOR covers all 16 chunk slots of the 256-byte row, so the upper k half (q >= 4) of a row whose
        // bit 3 is set sits in the LOWER 32 floats: kc_off holds the q < 4 offsets, the others are those ^ 32 floats
        if constexpr (BK == 64) return *reinterpret_cast<const float4 *>(s + ((q >> 2) ? (kc_off[q & 3] ^ 32) : kc_off[q & 3]));
        else return *reinterpret_cast<const float4 *>(s + kc_off[q & 3]);
    } else {
        const float *p = s + (8 * q + 4 * hh) * C + row;
        return make_float4(p[0], p[C], p[2 * C], p[3 * C]);
    }
}

__device__ __forceinline__ float f4(const float4 &v, int j) {
    return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w));
}

#ifdef GIST_GEMM_TRACE   // dev builds only (scripts/gemm_trace.py): per-block phase timestamps
__device__ unsigned long long g_gemm_trace[12 * 16384];
#define GIST_TRACE(slot)                                                               \
    if (threadIdx.x == 0 && blockIdx.z == 0 && blockIdx.x < 16384)                     \
        g_gemm_trace[12 * blockIdx.x + (slot)] = wall_clock64();
// phase accounting of wave 0 inside the main loop, in s_memtime ticks
#define GIST_PHASE_DECL unsigned long long ph_t = clock64(), ph_acc[6] = {0, 0, 0, 0, 0, 0};
#define GIST_PHASE(i) { const unsigned long long n_ = clock64(); ph_acc[i] += n_ - ph_t; ph_t = n_; }
#else
#define GIST_TRACE(slot)
#define GIST_PHASE_DECL
#define GIST_PHASE(i)
#endif

// ALIGNED: both operands have 16-B aligned bases and leading dimensions % 4 == 0
// (every buffer the engine allocates); otherwise the generic guarded loader runs.
// T = block tile edge (128 or 64); 4 waves in 2x2, each wave (T/2)x(T/2) = (T/64)^2 MFMA tiles.
// The kernel's body as a device function of (arguments, tile index, k slice): gemm_f32_kernel runs it on its own grid,
// gemm_f32_dual_kernel (below) runs TWO products' tiles in one grid.
template <bool A_KC, bool B_KC, bool ALIGNED, int T>
__device__ __forceinline__ void gemm_f32_body(const GemmArgs &g, const int block_x, const int block_z) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    GIST_TRACE(0)
    constexpr int BK = T == 128 ? 32 : 64;
    constexpr int TA = A_KC ? Img<T, BK>::KC : Img<T, BK>::MC;
    constexpr int TB = B_KC ? Img<T, BK>::KC : Img<T, BK>::MC;
    constexpr int IT = Img<T, BK>::ITERS;
    constexpr int W = T / 2;          // wave tile edge
    constexpr int NT = W / 32;        // MFMA tiles per wave per dimension
    // buffer b: A image at smem + b*(TA+TB), B image right behind it

    // ---- block -> output tile, 8x8 super-tiles per XCD ------------------------
    const int nwg = g.tiles_m * g.tiles_n;
    const int orig = block_x;
    const int qd = nwg / kXcds, rm = nwg % kXcds, xcd = orig % kXcds;
    const int L = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + orig / kXcds;
    constexpr int GM = 8;
    const int width = GM * g.tiles_n;
    const int group = L / width;
    const int first_m = group * GM;
    const int gsz = min(g.tiles_m - first_m, GM);
    const int bm = first_m + (L % width) % gsz;
    const int bn = (L % width) / gsz;
    const int row0 = bm * T, col0 = bn * T;

    const int k_begin = block_z * g.k_per_split;
    const int k_end = min(g.k, k_begin + g.k_per_split);
    const int n_kt = (k_end - k_begin + BK - 1) / BK;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, hh = lane >> 5;

    f32x16 acc[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Staging.  ALIGNED operands: full k tiles go global -> LDS by DMA (no VGPRs, no ds_write, no
    // vector address arithmetic: the resource base in SGPRs carries tile origin + k advance, the
    // thread's part is a 32-bit offset computed once).  The ragged last k tile of a split, and
    // every tile of unaligned operands, takes the register path (clamped loads, k mask, ds_write).
    float4 stA[IT], stB[IT];
    uint32_t offA[IT], offB[IT];
    const float *originA = nullptr, *originB = nullptr;     // tile origin at k = 0 (uniform)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    if constexpr (ALIGNED) {
        if constexpr (A_KC) { dma_offsets_kc<IT, BK>(g.lda, g.m, row0, wave_u, lane, offA); originA = g.a + (int64_t)row0 * g.lda; }
        else { dma_offsets_mc<IT, T>(g.lda, row0, wave_u, lane, offA); originA = g.a + row0; }
        if constexpr (B_KC) { dma_offsets_kc<IT, BK>(g.ldb, g.n, col0, wave_u, lane, offB); originB = g.b + (int64_t)col0 * g.ldb; }
        else { dma_offsets_mc<IT, T>(g.ldb, col0, wave_u, lane, offB); originB = g.b + col0; }
    }
    auto full_tile = [&](int kt) { return ALIGNED && k_begin + (kt + 1) * BK <= k_end; };
    auto dma = [&](int buf, int kt) {
        const int k0 = k_begin + kt * BK;
        float *sa = smem + buf * (TA + TB);
        dma_image<IT>(originA + (A_KC ? (int64_t)k0 : (int64_t)k0 * g.lda), offA, sa, wave_u);
        dma_image<IT>(originB + (B_KC ? (int64_t)k0 : (int64_t)k0 * g.ldb), offB, sa + TA, wave_u);
    };
    auto gload = [&](int kt) {
        const int k0 = k_begin + kt * BK;
        if constexpr (A_KC) load_kc<ALIGNED, IT, BK>(g.a, g.lda, g.m, k_end, row0, k0, stA);
        else load_mc<ALIGNED, IT, T>(g.a, g.lda, g.m, k_end, row0, k0, stA);
        if constexpr (B_KC) load_kc<ALIGNED, IT, BK>(g.b, g.ldb, g.n, k_end, col0, k0, stB);
        else load_mc<ALIGNED, IT, T>(g.b, g.ldb, g.n, k_end, col0, k0, stB);
    };
    auto sstore = [&](int buf, int kt) {
        float *sa = smem + buf * (TA + TB);
        float *sb = sa + TA;
        if constexpr (ALIGNED) {
            const int k0 = k_begin + kt * BK;
            if constexpr (A_KC) mask_kc<IT, BK>(stA, k_end, k0); else mask_mc<IT, T>(stA, k_end, k0);
            if constexpr (B_KC) mask_kc<IT, BK>(stB, k_end, k0); else mask_mc<IT, T>(stB, k_end, k0);
        }
        if constexpr (A_KC) store_kc<IT, BK>(sa, stA); else store_mc<IT, T>(sa, stA);
        if constexpr (B_KC) store_kc<IT, BK>(sb, stB); else store_mc<IT, T>(sb, stB);
    };

    if (n_kt > 0) {
        if (full_tile(0)) {
            dma(0, 0);
        } else {
            gload(0);
            sstore(0, 0);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);         // vmcnt(0): the DMA has landed in LDS
    __syncthreads();
    GIST_TRACE(1)
    GIST_PHASE_DECL

    // Fragment addresses.  k-contiguous image: offset of k block q = 0..3 for each slab's row
    // (the XOR swizzle is lane dependent, so these are computed once).  m/n-contiguous image:
    // the slab's row, made opaque to the compiler -- otherwise it pairs the two slabs' reads
    // into ds_read2_b32 and needs a v_add per two k values for the offsets that do not fit;
    // unrelated bases pair along k instead (ds_read2st64_b32, immediates only).
    int arow[NT], brow[NT];
    int akc[NT][4], bkc[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        arow[i] = wm * W + i * 32 + r;
        brow[i] = wn * W + i * 32 + r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            akc[i][q] = arow[i] * BK + (((2 * q + hh) ^ kc_swz<BK>(arow[i])) << 2);
            bkc[i][q] = brow[i] * BK + (((2 * q + hh) ^ kc_swz<BK>(brow[i])) << 2);
        }
        asm volatile("" : "+v"(arow[i]));
        asm volatile("" : "+v"(brow[i]));
    }
    // One k step; the LDS buffer index is a compile-time constant (the loop below is unrolled by
    // two), so every LDS address of the step is an invariant VGPR plus an immediate and the
    // buffer toggle costs no vector instruction.
    // mode 0: the next tile is staged by DMA; 1: through registers (ragged / unaligned);
    // 2: there is no next tile.  Compile-time, so the steady-state loop (mode 0) carries no
    // staging registers at all.
    auto kstep = [&](auto cur_c, auto mode_c, int kt) {
        constexpr int cur = decltype(cur_c)::value;
        constexpr int mode = decltype(mode_c)::value;
        if constexpr (mode == 0) dma(cur ^ 1, kt + 1);      // buffer cur^1 is free since the last barrier
        if constexpr (mode == 1) gload(kt + 1);
        __builtin_amdgcn_sched_barrier(0);      // staging is issued; keep the MFMAs below
        GIST_PHASE(0)
        const float *a_s = smem + cur * (TA + TB);
        const float *b_s = a_s + TA;
        if (g.setprio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            float4 af[NT], bf[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) af[i] = read_frag<A_KC, T, BK>(a_s, arow[i], q, hh, akc[i]);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = read_frag<B_KC, T, BK>(b_s, brow[j], q, hh, bkc[j]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            f4(af[i], s), f4(bf[j], s), acc[i][j], 0, 0, 0);
        }
        if (g.setprio) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);      // nothing of the store phase moves above the MFMAs
        GIST_PHASE(1)
        if constexpr (mode != 2) __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): DMA landed / loads arrived
        __builtin_amdgcn_sched_barrier(0);
        GIST_PHASE(2)
        if constexpr (mode == 1) sstore(cur ^ 1, kt + 1);
#ifdef GIST_GEMM_TRACE
        __builtin_amdgcn_sched_barrier(0);
        GIST_PHASE(3)
#endif
        __syncthreads();
        GIST_PHASE(4)
    };
    {
        using C0 = std::integral_constant<int, 0>;
        using C1 = std::integral_constant<int, 1>;
        using C2 = std::integral_constant<int, 2>;
        int kt = 0;
        if constexpr (ALIGNED) {
            // steps whose next tile is a full one: only the last tile of a split can be ragged
            const int n_dma = n_kt - 1 - (n_kt > 0 && !full_tile(n_kt - 1) ? 1 : 0);
            for (; kt + 1 < n_dma; kt += 2) {
                kstep(C0{}, C0{}, kt);
                kstep(C1{}, C0{}, kt + 1);
            }
        }
        for (; kt < n_kt; ++kt) {                // the last (<= 3) steps, or every step if unaligned
            const int mode = kt + 1 >= n_kt ? 2 : (full_tile(kt + 1) ? 0 : 1);
            if ((kt & 1) == 0) {
                if (mode == 0) kstep(C0{}, C0{}, kt); else if (mode == 1) kstep(C0{}, C1{}, kt); else kstep(C0{}, C2{}, kt);
            } else {
                if (mode == 0) kstep(C1{}, C0{}, kt); else if (mode == 1) kstep(C1{}, C1{}, kt); else kstep(C1{}, C2{}, kt);
            }
        }
    }
    GIST_TRACE(2)

    // ---- epilogue: C/D layout col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5) ----
    // A workgroup usually stores its tile while the CU's other workgroup streams MFMAs, and
    // vector ALU work then only issues in the gaps of that stream (a 64-element epilogue took
    // up to 60 us).  So the stores carry no vector arithmetic: raw buffer stores whose resource
    // covers exactly the valid rows of this tile, per-lane column offset computed once (columns
    // >= n get an out-of-range offset and are dropped by the hardware range check), and the
    // row advance in a scalar offset.
    float *cbase = g.c + (int64_t)block_z * g.split_stride + (int64_t)row0 * g.ldc + col0;
    const bool add_bias = g.bias != nullptr && g.split_stride == 0;
    const int rows_valid = min(g.m - row0, T);
    __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
        cbase, 0, (int)((int64_t)rows_valid * g.ldc * 4), 0x00020000);
    const uint32_t ldc_b = (uint32_t)g.ldc * 4;
    uint32_t cvoff[NT];
    float bv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int cl = wn * W + j * 32 + r;                       // column inside the tile
        const bool ok = col0 + cl < g.n;
        cvoff[j] = ok ? (uint32_t)(wm * W + 4 * hh) * ldc_b + (uint32_t)cl * 4 : 0x7fffffffu;
        bv[j] = (add_bias && ok) ? g.bias[col0 + cl] : 0.f;
    }
    if (add_bias) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] += bv[j];
    }
    // The hardware range check covers the VECTOR offset only (the scalar offset is added after
    // it), so the row advance may ride in the scalar offset only when every row of the tile is
    // valid; the last row tile of a ragged m adds it to the vector offset instead.
    if (rows_valid == T) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t soff = (uint32_t)(i * 32 + (e & 3) + 8 * (e >> 2)) * ldc_b;   // uniform
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float v = acc[i][j][e];       // (bit_cast of the vector element itself
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), crsrc, cvoff[j], soff, 0);
                }                                       //  picks element 0: go through a scalar)
            }
    } else {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t roff = (uint32_t)(i * 32 + (e & 3) + 8 * (e >> 2)) * ldc_b;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float v = acc[i][j][e];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), crsrc, cvoff[j] + roff, 0, 0);
                }
            }
    }
#ifdef GIST_GEMM_TRACE
    __builtin_amdgcn_s_waitcnt(0);          // this wave's stores are acknowledged
    GIST_TRACE(3)
    if (threadIdx.x == 0 && blockIdx.z == 0 && blockIdx.x < 16384) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_gemm_trace[12 * blockIdx.x + 4] = ((unsigned long long)xcc << 32) | hw;
        for (int i = 0; i < 5; ++i) g_gemm_trace[12 * blockIdx.x + 5 + i] = ph_acc[i];
        g_gemm_trace[12 * blockIdx.x + 10] = (unsigned long long)n_kt;
    }
#endif
}

template <bool A_KC, bool B_KC, bool ALIGNED, int T>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
    gemm_f32_body<A_KC, B_KC, ALIGNED, T>(g, (int)blockIdx.x, (int)blockIdx.z);
}

// Two INDEPENDENT products in one launch (round 4): the backward of a hidden layer needs dZ = dY . W (NN) and
// dW = dY^T . Z (TN, k split into slabs for the optimiser) from the same dY; at the per-rank widths each of them fills
// a fraction of the chip for 10-22 us (128-512 tiles of 64 x 64, one 32 x 32 MFMA tile per wave), and one after the
// other they cost their sum.  Workgroups [0, n1) run the first product's tiles, the rest the second's (tile index and
// k slice flattened into blockIdx.x); both are the fp32 kernel's own body, so the results are bit-identical to the
// two separate launches.
template <int T>
__global__ __launch_bounds__(256, 2) void gemm_f32_dual_kernel(GemmArgs g1, GemmArgs g2, int n1, int tiles1, int tiles2) {
    const int b = (int)blockIdx.x;
    if (b < n1) gemm_f32_body<true, false, true, T>(g1, b % tiles1, b / tiles1);
    else gemm_f32_body<false, false, true, T>(g2, (b - n1) % tiles2, (b - n1) / tiles2);
}

// C[m,n] = sum_s slab_s[m,n] + bias[n]; slabs are dense [m][n].
__global__ void splitk_reduce_kernel(const float *__restrict__ ws, int64_t slab, int splits,
                                     const float *__restrict__ bias, float *__restrict__ c,
                                     int64_t ldc, int m, int n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)m * n) return;
    const int rr = (int)(i / n), cc = (int)(i % n);
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[k * slab + i];
    if (bias) s += bias[cc];
    c[(int64_t)rr * ldc + cc] = s;
}

int splitk_reduce(const char *name, const float *slabs, int64_t slab, int splits, const float *bias, float *c,
                  int64_t ldc, int64_t m, int64_t n, hipStream_t st) {
    const int64_t total = m * n;
    if (total <= 0) return GIST_OK;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, slabs,
                       slab, splits, bias, c, ldc, (int)m, (int)n);
    return launch_status(name);
}

// ---- tile / split-K choice ---------------------------------------------------------------
// Everything here is fp32 MFMA work, so time ~ MFMA work of the busiest SIMD.  A block is 4
// waves (one per SIMD of a CU); blocks are dealt round-robin to 256 CUs, a CU keeps up to
// `cap` of them resident (128-tile: 2, 64-tile: 4) and fewer co-resident waves hide less
// latency.  Model, in units of one 64x64x32 MFMA block (16 MFMAs, ~0.43 us), calibrated on
// scripts/gemm_sweep.py measurements (MI355X, 34 shapes, within ~10% of the best config):
//   per_cu  = ceil(blocks / 256)
//   cost    = per_cu * unit(tile) * (k_tiles_per_split + 3) / eff(min(cap, per_cu))
//           + [splits > 1] * (12 + 6.5 * splits * m*n/1e6)      (slab traffic + reduce launch)
struct GemmCfg { int tile; int splits; };

// gemm_h3.hip: the f16x3 split path (fp32-accurate products on the f16 matrix cores)
int h3_gemm(const char *name, bool a_kc, bool b_kc, const float *a, int64_t lda, const float *b,
            int64_t ldb, const float *bias, float *c, int64_t ldc, int64_t m, int64_t n, int64_t k,
            void *ws, int64_t ws_bytes, hipStream_t st);
int64_t h3_workspace_bytes(int64_t m, int64_t n, int64_t k);
// gemm_b3.hip: the bf16x3 split path (all 24 operand bits on the bf16 matrix cores)
int b3_gemm(const char *name, bool a_kc, bool b_kc, const float *a, int64_t lda, const float *b,
            int64_t ldb, const float *bias, float *c, int64_t ldc, int64_t m, int64_t n, int64_t k,
            void *ws, int64_t ws_bytes, hipStream_t st);
int64_t b3_workspace_bytes(int64_t m, int64_t n, int64_t k);
// gemm_b3c.hip: the bf16x3 path that converts on load (shapes below the pre-split path's thresholds)
int b3c_gemm(const char *name, bool a_kc, bool b_kc, const float *a, int64_t lda, const float *b,
             int64_t ldb, const float *bias, float *c, int64_t ldc, int64_t m, int64_t n, int64_t k,
             void *ws, int64_t ws_bytes, hipStream_t st, int *deferred);
int64_t b3c_slab_bytes(int64_t m, int64_t n, int64_t k, bool tail = true);      // -1: shape not taken

// deferred: the k slices are summed by the kernel that consumes the result (gist_gemm_slabs_f32) -- no reduce
// launch, only the slabs' write and read.
static GemmCfg choose_cfg(int64_t m, int64_t n, int64_t k, bool deferred = false) {
    const int64_t kt = ceil_div(k, 32);      // the model counts k in units of 32
    // explicit override for tuning sweeps (scripts/gemm_sweep.py) and tests; 0 = the model decides
    const int t_tile = (int)tune(GIST_TUNE_GEMM_TILE), t_split = (int)tune(GIST_TUNE_GEMM_SPLITS);
    if (t_tile && t_split) {
        int sp = t_split;
        while (sp > 1 && kt / sp < 1) sp >>= 1;
        return GemmCfg{t_tile == 64 ? 64 : 128, sp < 1 ? 1 : sp};
    }
    static const double eff128[3] = {0.0, 0.90, 1.00};
    static const double eff64[5] = {0.0, 0.60, 0.80, 0.92, 1.00};
    GemmCfg best{128, 1};
    double best_cost = 1e300;
    const double mn = (double)m * (double)n / 1e6;
    for (int tile : {128, 64}) {
        const int64_t tiles = ceil_div(m, tile) * ceil_div(n, tile);
        const int cap = tile == 128 ? 2 : 4;
        const double unit = tile == 128 ? 4.0 : 1.12;
        for (int sp : {1, 2, 4, 8, 16, 32}) {
            if (sp > 1 && kt / sp < 2) break;
            const int64_t per_cu = ceil_div(tiles * sp, 256);
            const int64_t kt_per = ceil_div(kt, sp);
            const int conc = (int)(per_cu < cap ? per_cu : cap);
            const double eff = tile == 128 ? eff128[conc] : eff64[conc];
            double cost = (double)per_cu * unit * (double)(kt_per + 3) / eff;
            // (deferred: no reduce launch, but the slabs are written by this kernel and read by the consumer; fitted to
            // same-box A/B runs of the forward projections of BASELINE configs 2 and 4, scripts/ab_yslabs.sh)
            if (sp > 1) cost += deferred ? 4.5 + 9.0 * sp * mn : 12.0 + 6.5 * sp * mn;
            if (cost < best_cost) { best_cost = cost; best = GemmCfg{tile, sp}; }
        }
    }
    return best;
}

template <bool A_KC, bool B_KC, int T>
static int launch_tile(const char *name, GemmArgs &g, bool aligned, int splits, hipStream_t st) {
    constexpr int BK = T == 128 ? 32 : 64;
    constexpr int TA = A_KC ? Img<T, BK>::KC : Img<T, BK>::MC;
    constexpr int TB = B_KC ? Img<T, BK>::KC : Img<T, BK>::MC;
    const size_t smem = (size_t)2 * (TA + TB) * sizeof(float);
    static DeviceOnce once;       // one per template instance
    int dev;
    if (once.needed(&dev)) {
        hipError_t e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&gemm_f32_kernel<A_KC, B_KC, true, T>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(
                reinterpret_cast<const void *>(&gemm_f32_kernel<A_KC, B_KC, false, T>),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
            return GIST_ELAUNCH;
        }
        once.done(dev);
    }
    g.tiles_m = (int)ceil_div(g.m, T);
    g.tiles_n = (int)ceil_div(g.n, T);
    const dim3 grid((unsigned)(g.tiles_m * g.tiles_n), 1, (unsigned)splits);
    if (aligned)
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, true, T>), grid, dim3(256), smem, st, g);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, false, T>), grid, dim3(256), smem, st, g);
    return launch_status(name);
}

// deferred != NULL: a split-K call leaves its partial sums as dense slabs [splits][m][n] at ws and does
// NOT reduce them (the consumer sums them in slab order, + bias): *deferred = the slab count, or 1 when
// c holds the finished result (bias included).
template <bool A_KC, bool B_KC>
static int launch_gemm(const char *name, const float *a, int64_t lda, const float *b, int64_t ldb,
                       const float *bias, float *c, int64_t ldc, int64_t m, int64_t n, int64_t k,
                       void *ws, int64_t ws_bytes, hipStream_t st, int *deferred = nullptr) {
    if (deferred) *deferred = 1;
    if (m < 0 || n < 0 || k < 0) { set_error("%s: negative size", name); return GIST_EINVAL; }
    if (m == 0 || n == 0) return GIST_OK;
    if (!a || !b || !c) { set_error("%s: null pointer", name); return GIST_EINVAL; }
    if (m >= (1LL << 31) || n >= (1LL << 31) || k >= (1LL << 31)) {
        set_error("%s: size >= 2^31", name); return GIST_EINVAL;
    }
    // tile-relative byte offsets are 32-bit (DMA staging, buffer-store epilogue): 128 rows x ld x 4
    if (lda >= (1LL << 22) || ldb >= (1LL << 22) || ldc >= (1LL << 22)) {
        set_error("%s: leading dimension >= 2^22 elements", name); return GIST_EINVAL;
    }
    {   // large, chip-filling shapes: split operands + f16 MFMA (mode 1); 0 = not taken
        int rc = h3_gemm(name, A_KC, B_KC, a, lda, b, ldb, bias, c, ldc, m, n, k, ws, ws_bytes, st);
        if (rc != 0) return rc < 0 ? rc : GIST_OK;
        rc = b3_gemm(name, A_KC, B_KC, a, lda, b, ldb, bias, c, ldc, m, n, k, ws, ws_bytes, st);   // mode 2
        if (rc != 0) return rc < 0 ? rc : GIST_OK;
        rc = b3c_gemm(name, A_KC, B_KC, a, lda, b, ldb, bias, c, ldc, m, n, k, ws, ws_bytes, st, deferred);
        if (rc != 0) return rc < 0 ? rc : GIST_OK;                                                   // mode 2, small
    }
    GemmArgs g;
    g.a = a; g.lda = lda; g.b = b; g.ldb = ldb; g.bias = bias; g.c = c; g.ldc = ldc;
    g.m = (int)m; g.n = (int)n; g.k = (int)k;
    const bool aligned = aligned16(a) && (lda % 4 == 0) && lda >= 4 && aligned16(b) &&
                         (ldb % 4 == 0) && ldb >= 4 && k > 0;
    g.setprio = 1;
    GemmCfg cfg = choose_cfg(m, n, k, deferred != nullptr);
    int splits = cfg.splits;
    // a slab buffer too small for the model's slice count: the largest power of two that fits, not one slice
    // (the class layer's 41 x 4096 x 2046 weight gradient fell from 8 slices to 1 for most batch sizes of the
    // h = 2048 step -- 39 us instead of 13 -- because the count is not monotone in k and the step had sized its
    // buffer from a few sampled batch sizes)
    while (splits > 1 && (ws == nullptr || ws_bytes < (int64_t)splits * m * n * 4)) splits >>= 1;
    g.k_per_split = (int)(ceil_div(ceil_div(k, 64), splits) * 64);   // multiple of either BK
    splits = (int)ceil_div(k, g.k_per_split > 0 ? g.k_per_split : 1);
    if (splits < 1) splits = 1;
    if (k == 0) { g.k_per_split = 64; splits = 1; }
    if (splits == 1) {
        g.split_stride = 0;
    } else {
        g.c = static_cast<float *>(ws);
        g.ldc = n;
        g.split_stride = m * n;
        g.bias = nullptr;
    }
    int rc = cfg.tile == 128 ? launch_tile<A_KC, B_KC, 128>(name, g, aligned, splits, st)
                             : launch_tile<A_KC, B_KC, 64>(name, g, aligned, splits, st);
    if (rc || splits == 1) return rc;
    if (deferred) { *deferred = splits; return rc; }
    const int64_t total = m * n;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0,
                       st, static_cast<const float *>(ws), m * n, splits, bias, c, ldc, (int)m,
                       (int)n);
    return launch_status(name);
}

// dz[m, n1] = dy[m, k1] . w[k1, n1]  (NN)   and   dW[k1, n1] = dy[m, k1]^T . z[m, n1]  (TN: k = m rows, left as
// *n_slabs dense slabs [k1][n1] at `slabs` for the consumer, or written to dw when one slice) in ONE launch.
// Taken (gemm_dual_takes) when both run on the fp32 kernel's 64 x 64 tiles anyway and together fit ~one round of the
// chip's workgroup slots: the per-rank widths <= 512 and config 2.
static bool dual_shapes_ok(int64_t m, int64_t n1, int64_t k1, int *splits_out) {
    if (m <= 0 || n1 <= 0 || k1 <= 0 || (int)tune(GIST_TUNE_GEMM_DUAL) == 1) return false;
    if (tune(GIST_TUNE_GEMM_TILE) != 0.0 || tune(GIST_TUNE_GEMM_SPLITS) != 0.0) return false;
    if (h3_eligible(m, n1, k1) || b3_eligible(m, n1, k1) || h3_eligible(k1, n1, m) || b3_eligible(k1, n1, m)) return false;
    const GemmCfg c1 = choose_cfg(m, n1, k1), c2 = choose_cfg(k1, n1, m, true);
    if (c1.tile != 64 || c2.tile != 64 || c1.splits != 1) return false;
    const int64_t blocks = ceil_div(m, 64) * ceil_div(n1, 64) + ceil_div(k1, 64) * ceil_div(n1, 64) * c2.splits;
    if (blocks > 1280) return false;           // (2 workgroups per CU resident: beyond ~2.5 rounds each product fills the chip alone)
    if (splits_out) *splits_out = c2.splits;
    return true;
}

bool gemm_dual_takes(int64_t m, int64_t n1, int64_t k1, int64_t lddy, int64_t ldw, int64_t ldz, int64_t lddz,
                     const float *dy, const float *w, const float *z, const float *dz) {
    return dual_shapes_ok(m, n1, k1, nullptr) && lddy % 4 == 0 && ldw % 4 == 0 && ldz % 4 == 0 && lddz % 4 == 0 &&
           lddy >= k1 && ldw >= n1 && ldz >= n1 && lddz >= n1 && lddy < (1LL << 22) && ldw < (1LL << 22) &&
           ldz < (1LL << 22) && lddz < (1LL << 22) && m < (1LL << 31) && aligned16(dy) && aligned16(w) && aligned16(z) &&
           aligned16(dz);
}

int gemm_dual_nn_tn(const char *name, const float *dy, int64_t lddy, const float *w, int64_t ldw, float *dz,
                    int64_t lddz, const float *z, int64_t ldz, float *dw, int64_t lddw, int64_t m, int64_t n1,
                    int64_t k1, void *slabs, int64_t slab_bytes, int *n_slabs, hipStream_t st) {
    int splits = 1;
    if (!dual_shapes_ok(m, n1, k1, &splits) || !gemm_dual_takes(m, n1, k1, lddy, ldw, ldz, lddz, dy, w, z, dz)) {
        set_error("%s: shape not taken (gist_gemm_dual_takes)", name);
        return GIST_EINVAL;
    }
    if (!dw || !n_slabs || lddw < n1) { set_error("%s: bad dW / n_slabs", name); return GIST_EINVAL; }
    GemmArgs g1{}, g2{};
    g1.a = dy; g1.lda = lddy; g1.b = w; g1.ldb = ldw; g1.bias = nullptr; g1.c = dz; g1.ldc = lddz;
    g1.m = (int)m; g1.n = (int)n1; g1.k = (int)k1; g1.setprio = 1;
    g1.k_per_split = (int)(ceil_div(k1, 64) * 64); g1.split_stride = 0;
    g1.tiles_m = (int)ceil_div(m, 64); g1.tiles_n = (int)ceil_div(n1, 64);
    g2.a = dy; g2.lda = lddy; g2.b = z; g2.ldb = ldz; g2.bias = nullptr; g2.setprio = 1;
    g2.m = (int)k1; g2.n = (int)n1; g2.k = (int)m;
    while (splits > 1 && (slabs == nullptr || slab_bytes < (int64_t)splits * k1 * n1 * 4)) splits >>= 1;
    g2.k_per_split = (int)(ceil_div(ceil_div(m, 64), splits) * 64);
    splits = (int)ceil_div(m, g2.k_per_split > 0 ? g2.k_per_split : 1);
    if (splits < 1) splits = 1;
    if (splits == 1) { g2.c = dw; g2.ldc = lddw; g2.split_stride = 0; }
    else { g2.c = static_cast<float *>(slabs); g2.ldc = n1; g2.split_stride = k1 * n1; }
    g2.tiles_m = (int)ceil_div(k1, 64); g2.tiles_n = (int)ceil_div(n1, 64);
    constexpr int T = 64, BK = 64;
    constexpr size_t smem = (size_t)2 * (Img<T, BK>::KC + Img<T, BK>::MC) * sizeof(float);   // (KC == MC: either body fits)
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_f32_dual_kernel<64>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e)); return GIST_ELAUNCH; }
        once.done(dev);
    }
    const int tiles1 = g1.tiles_m * g1.tiles_n, tiles2 = g2.tiles_m * g2.tiles_n;
    const int n1b = tiles1;
    hipLaunchKernelGGL((gemm_f32_dual_kernel<64>), dim3((unsigned)(n1b + tiles2 * splits)), dim3(256), smem, st, g1, g2,
                       n1b, tiles1, tiles2);
    *n_slabs = splits;
    return launch_status(name);
}

void gemm_f32_choice(int64_t m, int64_t n, int64_t k, int *tile, int *splits) {
    const GemmCfg c = choose_cfg(m, n, k);
    *tile = c.tile;
    *splits = c.splits;
}

// slab bytes of the split-K choice for this shape on the fp32 kernel or the convert-on-load bf16x3 kernel,
// whichever is larger (0: one k slice)
int64_t gemm_f32_slab_bytes(int64_t m, int64_t n, int64_t k, bool tn) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const int sp = choose_cfg(m, n, k, true).splits;      // (sized for the deferred form: what these slabs are for)
    const int64_t f32 = sp > 1 ? (int64_t)sp * m * n * 4 : 0;
    const int64_t c3 = b3c_slab_bytes(m, n, k, !tn);
    return c3 > f32 ? c3 : f32;
}

int gemm_slabs(int layout, const float *a, int64_t lda, const float *b, int64_t ldb, const float *bias, float *c,
               int64_t ldc, int64_t m, int64_t n, int64_t k, void *slabs, int64_t slab_bytes, int *n_slabs,
               hipStream_t st) {
    switch (layout) {
        case 0:
            GIST_REQUIRE(lda >= k && ldb >= k && ldc >= n, "gist_gemm_slabs_f32: leading dimension too small");
            return launch_gemm<true, true>("gist_gemm_slabs_f32", a, lda, b, ldb, bias, c, ldc, m, n, k, slabs,
                                           slab_bytes, st, n_slabs);
        case 1:
            GIST_REQUIRE(lda >= k && ldb >= n && ldc >= n, "gist_gemm_slabs_f32: leading dimension too small");
            return launch_gemm<true, false>("gist_gemm_slabs_f32", a, lda, b, ldb, bias, c, ldc, m, n, k, slabs,
                                            slab_bytes, st, n_slabs);
        case 2:
            GIST_REQUIRE(lda >= m && ldb >= n && ldc >= n, "gist_gemm_slabs_f32: leading dimension too small");
            return launch_gemm<false, false>("gist_gemm_slabs_f32", a, lda, b, ldb, bias, c, ldc, m, n, k, slabs,
                                             slab_bytes, st, n_slabs);
    }
    set_error("gist_gemm_slabs_f32: layout must be 0 (NT), 1 (NN) or 2 (TN)");
    return GIST_EINVAL;
}

}  // namespace gist

extern "C" int gist_gemm_dual_takes(int64_t m, int64_t n, int64_t k, int64_t lddy, int64_t ldw, int64_t ldz,
                                    int64_t lddz, const float *dy, const float *w, const float *z, const float *dz) {
    return gist::gemm_dual_takes(m, n, k, lddy, ldw, ldz, lddz, dy, w, z, dz) ? 1 : 0;
}

extern "C" int gist_gemm_nn_tn_dual_f32(const float *dy, int64_t lddy, const float *w, int64_t ldw, float *dz,
                                        int64_t lddz, const float *z, int64_t ldz, float *dw, int64_t lddw, int64_t m,
                                        int64_t n, int64_t k, void *slabs, int64_t slab_bytes, int32_t *n_slabs,
                                        gist_stream_t stream) {
    GIST_REQUIRE(dy && w && dz && z && dw && n_slabs, "gist_gemm_nn_tn_dual_f32: null pointer");
    int ns = 1;
    const int rc = gist::gemm_dual_nn_tn("gist_gemm_nn_tn_dual_f32", dy, lddy, w, ldw, dz, lddz, z, ldz, dw, lddw, m, n, k,
                                         slabs, slab_bytes, &ns, gist::as_stream(stream));
    *n_slabs = ns;
    return rc;
}

extern "C" int gist_gemm_slabs_f32(int layout, const float *a, int64_t lda, const float *b, int64_t ldb,
                                   const float *bias, float *c, int64_t ldc, int64_t m, int64_t n, int64_t k,
                                   void *slabs, int64_t slab_bytes, int32_t *n_slabs, gist_stream_t stream) {
    GIST_REQUIRE(n_slabs != nullptr, "gist_gemm_slabs_f32: null n_slabs");
    int ns = 1;
    const int rc = gist::gemm_slabs(layout, a, lda, b, ldb, bias, c, ldc, m, n, k, slabs, slab_bytes, &ns,
                                    gist::as_stream(stream));
    *n_slabs = ns;
    return rc;
}

#ifdef GIST_GEMM_TRACE
extern "C" int gist_gemm_trace_read(unsigned long long *out, int64_t n_blocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gist::g_gemm_trace), n_blocks * 12 * 8);
}
#endif

/* 1 if a gist_gemm_* call of this shape splits its own operands in the current mode (f16x3 / bf16x3 pre-split
 * kernels: it needs the large workspace of gist_gemm_workspace_bytes and reduces its k slices itself). */
extern "C" int gist_gemm_splits_operands(int64_t m, int64_t n, int64_t k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    return (gist::h3_eligible(m, n, k) || gist::b3_eligible(m, n, k)) ? 1 : 0;
}

extern "C" int64_t gist_gemm_workspace_bytes(int64_t m, int64_t n, int64_t k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const int64_t h3 = gist::h3_workspace_bytes(m, n, k);
    if (h3 > 0) return h3;
    const int64_t b3 = gist::b3_workspace_bytes(m, n, k);
    if (b3 > 0) return b3;
    const int64_t c3 = gist::b3c_slab_bytes(m, n, k);      // >= 0: the convert-on-load bf16x3 path takes the shape
    const int s0 = gist::choose_cfg(m, n, k).splits, s1 = gist::choose_cfg(m, n, k, true).splits;
    const int s = s0 > s1 ? s0 : s1;                       // (the call that reduces itself / gist_gemm_slabs_f32)
    const int64_t f32 = s > 1 ? (int64_t)s * m * n * 4 : 0;
    return c3 > f32 ? c3 : f32;                            // (unaligned operands fall back to the fp32 kernel)
}

extern "C" int gist_gemm_nt_f32(const float *a, int64_t lda, const float *w, int64_t ldw,
                                const float *bias, float *y, int64_t ldy, int64_t m, int64_t n,
                                int64_t k, void *workspace, int64_t workspace_bytes,
                                gist_stream_t stream) {
    GIST_REQUIRE(lda >= k && ldw >= k && ldy >= n, "gist_gemm_nt_f32: leading dimension too small");
    return gist::launch_gemm<true, true>("gist_gemm_nt_f32", a, lda, w, ldw, bias, y, ldy, m, n, k,
                                         workspace, workspace_bytes, gist::as_stream(stream));
}

extern "C" int gist_gemm_nn_f32(const float *g, int64_t ldg, const float *w, int64_t ldw, float *z,
                                int64_t ldz, int64_t m, int64_t n, int64_t k, void *workspace,
                                int64_t workspace_bytes, gist_stream_t stream) {
    GIST_REQUIRE(ldg >= k && ldw >= n && ldz >= n, "gist_gemm_nn_f32: leading dimension too small");
    return gist::launch_gemm<true, false>("gist_gemm_nn_f32", g, ldg, w, ldw, nullptr, z, ldz, m, n,
                                          k, workspace, workspace_bytes, gist::as_stream(stream));
}

extern "C" int gist_gemm_tn_f32(const float *g, int64_t ldg, const float *a, int64_t lda, float *d,
                                int64_t ldd, int64_t m, int64_t n, int64_t k, void *workspace,
                                int64_t workspace_bytes, gist_stream_t stream) {
    GIST_REQUIRE(ldg >= m && lda >= n && ldd >= n, "gist_gemm_tn_f32: leading dimension too small");
    return gist::launch_gemm<false, false>("gist_gemm_tn_f32", g, ldg, a, lda, nullptr, d, ldd, m,
                                           n, k, workspace, workspace_bytes,
                                           gist::as_stream(stream));
}
