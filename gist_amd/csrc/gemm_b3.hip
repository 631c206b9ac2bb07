// fp32 GEMM on the bf16 matrix cores with ALL 24 operand bits: every fp32 operand x is split once
// into three bf16 pieces, x = b1 + b2 + b3 exactly (b1 = bf16(x), b2 = bf16(x - b1),
// b3 = bf16(x - b1 - b2): 3 x 8 significant bits, round-to-nearest at every step, bf16 has fp32's
// exponent range so there is no scale and no row maximum), and
//     (A.B^T)[i][j] = sum_k ( a1.b3 + a3.b1 + a2.b2 + a1.b2 + a2.b1 + a1.b1 )      (fp32 accumulation)
// on v_mfma_f32_16x16x32_bf16: the six cross terms down to 2^-16 of a product.  What is dropped
// (a2.b3, a3.b2, a3.b3) is below 2^-23 of |a||b| per product, the size of ONE fp32 rounding of that
// product; every bf16 x bf16 product is exact in fp32, and a k tile of 32 products is rounded into the
// accumulator 6 times instead of 16 times on v_mfma_f32_32x32x2_f32.  Measured against float64 the
// result is at the fp32-MFMA kernel's error level on every tested operand class, adversarial ones
// included (rms within 1.25x, max over 10^6 outputs within 3x: a few ulp of sum |a||b| either way;
// tests/test_gemm_b3_gpu.py).  Six bf16 MFMAs (16 cycles each) replace sixteen
// fp32-rate MFMA slots: 2.7x less matrix-core time per tile than the fp32 kernel.
//
// Pre-pass (HBM-bound, per operand): one kernel reads the fp32 source once and writes the operand
// k-contiguous whatever its source layout (the transposed form of NN/TN operands is produced here,
// so there is ONE GEMM kernel, NT), rows padded with zeros to a multiple of 64 k (an even number of k tiles), in the layout
//     row r : [k0..7 b1 (16 B)] [k0..7 b2] [k0..7 b3] [k8..15 b1] ...       (6 bytes per element)
// A lane's MFMA fragment (8 consecutive k of one row, one piece) is one 16-byte chunk.
//
// Main kernel: 256 x 128 x 32 block tile (A 48 KiB + B 24 KiB per stage, two stages = 144 KiB, one
// 512-thread workgroup per CU), 8 waves in 4 x 2, wave tile 64 x 64 = 4 x 4 MFMA tiles x 6 terms =
// 96 MFMAs per k tile, LDS-DMA staging issued a full k step ahead (below: one barrier per step, in
// its middle).  The tile is this large because at 6 bytes
// per element the L2 -> LDS traffic bounds smaller ones: 128 x 64 tiles with two workgroups per CU
// (9.4 GB staged for 2046 x 4096 x 8192) ran at 31 % matrix-pipe occupancy, this one (4.7 GB) at
// 45 %; a one-stage 128 x 128 variant with two workgroups per CU (fragment reads of one under the
// MFMAs of the other) was slower than either two-stage kernel (1.05 vs 0.95 ms per call).
// LDS image = [16-row slab][row][chunk] x 16 B, row-major like the operand: slot L of a slab
// (L = 12 row + position) holds chunk position ^ (2 if row >= 8), and a slab is three 1 KiB DMA
// instructions of 64 consecutive slots.  The four lanes of a DMA quad therefore read ONE aligned
// 64-byte piece of one row: the texture addresser handles a quad per cycle when it lies in one
// cache line, and the earlier image ([chunk][row]: a quad = 16 bytes from each of four rows, four
// lines) made the staging address-bound -- 1.08 -> 0.77 ms per 2046 x 4096 x 8192 call for the
// change of image alone.  The pair swap of rows 8-15 keeps the fragment reads conflict free: a
// ds_read_b128 lane group is rows {0-3, 12-15} of k group g with rows 4-11 of k group g + 1, i.e.
// chunks c and c + 3, and 16 B slot index mod 16 = (12 row + (c ^ swap)) mod 16 takes 16 different
// values over such a group (exhaustive check over the 3 pieces x 2 group kinds; no rotation by whole
// 64-byte pieces can do it, the slot index mod 4 would repeat).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

namespace gist {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float b3_f32x4 __attribute__((ext_vector_type(4)));

constexpr int B3_TM = 256, B3_TN = 128, B3_BK = 32;
constexpr int B3_THREADS = 512;
constexpr int B3_STAGES = 2;
#ifndef B3_SKEW
#define B3_SKEW 16               // MFMAs by which waves 4-7 run behind waves 0-3 inside a k step
#endif
#ifndef B3_DMA_AFTER
#define B3_DMA_AFTER 16        // MFMAs of a k step issued before the next tile's DMA
#endif
constexpr int B3_KT_BYTES = B3_BK * 6;                 // 192 B of one row per k tile (12 chunks)
constexpr int B3_A_BYTES = B3_TM * B3_KT_BYTES;        // 24 KiB
constexpr int B3_B_BYTES = B3_TN * B3_KT_BYTES;        // 12 KiB
constexpr int B3_BUF_BYTES = B3_A_BYTES + B3_B_BYTES;

// ---- pre-pass ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t b3_pack(__bf16 lo, __bf16 hi) {
    return (uint32_t)__builtin_bit_cast(unsigned short, lo) |
           ((uint32_t)__builtin_bit_cast(unsigned short, hi) << 16);
}

// 8 consecutive k of one row -> 3 x 16 bytes at dst (bf16 pieces 1, 2, 3)
__device__ __forceinline__ void b3_emit(const float (&v)[8], uint16_t *__restrict__ dst) {
    uint32_t w[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __bf16 p[2][3];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float x = v[2 * j + h];
            p[h][0] = (__bf16)x;
            const float r1 = x - (float)p[h][0];            // exact
            p[h][1] = (__bf16)r1;
            const float r2 = r1 - (float)p[h][1];           // exact
            p[h][2] = (__bf16)r2;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) w[q][j] = b3_pack(p[0][q], p[1][q]);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
        *reinterpret_cast<uint4 *>(dst + 8 * q) = make_uint4(w[q][0], w[q][1], w[q][2], w[q][3]);
}

// One read of src[rows, cols] -> the split operand with k = columns (dst_r: [rows][kpad(cols)]) and/or
// the one with k = rows (dst_t: [cols][kpad(rows)]), 64 x 64 per block through LDS; pitches in k
// elements (6 bytes each).  Dropout is applied on the fly with gist_dropout_f32's generator (element
// index offset + r * cols + c), so the fp32 dropped tensor never has to exist.
__global__ __launch_bounds__(256) void b3_dual_split_kernel(B3Dual d, int64_t ldd_r, int64_t ldd_t,
                                                            float keep) {
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int t = threadIdx.x;
    const int c4 = (t & 15) * 4;
    const uint64_t sm = d.seed * 0x9E3779B97F4A7C15ULL;
    const float inv24 = 1.0f / 16777216.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = (t >> 4) + 16 * i;
        const int64_t r = r0 + rr;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < d.rows) {
            const float *p = d.src + r * d.ld + c0 + c4;
            if (c0 + c4 + 3 < d.cols && d.vec4) {
                const float4 q = *reinterpret_cast<const float4 *>(p);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c0 + c4 + j < d.cols) v[j] = p[j];
            }
            if (d.p > 0.f) {
                const uint64_t idx0 = d.offset + (uint64_t)r * (uint64_t)d.cols + (uint64_t)(c0 + c4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint64_t idx = idx0 + j;
                    const uint64_t h = splitmix64((idx >> 1) + sm);
                    const uint32_t w = (idx & 1) ? (uint32_t)(h >> 32) : (uint32_t)h;
                    v[j] *= ((float)(w >> 8) * inv24 >= d.p) ? keep : 0.f;
                }
            }
        }
        tile[rr][c4 + 0] = v[0]; tile[rr][c4 + 1] = v[1];
        tile[rr][c4 + 2] = v[2]; tile[rr][c4 + 3] = v[3];
    }
    __syncthreads();
    // column sums of this 64-row chunk (rows past the end are zeros in the tile), fixed order
    if (d.col_partials != nullptr && t < 64 && c0 + t < d.cols && r0 < d.rows) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int r = 0; r < 64; r += 4) {
            s0 += tile[r][t]; s1 += tile[r + 1][t]; s2 += tile[r + 2][t]; s3 += tile[r + 3][t];
        }
        d.col_partials[(int64_t)blockIdx.y * d.cols + c0 + t] = (s0 + s1) + (s2 + s3);
    }
    const int kb = t & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = (t >> 3) + 32 * i;
        if (d.dst_r != nullptr && r0 + q < d.rows && c0 + kb * 8 < ldd_r) {      // row q, 8 columns
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[q][kb * 8 + j];
            b3_emit(v, d.dst_r + ((int64_t)(r0 + q) * ldd_r + c0 + kb * 8) * 3);
        }
        if (d.dst_t != nullptr && c0 + q < d.cols && r0 + kb * 8 < ldd_t) {      // column q, 8 rows
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[kb * 8 + j][q];
            b3_emit(v, d.dst_t + ((int64_t)(c0 + q) * ldd_t + r0 + kb * 8) * 3);
        }
    }
}

// ---- the GEMM -----------------------------------------------------------------------------------
struct B3Args {
    const uint16_t *a; int64_t lda;      // split operands, [rows][kpad] elements of 6 bytes
    const uint16_t *b; int64_t ldb;
    const float *bias;
    float *c; int64_t ldc;
    int m, n, kpad;
    int tiles_m, tiles_n;
    int kt_per_split;                    // k tiles per blockIdx.y (even); split s writes c + s * split_stride
    int64_t split_stride;
    // Tail units (one k slice, tiles > one per CU): the first dp_tiles logical tiles run whole, one workgroup
    // each; every remaining tile is cut into tail_splits k slices of tail_kt k tiles (even), one workgroup
    // each (blockIdx.x = dp_tiles + tile * tail_splits + slice), which leave their accumulators at
    // tail_partials[unit][16 registers][512 threads] x 16 B; gemm_b3_tail_sum_kernel, the next launch,
    // sums a tile's partials in slice order and stores C (+ bias).
    int dp_tiles, tail_splits, tail_kt;
    float *tail_partials;
};

// DMA instruction `inst` of an image: slot L = 64 (inst % 3) + lane of 16-row slab inst / 3; slot
// L holds row L / 12, chunk (L % 12) ^ (2 if row >= 8): the four lanes of a quad read one aligned
// 64-byte piece of one row
template <int NI>
__device__ __forceinline__ void b3_dma_offsets(int64_t ld, int rows, int row0, int first, int lane,
                                                uint32_t (&off)[NI]) {
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) {
        const int inst = first + jj;
        const int slot = (inst % 3) * 64 + lane;
        const int rs = slot / 12;
        const int chunk = (slot % 12) ^ ((rs >> 3) << 1);
        const int r = (inst / 3) * 16 + rs;
        const int dr = min(r, rows - 1 - row0);
        off[jj] = (uint32_t)((int64_t)dr * ld * 6 + chunk * 16);
    }
}

// acc += a . b in place.  (Written as asm with a tied accumulator: left to itself the register
// allocator gives many of the 96 MFMAs of a k step a destination different from their addend,
// rotates the 64 accumulator registers through a pool it does not have, and spills accumulators to
// scratch inside the k loop -- 113 VGPRs with the fragments live across the loop edge.)
__device__ __forceinline__ void b3_mfma(b3_f32x4 &acc, const bf16x8 &a, const bf16x8 &b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

template <int NI>
__device__ __forceinline__ void b3_dma_image(const char *base, const uint32_t (&off)[NI], char *image,
                                             int first) {
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(base), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int jj = 0; jj < NI; ++jj)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrc, (__attribute__((address_space(3))) void *)(image + (first + jj) * 1024), 16, off[jj],
            0, 0, 0);
}

#ifdef B3_CLOCK_PROBE   // dev build (scripts/b3_clock_probe.py): core cycles and 100 MHz ticks of every workgroup's k loop
__device__ unsigned long long g_b3_clock[2 * 4096];
#endif

__global__ __launch_bounds__(B3_THREADS, 2) void gemm_b3_kernel(B3Args g) {
    extern __shared__ __attribute__((aligned(16))) char b3_smem[];
    constexpr int NI = 4, NJ = 4;                 // 16-row slabs per wave: 64 x 64 wave tile
    const int nwg = g.dp_tiles;
    const int orig = blockIdx.x;
    const bool tail = orig >= nwg;       // a k slice of one of the tiles past the last full round
    const int tail_tile = tail ? (orig - nwg) / g.tail_splits : 0;
    const int tail_slice = tail ? (orig - nwg) % g.tail_splits : 0;
    const int qd = nwg / kXcds, rm = nwg % kXcds, xcd = orig % kXcds;
    const int L = tail ? nwg + tail_tile
                       : (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + orig / kXcds;
    constexpr int GM = 8;
    const int width = GM * g.tiles_n;
    const int group = L / width;
    const int first_m = group * GM;
    const int gsz = min(g.tiles_m - first_m, GM);
    const int bm = first_m + (L % width) % gsz;
    const int bn = (L % width) / gsz;
    const int row0 = bm * B3_TM, col0 = bn * B3_TN;
    // split-K: this workgroup's k tiles
    const int kt0 = tail ? tail_slice * g.tail_kt : (int)blockIdx.y * g.kt_per_split;
    const int n_kt = min(g.kpad / B3_BK - kt0, tail ? g.tail_kt : g.kt_per_split);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int rr = lane & 15, kg = lane >> 4;

    b3_f32x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    // DMA instructions of 1 KiB: 3 per 16 rows of an image, dealt evenly to the waves
    constexpr int NWAVES = B3_THREADS / 64;
    constexpr int DA = B3_TM / 16 * 3 / NWAVES, DB = B3_TN / 16 * 3 / NWAVES;
    uint32_t offA[DA], offB[DB];
    b3_dma_offsets<DA>(g.lda, g.m, row0, DA * wave, lane, offA);
    b3_dma_offsets<DB>(g.ldb, g.n, col0, DB * wave, lane, offB);
    const char *originA = reinterpret_cast<const char *>(g.a) + (int64_t)row0 * g.lda * 6 + (int64_t)kt0 * B3_KT_BYTES;
    const char *originB = reinterpret_cast<const char *>(g.b) + (int64_t)col0 * g.ldb * 6 + (int64_t)kt0 * B3_KT_BYTES;
    auto dma = [&](int buf, int kt) {
        char *sa = b3_smem + buf * B3_BUF_BYTES;
        b3_dma_image<DA>(originA + (int64_t)kt * B3_KT_BYTES, offA, sa, DA * wave);
        b3_dma_image<DB>(originB + (int64_t)kt * B3_KT_BYTES, offB, sa + B3_A_BYTES, DB * wave);
    };

    // fragment addresses: chunk c = 3 kg + piece of row rr of slab s sits at slot
    // 192 s + 12 rr + (c ^ (2 if rr >= 8)) of its image.  One lane-dependent LDS address per (stage,
    // image, piece) = 12 VGPRs; the slab (3 KiB apart) goes into the ds_read offset field.
    const char *fa[B3_STAGES][3], *fb[B3_STAGES][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int c = 3 * kg + p;
        const int in_slab = (rr * 12 + (c ^ ((rr >> 3) << 1))) * 16;
#pragma unroll
        for (int st = 0; st < B3_STAGES; ++st) {
            fa[st][p] = b3_smem + st * B3_BUF_BYTES + wm * NI * 3072 + in_slab;
            fb[st][p] = b3_smem + st * B3_BUF_BYTES + B3_A_BYTES + wn * NJ * 3072 + in_slab;
        }
    }

    // One barrier per k step, in the MIDDLE of the step.  Terms in the order
    //   a2.b1  a3.b1 | a2.b2  a1.b1 | a1.b2  a1.b3        (16 MFMAs each; pieces 1-based as above)
    // so that after the first four (64 MFMAs) the registers of a2, a3 and b1 are dead, and by then
    // every fragment read of the current image has been consumed.  At that point: wait for the own
    // pieces of the next tile's DMA (issued a whole step earlier), barrier -- next image complete AND
    // current image free --, issue the DMA of the tile after next into the current image, read the
    // a2 / a3 / b1 fragments of the next tile, and only then issue the last 32 MFMAs, which need
    // none of these.  The matrix pipe never waits on a post-barrier LDS round trip (with the barrier
    // at the end of the step all eight waves start the next one with 8 KiB of fragment reads each and
    // nothing to issue), and the DMA has a full step to land instead of 80 MFMA slots.
    // In-kernel stamps (scripts/b3_clock_probe.py, 2046 x 4096 x 8192 / 4096 x 8192 x 2046): MFMA
    // cycles are 0.79 / 0.84 of the k loop's cycles at 1.95 / 1.93 GHz, against 0.71 / 0.80 at
    // 2.08 / 1.96 GHz with the barrier at the end of the step -- the chip gives back in clock most of
    // what the pipe share gains (power), the loop itself is 2-4 % shorter (508 vs 531 us).
    // A wave whose 64 rows all lie past the end of C (the last row tile of a batch a few rows over a multiple of 256)
    // stages its share of the images and meets the barriers, nothing else: its SIMD's other wave has the pipe alone.
    const bool live = __builtin_amdgcn_readfirstlane(row0 + wm * (16 * NI) < g.m);
    bf16x8 a[NI][3], b[NJ][3];
    auto read_head = [&](auto st_c) {
        constexpr int st = decltype(st_c)::value;
#ifndef B3_NO_DEAD_SKIP
        if (!live) return;
#endif
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j][0] = *reinterpret_cast<const bf16x8 *>(fb[st][0] + j * 3072);
#pragma unroll
        for (int i = 0; i < NI; ++i) a[i][1] = *reinterpret_cast<const bf16x8 *>(fa[st][1] + i * 3072);
#pragma unroll
        for (int i = 0; i < NI; ++i) a[i][2] = *reinterpret_cast<const bf16x8 *>(fa[st][2] + i * 3072);
    };
    auto read_rest = [&](auto st_c) {
        constexpr int st = decltype(st_c)::value;
#ifndef B3_NO_DEAD_SKIP
        if (!live) return;
#endif
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j][1] = *reinterpret_cast<const bf16x8 *>(fb[st][1] + j * 3072);
#pragma unroll
        for (int i = 0; i < NI; ++i) a[i][0] = *reinterpret_cast<const bf16x8 *>(fa[st][0] + i * 3072);
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j][2] = *reinterpret_cast<const bf16x8 *>(fb[st][2] + j * 3072);
    };
    constexpr int pa[6] = {1, 2, 1, 0, 0, 0};
    constexpr int pb[6] = {0, 0, 1, 0, 1, 2};
    constexpr int B3_MID = 4 * NI * NJ;           // MFMAs before the barrier

    dma(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    dma(1, 1);                                       // n_kt is even and >= 2
    read_head(std::integral_constant<int, 0>{});

    // The two waves of a SIMD (w and w + 4) run the step half a term apart, so that one of them has
    // MFMAs to issue while the other issues its fragment reads and DMA instructions: waves 4-7
    // (`late`) issue 16 MFMAs before each of the two non-MFMA blocks of the step.
    auto mfmas = [&](auto t0_c, auto t1_c) {
        constexpr int t0 = decltype(t0_c)::value, t1 = decltype(t1_c)::value;
#ifndef B3_NO_DEAD_SKIP
        if (!live) return;
#endif
#pragma unroll
        for (int t = t0; t < t1; ++t) {
            const int term = t / (NI * NJ), i = (t % (NI * NJ)) / NJ, j = t % NJ;
            b3_mfma(acc[i][j], a[i][pa[term]], b[j][pb[term]]);
        }
    };
    // Every step is the same straight-line code: k is padded to an EVEN number of tiles (zeros), the
    // DMA of a tile past the end re-reads the last one into an image nobody consumes, and the last
    // step's barrier and fragment prefetch are simply wasted.  (Variants per remaining-tile count
    // made the compiler merge accumulator registers across branches with hundreds of v_mov.)
    const int last_kt = n_kt - 1;
    auto kstep = [&](auto cur_c, auto late_c, int kt) {
        constexpr int cur = decltype(cur_c)::value;
        constexpr int skew = decltype(late_c)::value * B3_SKEW;
        using I = std::integral_constant<int, 0>;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        mfmas(I{}, std::integral_constant<int, skew>{});
        __builtin_amdgcn_sched_barrier(0);
        read_rest(cur_c);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(std::integral_constant<int, skew>{}, std::integral_constant<int, B3_MID>{});
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0)
        __syncthreads();
        mfmas(std::integral_constant<int, B3_MID>{}, std::integral_constant<int, B3_MID + skew>{});
        __builtin_amdgcn_sched_barrier(0);
        dma(cur, min(kt + 2, last_kt));
        read_head(std::integral_constant<int, cur ^ 1>{});
        __builtin_amdgcn_sched_barrier(0);
        mfmas(std::integral_constant<int, B3_MID + skew>{}, std::integral_constant<int, 6 * NI * NJ>{});
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto run = [&](auto late_c) {
        using C0 = std::integral_constant<int, 0>;
        using C1 = std::integral_constant<int, 1>;
        for (int kt = 0; kt < n_kt; kt += 2) {
            kstep(C0{}, late_c, kt);
            kstep(C1{}, late_c, kt + 1);
        }
    };
#ifdef B3_CLOCK_PROBE
    const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifndef B3_NO_SKEW
    if (wave >= 4) run(std::integral_constant<int, 1>{});
    else
#endif
        run(std::integral_constant<int, 0>{});
#ifdef B3_CLOCK_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        g_b3_clock[2 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime() - ck0;
        g_b3_clock[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif
    __builtin_amdgcn_s_waitcnt(0x0070);              // the wasted DMA and reads of the last step

    // the MFMAs above are opaque to the compiler's hazard recognizer: let the last ones retire
    // before the accumulators are read (4 passes + write-back)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const int rows_valid = min(g.m - row0, B3_TM);
    if (tail) {
        // A wave whose 64 rows lie past the end of C has nothing to exchange (the usual tail tile is the
        // few rows a batch has beyond a multiple of 256).
        const bool rows_live = wm * (16 * NI) < rows_valid;
        b3_f32x4 *mine = reinterpret_cast<b3_f32x4 *>(g.tail_partials) +
                         (int64_t)(tail_tile * g.tail_splits + tail_slice) * (NI * NJ * B3_THREADS) + threadIdx.x;
        if (rows_live) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) mine[(i * NJ + j) * B3_THREADS] = acc[i][j];
        }
        return;                                      // gemm_b3_tail_sum_kernel, the next launch, finishes these tiles
    }
    // ---- epilogue: C/D of 16x16x32: col = lane & 15, row = 4 (lane >> 4) + e ----
    float *cbase = g.c + (int64_t)blockIdx.y * g.split_stride + (int64_t)row0 * g.ldc + col0;
    __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
        cbase, 0, (int)((int64_t)rows_valid * g.ldc * 4), 0x00020000);
    const uint32_t ldc_b = (uint32_t)g.ldc * 4;
    uint32_t cvoff[NJ];
    float bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int cl = wn * (16 * NJ) + j * 16 + rr;
        const bool ok = col0 + cl < g.n;
        cvoff[j] = ok ? (uint32_t)(wm * (16 * NI) + 4 * kg) * ldc_b + (uint32_t)cl * 4 : 0x7fffffffu;
        bv[j] = (g.bias != nullptr && ok) ? g.bias[col0 + cl] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t roff = (uint32_t)(i * 16 + e) * ldc_b;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float v = acc[i][j][e] + bv[j];
                if (rows_valid == B3_TM)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), crsrc, cvoff[j], roff, 0);
                else
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), crsrc, cvoff[j] + roff, 0, 0);
            }
        }
}

// The tiles the tail units computed: C tile = sum over its k slices of the partials, in slice order (+ bias).
// Grid (tile, accumulator register 0..15); a thread holds the element(s) its twin in gemm_b3_kernel held.
__global__ __launch_bounds__(B3_THREADS) void gemm_b3_tail_sum_kernel(B3Args g) {
    constexpr int NI = 4, NJ = 4, GM = 8;
    const int L = g.dp_tiles + (int)blockIdx.x;
    const int width = GM * g.tiles_n;
    const int first_m = (L / width) * GM;
    const int gsz = min(g.tiles_m - first_m, GM);
    const int row0 = (first_m + (L % width) % gsz) * B3_TM, col0 = ((L % width) / gsz) * B3_TN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, rr = lane & 15, kg = lane >> 4;
    const int reg = blockIdx.y, i = reg / NJ, j = reg % NJ;
    const int row = row0 + wm * (16 * NI) + i * 16 + 4 * kg, col = col0 + wn * (16 * NJ) + j * 16 + rr;
    if (row0 + wm * (16 * NI) >= g.m || col >= g.n) return;      // (the first test is the one the producer made)
    const b3_f32x4 *part = reinterpret_cast<const b3_f32x4 *>(g.tail_partials) +
                           ((int64_t)blockIdx.x * g.tail_splits * (NI * NJ) + reg) * B3_THREADS + threadIdx.x;
    b3_f32x4 sum = part[0];
    for (int q = 1; q < g.tail_splits; ++q) sum += part[(int64_t)q * (NI * NJ) * B3_THREADS];
    const float bv = g.bias != nullptr ? g.bias[col] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (row + e < g.m) g.c[(int64_t)(row + e) * g.ldc + col] = sum[e] + bv;
}

// ---- host side ------------------------------------------------------------------------------------
int64_t b3_kpad(int64_t k) { return ceil_div(k, 2 * B3_BK) * (2 * B3_BK); }      // an even number of k tiles

// Split-K: an output with fewer than ~3/4 of 256 tiles leaves CUs idle (one 512-thread workgroup per
// CU), so its k range is cut into `splits` slices, one workgroup each (blockIdx.y), which write fp32
// slabs that one pass sums (+ bias).  A slice keeps >= 8 k tiles; slices are an even number of tiles.
int b3_splits(int64_t m, int64_t n, int64_t k) {
    const int64_t tiles = ceil_div(m, B3_TM) * ceil_div(n, B3_TN);
    const int64_t n_kt = b3_kpad(k) / B3_BK;
    const int forced = (int)tune(GIST_TUNE_GEMM_SPLITS);
    int64_t s = forced > 0 ? forced : (tiles >= 192 ? 1 : 256 / tiles);
    if (forced <= 0 && tiles > 128 && tiles < 192) {
        // between half a chip and 3/4 of one (dW_0 of the H = 4096 step: 160 tiles) one slice runs a single
        // under-full round; the slice count that minimises rounds x k tiles per slice wins even with the
        // slab sum (4096 x 1204 x 2046: 3 slices = 2 rounds of 22 k tiles against 1 of 64; 165 -> 154 us
        // per call with the sum in the call, scripts/b3_split_probe.py)
        int64_t best = n_kt + 4;      // one slice: no slabs (the +4: a slab sum costs about 4 k tiles)
        for (int64_t c = 2; c <= 4; ++c) {
            const int64_t per = ceil_div(ceil_div(n_kt, c), 2) * 2;
            const int64_t cost = ceil_div(tiles * c, 256) * per + 8;
            if (per >= 8 && cost < best) { best = cost; s = c; }
        }
    }
    if (forced <= 0 && s > n_kt / 8) s = n_kt / 8;
    if (s > n_kt / 2) s = n_kt / 2;
    if (s < 1) s = 1;
    const int64_t per = ceil_div(ceil_div(n_kt, s), 2) * 2;
    return (int)ceil_div(n_kt, per);
}
// Tail units: with one k slice and T tiles on P = 256 one-workgroup CUs the last round holds r = T mod P
// tiles and leaves P - r CUs idle for a whole tile's k loop (a batch of 2049-2304 rows has a ninth row
// tile: 288 tiles = one round + 32, twice the time of 256).  The r tiles of that round are cut into
// floor(P / r) k slices of >= 8 k tiles, one workgroup each, so the round lasts 1 / slices of a tile;
// the slices of a tile leave fp32 partials that gemm_b3_tail_sum_kernel, the next launch, adds in slice
// order (a hand-over inside the kernel -- the last slice to arrive sums -- was measured first: its
// device-scope fences cost 45-70 us per launch).  Nothing here depends on anything but the shape.
struct B3Tail { int dp_tiles, splits, kt; };
constexpr int B3_CUS = 256;
static B3Tail b3_tail(int64_t m, int64_t n, int64_t k) {
    const int64_t tiles = ceil_div(m, B3_TM) * ceil_div(n, B3_TN);
    B3Tail t{(int)tiles, 1, 0};
    if (tune(GIST_TUNE_B3_TAIL) == 1.0 || tiles <= B3_CUS) return t;
    const int64_t r = tiles % B3_CUS;
    if (r == 0 || r > B3_CUS / 2) return t;
    const int64_t n_kt = b3_kpad(k) / B3_BK;
    int64_t s = B3_CUS / r;
    if (s > n_kt / 8) s = n_kt / 8;
    if (s < 2) return t;
    const int64_t per = ceil_div(ceil_div(n_kt, s), 2) * 2;
    s = ceil_div(n_kt, per);
    if (s < 2) return t;
    t.dp_tiles = (int)(tiles - r); t.splits = (int)s; t.kt = (int)per;
    return t;
}
static int64_t b3_tail_bytes(int64_t m, int64_t n, int64_t k) {
    const B3Tail t = b3_tail(m, n, k);
    if (t.splits < 2) return 0;
    const int64_t r = ceil_div(m, B3_TM) * ceil_div(n, B3_TN) - t.dp_tiles;
    return r * t.splits * (int64_t)(B3_TM * B3_TN * 4);
}
// scratch of one call: fp32 slabs of a split-K call, or the partials of its tail units (never both)
int64_t b3_slab_bytes(int64_t m, int64_t n, int64_t k) {
    const int s = b3_splits(m, n, k);
    return s > 1 ? (int64_t)s * m * n * 4 : b3_tail_bytes(m, n, k);
}

// Shapes the bf16x3 path takes: enough workgroups (256 x 128 tiles x k slices) to occupy the chip, and
// enough flops to pay for the pre-pass (~ the f16x3 path's thresholds).  Everything else stays on the
// fp32 kernel.
static bool b3_shape_ok(int64_t m, int64_t n, int64_t k, double default_min_gflop) {
    if (h3_mode() != 2) return false;
    const double t_gflop = tune(GIST_TUNE_H3_MIN_GFLOP), t_tiles = tune(GIST_TUNE_H3_MIN_TILES);
    const double min_gflop = t_gflop > 0.0 ? t_gflop : default_min_gflop;
    const int min_wgs = t_tiles > 0.0 ? (int)t_tiles : 128;
    // (an explicit tile threshold -- tests -- also lifts the minimum extents: the kernel itself
    // handles any m, n, k >= 1)
    if (t_tiles <= 0.0 && (m < 64 || n < 64 || k < 64)) return false;
    if (m < 1 || n < 1 || k < 1) return false;
    const int64_t tiles = ceil_div(m, B3_TM) * ceil_div(n, B3_TN);
    if (tiles * b3_splits(m, n, k) < min_wgs) return false;
    if (t_tiles <= 0.0 && (double)m * (double)n < 0.6 * (double)(tiles * B3_TM * B3_TN)) return false;   // mostly padding
    if (2.0 * (double)m * (double)n * (double)k < min_gflop * 1e9) return false;
    if (b3_kpad(k) * 6 >= (1LL << 23)) return false;          // 32-bit DMA byte offsets: 256 rows * pitch
    return true;
}
bool b3_eligible(int64_t m, int64_t n, int64_t k) { return b3_shape_ok(m, n, k, 16.0); }
// (inside the step, operands split once per tensor: measured break-even between 8.6 GFLOP -- the h = 1024
// projections, 0.595 vs 0.587 ms/step on the fp32 kernel -- and 10.1 GFLOP -- the layer-0 projections at
// h = 2048, 1.071 vs 1.095)
bool b3_eligible_kept(int64_t m, int64_t n, int64_t k) { return b3_shape_ok(m, n, k, 9.0); }

static int64_t b3_operand_bytes(int64_t m, int64_t n, int64_t k) {
    return ceil_div((m + n) * b3_kpad(k) * 6 + 512, 256) * 256;
}
int64_t b3_workspace_bytes(int64_t m, int64_t n, int64_t k) {
    if (!b3_eligible(m, n, k)) return 0;
    return b3_operand_bytes(m, n, k) + b3_slab_bytes(m, n, k);
}

int b3_dual_split(const B3Dual &d, hipStream_t st) {
    if (d.rows <= 0 || d.cols <= 0) return GIST_OK;
    const int64_t ldd_r = b3_kpad(d.cols), ldd_t = b3_kpad(d.rows);
    const int64_t gx = ceil_div(d.dst_r ? ldd_r : d.cols, 64), gy = ceil_div(d.dst_t ? ldd_t : d.rows, 64);
    B3Dual dd = d;
    dd.vec4 = aligned16(d.src) && d.ld % 4 == 0;
    hipLaunchKernelGGL(b3_dual_split_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, st, dd, ldd_r,
                       ldd_t, d.p > 0.f ? 1.0f / (1.0f - d.p) : 1.0f);
    return launch_status("b3_dual_split");
}

int b3_gemm_presplit(const char *name, const uint16_t *sa, const uint16_t *sb, const float *bias, float *c,
                     int64_t ldc, int64_t m, int64_t n, int64_t k, float *slabs, int64_t slab_bytes,
                     hipStream_t st, int *deferred) {
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_b3_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, B3_STAGES * B3_BUF_BYTES);
        if (e != hipSuccess) {
            set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
            return GIST_ELAUNCH;
        }
        once.done(dev);
    }
    if (ldc * B3_TM * 4 >= (1LL << 31)) {      // the store epilogue's 32-bit byte offsets: 256 rows x ldc
        set_error("%s: leading dimension of the output >= 2^21 elements on the bf16x3 path", name);
        return GIST_EINVAL;
    }
    B3Args g;
    const int64_t kpad = b3_kpad(k);
    g.a = sa; g.lda = kpad; g.b = sb; g.ldb = kpad; g.bias = bias; g.c = c; g.ldc = ldc;
    g.m = (int)m; g.n = (int)n; g.kpad = (int)kpad;
    g.tiles_m = (int)ceil_div(m, B3_TM);
    g.tiles_n = (int)ceil_div(n, B3_TN);
    const int64_t n_kt = kpad / B3_BK;
    int splits = b3_splits(m, n, k);
    if (splits > 1 && (slabs == nullptr || slab_bytes < (int64_t)splits * m * n * 4)) splits = 1;
    g.kt_per_split = (int)(ceil_div(ceil_div(n_kt, splits), 2) * 2);
    splits = (int)ceil_div(n_kt, g.kt_per_split);
    g.split_stride = 0;
    if (splits > 1) { g.c = slabs; g.ldc = n; g.split_stride = m * n; g.bias = nullptr; }
    g.dp_tiles = g.tiles_m * g.tiles_n; g.tail_splits = 1; g.tail_kt = 0;
    g.tail_partials = nullptr;
    unsigned grid_x = (unsigned)g.dp_tiles;
    if (splits == 1 && slabs != nullptr && aligned16(slabs)) {
        const B3Tail t = b3_tail(m, n, k);
        if (t.splits > 1 && slab_bytes >= b3_tail_bytes(m, n, k)) {
            const int r = g.dp_tiles - t.dp_tiles;
            g.dp_tiles = t.dp_tiles; g.tail_splits = t.splits; g.tail_kt = t.kt;
            g.tail_partials = slabs;
            grid_x = (unsigned)(t.dp_tiles + r * t.splits);
        }
    }
    const int64_t slot = timer_begin(tl_timer, 2, m, n, k, st);      // kind 2: the main kernel (+ slab sum)
    hipLaunchKernelGGL(gemm_b3_kernel, dim3(grid_x, (unsigned)splits),
                       dim3(B3_THREADS), B3_STAGES * B3_BUF_BYTES, st, g);
    int rc = launch_status(name);
    if (rc == GIST_OK && g.tail_splits > 1) {
        hipLaunchKernelGGL(gemm_b3_tail_sum_kernel, dim3((grid_x - (unsigned)g.dp_tiles) / (unsigned)g.tail_splits, 16),
                           dim3(B3_THREADS), 0, st, g);
        rc = launch_status(name);
    }
    // deferred: the caller's consumer sums the slabs (slab s at slabs + s m n, in slab order); bias must be null
    if (deferred) *deferred = splits;
    else if (rc == GIST_OK && splits > 1) rc = splitk_reduce(name, slabs, m * n, splits, bias, c, ldc, m, n, st);
    timer_end(tl_timer, slot, st);
    return rc;
}

// A: a_kc ? [m][k] : [k][m];  B: b_kc ? [n][k] : [k][n].  Returns 1 if the GEMM was issued,
// 0 if this call is not for the bf16x3 path (caller falls back to fp32), < 0 on error.
int b3_gemm(const char *name, bool a_kc, bool b_kc, const float *a, int64_t lda, const float *b,
            int64_t ldb, const float *bias, float *c, int64_t ldc, int64_t m, int64_t n, int64_t k,
            void *ws, int64_t ws_bytes, hipStream_t st) {
    if (!b3_eligible(m, n, k)) return 0;
    if (ldc * B3_TM * 4 >= (1LL << 31)) return 0;      // the store epilogue's 32-bit byte offsets: 256 rows x ldc
    if (ws == nullptr || !aligned16(ws) || ws_bytes < b3_workspace_bytes(m, n, k)) return 0;
    const int64_t kpad = b3_kpad(k);
    uint16_t *sa = static_cast<uint16_t *>(ws);
    uint16_t *sb = sa + m * kpad * 3;
    float *slabs = reinterpret_cast<float *>(static_cast<char *>(ws) + b3_operand_bytes(m, n, k));
    auto split = [&](bool kc, const float *src, int64_t ld, int64_t rows, uint16_t *dst) {
        B3Dual d{};
        d.src = src; d.ld = ld;
        if (kc) { d.rows = rows; d.cols = k; d.dst_r = dst; }      // [rows][k]
        else { d.rows = k; d.cols = rows; d.dst_t = dst; }         // [k][rows] -> transposed
        return b3_dual_split(d, st);
    };
    int rc = split(a_kc, a, lda, m, sa);
    if (rc != GIST_OK) return rc;
    rc = split(b_kc, b, ldb, n, sb);
    if (rc != GIST_OK) return rc;
    rc = b3_gemm_presplit(name, sa, sb, bias, c, ldc, m, n, k, slabs, b3_slab_bytes(m, n, k), st, nullptr);
    return rc == GIST_OK ? 1 : rc;
}

}  // namespace gist

#ifdef B3_CLOCK_PROBE
extern "C" int gist_b3_clock_read(unsigned long long *out, int64_t n_blocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(gist::g_b3_clock), n_blocks * 2 * sizeof(unsigned long long)) ==
                   hipSuccess ? 0 : -1;
}
#endif
