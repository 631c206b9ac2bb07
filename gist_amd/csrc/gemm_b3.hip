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
// so there is ONE GEMM kernel, NT), rows padded with zeros to a multiple of 32 k, in the layout
//     row r : [k0..7 b1 (16 B)] [k0..7 b2] [k0..7 b3] [k8..15 b1] ...       (6 bytes per element)
// A lane's MFMA fragment (8 consecutive k of one row, one piece) is one 16-byte chunk.
//
// Main kernel: 256 x 128 x 32 block tile (A 48 KiB + B 24 KiB per stage, two stages = 144 KiB, one
// 512-thread workgroup per CU), 8 waves in 4 x 2, wave tile 64 x 64 = 4 x 4 MFMA tiles x 6 terms =
// 96 MFMAs per k tile, LDS-DMA staging one k tile ahead.  The tile is this large because at 6 bytes
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

__global__ __launch_bounds__(B3_THREADS, 2) void gemm_b3_kernel(B3Args g) {
    extern __shared__ __attribute__((aligned(16))) char b3_smem[];
    constexpr int NI = 4, NJ = 4;                 // 16-row slabs per wave: 64 x 64 wave tile
    const int nwg = g.tiles_m * g.tiles_n;
    const int orig = blockIdx.x;
    const int qd = nwg / kXcds, rm = nwg % kXcds, xcd = orig % kXcds;
    const int L = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + orig / kXcds;
    constexpr int GM = 8;
    const int width = GM * g.tiles_n;
    const int group = L / width;
    const int first_m = group * GM;
    const int gsz = min(g.tiles_m - first_m, GM);
    const int bm = first_m + (L % width) % gsz;
    const int bn = (L % width) / gsz;
    const int row0 = bm * B3_TM, col0 = bn * B3_TN;
    const int n_kt = g.kpad / B3_BK;

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
    const char *originA = reinterpret_cast<const char *>(g.a) + (int64_t)row0 * g.lda * 6;
    const char *originB = reinterpret_cast<const char *>(g.b) + (int64_t)col0 * g.ldb * 6;
    auto dma = [&](int buf, int kt) {
        char *sa = b3_smem + buf * B3_BUF_BYTES;
        b3_dma_image<DA>(originA + (int64_t)kt * B3_KT_BYTES, offA, sa, DA * wave);
        b3_dma_image<DB>(originB + (int64_t)kt * B3_KT_BYTES, offB, sa + B3_A_BYTES, DB * wave);
    };

    // fragment byte offsets inside an image: chunk c = 3 kg + piece of row rr of slab s at slot
    // 192 s + 12 rr + (c ^ (2 if rr >= 8))
    int fa[NI][3], fb[NJ][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int c = 3 * kg + p;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            fa[i][p] = ((wm * NI + i) * 192 + rr * 12 + (c ^ ((rr >> 3) << 1))) * 16;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            fb[j][p] = ((wn * NJ + j) * 192 + rr * 12 + (c ^ ((rr >> 3) << 1))) * 16 + B3_A_BYTES;
    }

    if (n_kt > 0) dma(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();

    auto kstep = [&](auto cur_c, auto next_c, int kt) {
        constexpr int cur = decltype(cur_c)::value;
        constexpr bool has_next = decltype(next_c)::value != 0;
        __builtin_amdgcn_sched_barrier(0);
        const char *img = b3_smem + cur * B3_BUF_BYTES;
        __builtin_amdgcn_s_setprio(1);
        // fragments in the order the terms need them (LDS returns in order): the first 16 MFMAs wait
        // for 8 reads, not for all 24
        bf16x8 a[NI][3], b[NJ][3];
#pragma unroll
        for (int i = 0; i < NI; ++i) a[i][0] = *reinterpret_cast<const bf16x8 *>(img + fa[i][0]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j][2] = *reinterpret_cast<const bf16x8 *>(img + fb[j][2]);
#pragma unroll
        for (int i = 0; i < NI; ++i) a[i][2] = *reinterpret_cast<const bf16x8 *>(img + fa[i][2]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j][0] = *reinterpret_cast<const bf16x8 *>(img + fb[j][0]);
#pragma unroll
        for (int i = 0; i < NI; ++i) a[i][1] = *reinterpret_cast<const bf16x8 *>(img + fa[i][1]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j][1] = *reinterpret_cast<const bf16x8 *>(img + fb[j][1]);
        // 6 terms x 16 output tiles, smallest terms first; the next tile's DMA goes out behind the
        // first 16 MFMAs (as in the f16x3 kernel: issued first it delays this wave's own MFMAs,
        // issued late it has not landed at the barrier)
        constexpr int pa[6] = {0, 2, 1, 0, 1, 0};
        constexpr int pb[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int t = 0; t < 6 * NI * NJ; ++t) {
            if (t == B3_DMA_AFTER) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (has_next) dma(cur ^ 1, kt + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            const int term = t / (NI * NJ), i = (t % (NI * NJ)) / NJ, j = t % NJ;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][pa[term]], b[j][pb[term]],
                                                                acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (has_next) __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
    };
    {
        using C0 = std::integral_constant<int, 0>;
        using C1 = std::integral_constant<int, 1>;
        int kt = 0;
        for (; kt + 2 < n_kt; kt += 2) {
            kstep(C0{}, C1{}, kt);
            kstep(C1{}, C1{}, kt + 1);
        }
        for (; kt < n_kt; ++kt) {
            const bool nx = kt + 1 < n_kt;
            if ((kt & 1) == 0) { if (nx) kstep(C0{}, C1{}, kt); else kstep(C0{}, C0{}, kt); }
            else               { if (nx) kstep(C1{}, C1{}, kt); else kstep(C1{}, C0{}, kt); }
        }
    }

    // ---- epilogue: C/D of 16x16x32: col = lane & 15, row = 4 (lane >> 4) + e ----
    float *cbase = g.c + (int64_t)row0 * g.ldc + col0;
    const int rows_valid = min(g.m - row0, B3_TM);
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

// ---- host side ------------------------------------------------------------------------------------
int64_t b3_kpad(int64_t k) { return ceil_div(k, B3_BK) * B3_BK; }

// Shapes the bf16x3 path takes: enough 128 x 64 tiles to occupy the chip, and enough flops to pay for
// the pre-pass (~ the f16x3 path's thresholds).  Everything else stays on the fp32 kernel.
static bool b3_shape_ok(int64_t m, int64_t n, int64_t k, double default_min_gflop) {
    if (h3_mode() != 2) return false;
    const double t_gflop = tune(GIST_TUNE_H3_MIN_GFLOP), t_tiles = tune(GIST_TUNE_H3_MIN_TILES);
    const double min_gflop = t_gflop > 0.0 ? t_gflop : default_min_gflop;
    const int min_tiles = t_tiles > 0.0 ? (int)t_tiles : 128;      // of 256 x 128
    // (an explicit tile threshold -- tests -- also lifts the minimum extents: the kernel itself
    // handles any m, n, k >= 1)
    if (t_tiles <= 0.0 && (m < 64 || n < 64 || k < 64)) return false;
    if (m < 1 || n < 1 || k < 1) return false;
    if (ceil_div(m, B3_TM) * ceil_div(n, B3_TN) < min_tiles) return false;
    if (2.0 * (double)m * (double)n * (double)k < min_gflop * 1e9) return false;
    if (b3_kpad(k) * 6 >= (1LL << 23)) return false;          // 32-bit DMA byte offsets: 256 rows * pitch
    return true;
}
bool b3_eligible(int64_t m, int64_t n, int64_t k) { return b3_shape_ok(m, n, k, 16.0); }
bool b3_eligible_kept(int64_t m, int64_t n, int64_t k) { return b3_shape_ok(m, n, k, 4.0); }

int64_t b3_workspace_bytes(int64_t m, int64_t n, int64_t k) {
    if (!b3_eligible(m, n, k)) return 0;
    return (m + n) * b3_kpad(k) * 6 + 512;
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
                     int64_t ldc, int64_t m, int64_t n, int64_t k, hipStream_t st) {
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
    B3Args g;
    const int64_t kpad = b3_kpad(k);
    g.a = sa; g.lda = kpad; g.b = sb; g.ldb = kpad; g.bias = bias; g.c = c; g.ldc = ldc;
    g.m = (int)m; g.n = (int)n; g.kpad = (int)kpad;
    g.tiles_m = (int)ceil_div(m, B3_TM);
    g.tiles_n = (int)ceil_div(n, B3_TN);
    const int64_t slot = timer_begin(tl_timer, 2, m, n, k, st);      // kind 2: the main kernel alone
    hipLaunchKernelGGL(gemm_b3_kernel, dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(B3_THREADS),
                       B3_STAGES * B3_BUF_BYTES, st, g);
    timer_end(tl_timer, slot, st);
    return launch_status(name);
}

// A: a_kc ? [m][k] : [k][m];  B: b_kc ? [n][k] : [k][n].  Returns 1 if the GEMM was issued,
// 0 if this call is not for the bf16x3 path (caller falls back to fp32), < 0 on error.
int b3_gemm(const char *name, bool a_kc, bool b_kc, const float *a, int64_t lda, const float *b,
            int64_t ldb, const float *bias, float *c, int64_t ldc, int64_t m, int64_t n, int64_t k,
            void *ws, int64_t ws_bytes, hipStream_t st) {
    if (!b3_eligible(m, n, k)) return 0;
    if (ws == nullptr || !aligned16(ws) || ws_bytes < b3_workspace_bytes(m, n, k)) return 0;
    const int64_t kpad = b3_kpad(k);
    uint16_t *sa = static_cast<uint16_t *>(ws);
    uint16_t *sb = sa + m * kpad * 3;
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
    rc = b3_gemm_presplit(name, sa, sb, bias, c, ldc, m, n, k, st);
    return rc == GIST_OK ? 1 : rc;
}

}  // namespace gist
