// fp32-accurate GEMM on the f16 matrix cores: every fp32 operand x is split once into two
// halves, x * 2^e = hi + lo with hi = f16(x 2^e), lo = f16(x 2^e - hi) (22 significant bits, one
// power-of-two scale per operand ROW so neither half leaves the f16 range), and
//     (A.B^T)[i][j]  ~=  2^-(ea_i+eb_j) * sum_k (ah.bh + ah.bl + al.bh)   (fp32 accumulation)
// on v_mfma_f32_16x16x32_f16.  The dropped al.bl term is 2^-22 of a product, below the fp32
// rounding of the running sum; measured against fp64 the result has the error of the
// fp32-MFMA kernel or less (tests/test_gemm_h3_gpu.py, profiles/r01_gemm_h3_error_vs_float64.txt).
// Three f16 MFMAs replace sixteen fp32-rate MFMA slots: the matrix-core time of a tile drops
// 5.3x; what bounds the kernel then is the clock the chip holds under the MFMA load (1.8 GHz
// measured) and, at 64 % matrix-pipe occupancy in cycles, the L2 -> LDS staging next to it.
//
// Pre-pass (HBM-bound, per operand): row maxima -> scale exponents; split kernel writes the operand
// k-contiguous whatever its source layout (the transposed form of NN/TN operands is produced
// here, so there is ONE GEMM kernel, NT), rows padded with zeros to a multiple of 32 k, in the
// "chunk-interleaved" layout
//     row r : [k0..7 hi (16 B)] [k0..7 lo (16 B)] [k8..15 hi] [k8..15 lo] ...
// i.e. 4 bytes per element like fp32: a 128 x 32-k tile is the same 16 KiB LDS image the
// fp32 kernel stages (LDS-DMA, buffer_load_dwordx4 ... lds), and one lane's MFMA fragment
// (8 consecutive k of one row) is one ds_read_b128 for hi and one for lo.
//
// Measured and dropped: the same tile on
// v_mfma_f32_32x32x16_f16 (8 % slower per call: lower clock under load), a 128 x 256 x 16 tile
// with three LDS stages (-25 % staged bytes, prefetch two steps ahead: equal or slower), and a
// start-up stagger between the two workgroups of a CU (no effect).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

namespace gist {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int H3_T = 128;        // block tile edge
constexpr int H3_BK = 32;        // k per tile = 32-bit words per image row
constexpr int H3_IMG = H3_T * H3_BK;              // words per image (16 KiB)
constexpr int H3_BUF_BYTES = 2 * H3_IMG * 4;      // A + B image of one stage

// ---- pre-pass ---------------------------------------------------------------------------
// One power-of-two scale per ROW of a split operand (= per output row for A, per output column
// for B): a row's error depends on its own magnitude only, whatever the spread between rows
// (gradient rows differ by orders of magnitude).
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// x * scale has its largest magnitude in [2^13, 2^14): hi never overflows f16 (65504), lo of
// every element above 2^-17 of the row maximum is a normal f16, and the representation error
// of any element is below max(2^-22 |x|, 2^-39 row max).
__device__ __forceinline__ float h3_scale(unsigned amax_bits, float *inv) {
    const int e = (int)((amax_bits >> 23) & 0xffu);
    int shift = amax_bits == 0u ? 0 : 13 - (e - 127);
    shift = max(-60, min(60, shift));
    *inv = __uint_as_float((unsigned)(127 - shift) << 23);
    return __uint_as_float((unsigned)(127 + shift) << 23);
}

__device__ __forceinline__ void h3_emit(const float (&v)[8], float s, uint32_t *__restrict__ dst) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x0 = v[2 * j] * s, x1 = v[2 * j + 1] * s;
        const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
        const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
        h[j] = (uint32_t)__builtin_bit_cast(unsigned short, h0) |
               ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
        l[j] = (uint32_t)__builtin_bit_cast(unsigned short, l0) |
               ((uint32_t)__builtin_bit_cast(unsigned short, l1) << 16);
    }
    *reinterpret_cast<uint4 *>(dst) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4 *>(dst + 4) = make_uint4(l[0], l[1], l[2], l[3]);
}

__device__ __forceinline__ void h3_load8(const float *__restrict__ p, int k0, int k, float (&v)[8]) {
    if (k0 + 8 <= k) {
        const float4 a = *reinterpret_cast<const float4 *>(p);
        const float4 b = *reinterpret_cast<const float4 *>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = k0 + j < k ? p[j] : 0.f;
    }
}

// source [rows][k] (k contiguous) -> dst [rows][kpad words] + inv[rows]; one workgroup per row:
// pass 1 row maximum, pass 2 (the row is L2/L1-hot) split; a thread handles 8 consecutive k.
__global__ __launch_bounds__(256) void h3_split_rows_kernel(const float *__restrict__ src, int64_t ld,
                                                            int k, uint32_t *__restrict__ dst,
                                                            int64_t ldd, float *__restrict__ inv_out) {
    __shared__ float wmax[4];
    const int64_t r = blockIdx.x;
    const float *row = src + r * ld;
    const int nkb = (int)(ldd >> 3);
    float m = 0.f;
    for (int kb = threadIdx.x; kb < nkb; kb += 256) {
        float v[8];
        h3_load8(row + kb * 8, kb * 8, k, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j]));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    float inv;
    const float s = h3_scale(__float_as_uint(m), &inv);
    if (threadIdx.x == 0) inv_out[r] = inv;
    for (int kb = threadIdx.x; kb < nkb; kb += 256) {
        float v[8];
        h3_load8(row + kb * 8, kb * 8, k, v);
        h3_emit(v, s, dst + r * ldd + kb * 8);
    }
}

// source [k][cols] (cols contiguous): column maxima.  256 k x 64 cols per block.
__global__ __launch_bounds__(256) void h3_colmax_kernel(const float *__restrict__ src, int64_t ld,
                                                        int k, int cols, unsigned *__restrict__ cmax) {
    __shared__ float part[16][65];
    const int c0 = blockIdx.x * 64, k0 = blockIdx.y * 256;
    const int t = threadIdx.x;
    const int c4 = (t & 15) * 4, kr = t >> 4;
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int kk = k0 + kr + 16 * i;
        if (kk < k) {
            const float *p = src + (int64_t)kk * ld + c0 + c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c0 + c4 + 3 < cols) {
                v = *reinterpret_cast<const float4 *>(p);
            } else {
                if (c0 + c4 + 0 < cols) v.x = p[0];
                if (c0 + c4 + 1 < cols) v.y = p[1];
                if (c0 + c4 + 2 < cols) v.z = p[2];
            }
            m.x = fmaxf(m.x, fabsf(v.x)); m.y = fmaxf(m.y, fabsf(v.y));
            m.z = fmaxf(m.z, fabsf(v.z)); m.w = fmaxf(m.w, fabsf(v.w));
        }
    }
    part[kr][c4 + 0] = m.x; part[kr][c4 + 1] = m.y; part[kr][c4 + 2] = m.z; part[kr][c4 + 3] = m.w;
    __syncthreads();
    if (t < 64 && c0 + t < cols) {
        float mm = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) mm = fmaxf(mm, part[i][t]);
        if (mm > 0.f) atomicMax(cmax + c0 + t, __float_as_uint(mm));
    }
}

// source [k][cols] (cols contiguous) -> dst [cols][kpad words] + inv[cols]: 64 k x 64 cols per
// block through LDS; 8 consecutive lanes write 256 contiguous bytes of one output row.
__global__ __launch_bounds__(256) void h3_split_t_kernel(const float *__restrict__ src, int64_t ld,
                                                         int k, int cols, uint32_t *__restrict__ dst,
                                                         int64_t ldd, const unsigned *__restrict__ cmax,
                                                         float *__restrict__ inv_out) {
    __shared__ float tile[64][65];
    __shared__ float cscale[64];
    const int c0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int t = threadIdx.x;
    if (t < 64) {
        float inv = 1.f;
        const float s = c0 + t < cols ? h3_scale(cmax[c0 + t], &inv) : 1.f;
        cscale[t] = s;
        if (blockIdx.y == 0 && c0 + t < cols) inv_out[c0 + t] = inv;
    }
    const int c4 = (t & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kk = (t >> 4) + 16 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k0 + kk < k) {
            const float *p = src + (int64_t)(k0 + kk) * ld + c0 + c4;
            if (c0 + c4 + 3 < cols) {
                v = *reinterpret_cast<const float4 *>(p);
            } else {
                if (c0 + c4 + 0 < cols) v.x = p[0];
                if (c0 + c4 + 1 < cols) v.y = p[1];
                if (c0 + c4 + 2 < cols) v.z = p[2];
            }
        }
        tile[kk][c4 + 0] = v.x; tile[kk][c4 + 1] = v.y;
        tile[kk][c4 + 2] = v.z; tile[kk][c4 + 3] = v.w;
    }
    __syncthreads();
    const int kb = t & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (t >> 3) + 32 * i;
        if (c0 + c < cols && k0 + kb * 8 < (int)ldd) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[kb * 8 + j][c];
            h3_emit(v, cscale[c], dst + (int64_t)(c0 + c) * ldd + k0 + kb * 8);
        }
    }
}

// One read of src[rows, cols] -> the split operand with k = columns (dst_r) and/or the one with
// k = rows (dst_t), 64 x 64 per block through LDS.  Scales: uniform (a given exponent, or from
// the tensor's max |x|), or per row / per column from given maxima.  Dropout is applied on the
// fly with gist_dropout_f32's generator (element index offset + r * cols + c), so the fp32
// dropped tensor never has to exist.
__global__ __launch_bounds__(256) void h3_dual_split_kernel(H3Dual d, int64_t ldd_r, int64_t ldd_t,
                                                            float keep) {
    __shared__ float tile[64][65];
    __shared__ float srow[64], scol[64];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int t = threadIdx.x;
    if (t < 128) {
        const bool is_row = t < 64;
        const int i = t & 63;
        const int64_t g = (is_row ? r0 : c0) + i;
        const float *mx = is_row ? d.rowmax : d.colmax;
        float inv, s;
        if (mx != nullptr) {
            s = g < (is_row ? d.rows : d.cols) ? h3_scale(__float_as_uint(mx[g]), &inv) : 1.f;
        } else if (d.amax != nullptr) {
            s = h3_scale(*d.amax, &inv);
        } else {
            s = __uint_as_float((unsigned)(127 + d.fixed_shift) << 23);
            inv = __uint_as_float((unsigned)(127 - d.fixed_shift) << 23);
        }
        if (is_row) {
            srow[i] = s;
            if (blockIdx.x == 0 && d.inv_r != nullptr && g < d.rows) d.inv_r[g] = inv;
        } else {
            scol[i] = s;
            if (blockIdx.y == 0 && d.inv_t != nullptr && g < d.cols) d.inv_t[g] = inv;
        }
    }
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
            if (c0 + c4 + 3 < d.cols) {
                const float4 q = *reinterpret_cast<const float4 *>(p);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j)
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
            h3_emit(v, srow[q], d.dst_r + (int64_t)(r0 + q) * ldd_r + c0 + kb * 8);
        }
        if (d.dst_t != nullptr && c0 + q < d.cols && r0 + kb * 8 < ldd_t) {      // column q, 8 rows
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[kb * 8 + j][q];
            h3_emit(v, scol[q], d.dst_t + (int64_t)(c0 + q) * ldd_t + r0 + kb * 8);
        }
    }
}

// max |x| of a [rows, cols] window into *out (float bits; zeroed by the caller).  16 rows per
// block, 8 independent 16-byte loads in flight per thread, ONE atomic per block (4 rows per block
// = 4x the atomics on one address ran 32 -> 59 us).
__global__ __launch_bounds__(256) void h3_absmax_kernel(const float *__restrict__ src, int64_t ld,
                                                        int rows, int cols,
                                                        unsigned *__restrict__ out) {
    const int c4n = cols >> 2;
    float m = 0.f;
    __shared__ float wm[4];
    const int rbeg = blockIdx.x * 16, rend = min(rows, rbeg + 16);
    for (int r = rbeg; r < rend; ++r) {
        const float *p = src + (int64_t)r * ld;
        int c = threadIdx.x;
        for (; c + 7 * 256 < c4n; c += 8 * 256) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(p + 4 * (c + 256 * u));
#pragma unroll
            for (int u = 0; u < 8; ++u)
                m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
        }
        for (; c < c4n; c += 256) {
            const float4 v = *reinterpret_cast<const float4 *>(p + 4 * c);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        if ((int)threadIdx.x < (cols & 3)) m = fmaxf(m, fabsf(p[4 * c4n + threadIdx.x]));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > 0.f) atomicMax(out, __float_as_uint(m));
    }
}

// ---- the GEMM -----------------------------------------------------------------------------
struct H3Args {
    const uint32_t *a; int64_t lda;      // split operands, [rows][kpad] 32-bit words
    const uint32_t *b; int64_t ldb;
    const float *inv_a, *inv_b;          // [m], [n]: 2^-e per row of A / of B (split kernels)
    const float *bias;
    float *c; int64_t ldc;
    int m, n, kpad;
    int tiles_m, tiles_n;
};

template <int NI>
__device__ __forceinline__ void h3_dma_image(const uint32_t *ubase, const uint32_t (&off)[NI],
                                             char *image, int wave) {
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(ubase), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int jj = 0; jj < NI; ++jj)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrc, (__attribute__((address_space(3))) void *)(image + (NI * wave + jj) * 1024), 16,
            off[jj], 0, 0, 0);
}

// 128 x 128 x 32 block tile, 256 threads = 4 waves in 2x2, each wave 64x64 = 4x4 tiles of
// v_mfma_f32_16x16x32_f16 (48 MFMAs of 16 cycles per k tile), two LDS stages of 32 KiB, two
// workgroups per CU, LDS-DMA staging issued one k tile ahead.  The 16x16x32 shape is chosen over
// 32x32x16 (same cycles per flop, same LDS traffic) because the chip holds a higher clock
// under it (MI355X_MICROARCH.md, DVFS give-back item 7): measured 8 % faster per call here.
// A lane's fragment: row l&15 of a 16-row slab, k = 8 (l>>4) .. +7  ->  chunk 2 (l>>4) (hi) /
// +1 (lo) of the 128-byte image row.  A ds_read_b128 lane group mixes two k groups
// ({0-3,12-15} with k group g, {20-27} = rows 4-11 with g+1): chunk c of row r sits in slot
// c ^ (bit1(r) | bit3(r) << 2), which keeps both kinds of group on 16 different bank quads
// (SQ_LDS_BANK_CONFLICT = 0).
typedef float h3_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int h3_swz(int r) { return ((r >> 1) & 1) | (((r >> 3) & 1) << 2); }

template <int NI>
__device__ __forceinline__ void h3_dma_offsets(int64_t ld, int rows, int row0, int wave, int lane,
                                                uint32_t (&off)[NI]) {
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) {
        const int p = 64 * (NI * wave + jj) + lane;
        const int r = p >> 3, slot = p & 7;
        const int dr = min(r, rows - 1 - row0);
        off[jj] = (uint32_t)(((int64_t)dr * ld + ((slot ^ h3_swz(r)) << 2)) * 4);
    }
}

// TM = rows of the A tile: 128, or 64 for outputs with fewer than two 128x128 tiles per CU
// (wave tile 32 x 64, 24 MFMAs per k tile, 24 KiB per stage, 512 instead of 256 workgroups on a
// 2046 x 2048 output).
#ifdef H3_CLOCK_PROBE   // dev build (scripts/h3_clock_probe.py): in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz
__device__ unsigned long long g_h3_clock[4 * 4096];
#endif

template <int TM>
__global__ __launch_bounds__(256, 2) void gemm_h3_kernel(H3Args g) {
    extern __shared__ __attribute__((aligned(16))) char h3_smem[];
    constexpr int T = H3_T;                      // columns of the output tile = rows of the B tile
    constexpr int NI = TM / 32;                  // 16-row A slabs per wave (wave tile TM/2 x 64)
    constexpr int A_BYTES = TM * 128;            // A image of one stage
    constexpr int BUF_BYTES = A_BYTES + H3_IMG * 4;

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
    const int row0 = bm * TM, col0 = bn * T;
    const int n_kt = g.kpad / H3_BK;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int rr = lane & 15, kg = lane >> 4;

    h3_f32x4 acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    uint32_t offA[NI], offB[4];
    h3_dma_offsets<NI>(g.lda, g.m, row0, wave, lane, offA);
    h3_dma_offsets<4>(g.ldb, g.n, col0, wave, lane, offB);
    const uint32_t *originA = g.a + (int64_t)row0 * g.lda;
    const uint32_t *originB = g.b + (int64_t)col0 * g.ldb;
    auto dma = [&](int buf, int kt) {
        char *sa = h3_smem + buf * BUF_BYTES;
        h3_dma_image<NI>(originA + (int64_t)kt * H3_BK, offA, sa, wave);
        h3_dma_image<4>(originB + (int64_t)kt * H3_BK, offB, sa + A_BYTES, wave);
    };

    // fragment byte offsets inside an image: [16-row slab][hi/lo]
    int fa[NI][2], fb[4][2];
#pragma unroll
    for (int lo = 0; lo < 2; ++lo) {
        const int c = 2 * kg + lo;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int ar = wm * (TM / 2) + i * 16 + rr;
            fa[i][lo] = ar * 128 + ((c ^ h3_swz(ar)) << 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int br = wn * 64 + i * 16 + rr;
            fb[i][lo] = br * 128 + ((c ^ h3_swz(br)) << 4) + A_BYTES;
        }
    }

    if (n_kt > 0) dma(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();

    auto kstep = [&](auto cur_c, auto next_c, int kt) {
        constexpr int cur = decltype(cur_c)::value;
        constexpr bool has_next = decltype(next_c)::value != 0;
#ifndef H3_DMA_AFTER
#define H3_DMA_AFTER (-1)    // dev knob: MFMAs of the k step issued before the next tile's DMA
#endif
        __builtin_amdgcn_sched_barrier(0);
        const char *img = h3_smem + cur * BUF_BYTES;
        __builtin_amdgcn_s_setprio(1);
#ifndef H3_PROBE_NO_MFMA
        f16x8 ah[NI], al[NI], bh[4], bl[4];
#pragma unroll
        for (int i = 0; i < NI; ++i) ah[i] = *reinterpret_cast<const f16x8 *>(img + fa[i][0]);
#pragma unroll
        for (int i = 0; i < 4; ++i) bh[i] = *reinterpret_cast<const f16x8 *>(img + fb[i][0]);
#pragma unroll
        for (int i = 0; i < 4; ++i) bl[i] = *reinterpret_cast<const f16x8 *>(img + fb[i][1]);
#pragma unroll
        for (int i = 0; i < NI; ++i) al[i] = *reinterpret_cast<const f16x8 *>(img + fa[i][1]);
        // 3 * NI * 4 MFMAs: ah.bh, ah.bl, al.bh.  The next tile's DMA goes out behind the first
        // 16 (the ah.bh group) in the 128-row kernel: issued first thing in the step it delays this
        // wave's own MFMAs by its issue time (8 instructions of 60+ cycles); issued late it is not
        // landed at the barrier (per launch: first 0.2258 ms, after 8 MFMAs 0.2244, after 16 0.2215,
        // after 24 0.2250, after 32 0.2332).  The 64-row kernel (24 MFMAs, 3 workgroups per CU)
        // runs best with the DMA first (h=1024 step 0.525 vs 0.533 ms).
#pragma unroll
        for (int t = 0; t < 3 * NI * 4; ++t) {
            constexpr int dma_after = H3_DMA_AFTER < 0 ? (TM == 128 ? 16 : 0)
                                                        : (H3_DMA_AFTER < 3 * NI * 4 ? H3_DMA_AFTER : 3 * NI * 4 - 1);
            if (t == dma_after) {
                __builtin_amdgcn_sched_barrier(0);
#ifndef H3_PROBE_NO_DMA      // dev probes (scripts/h3_probe.sh): which side bounds the loop
                if constexpr (has_next) dma(cur ^ 1, kt + 1);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            const int term = t / (NI * 4), i = (t % (NI * 4)) / 4, j = t % 4;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(term == 2 ? al[i] : ah[i],
                                                               term == 1 ? bl[j] : bh[j], acc[i][j], 0, 0, 0);
        }
#else
#ifndef H3_PROBE_NO_DMA
        if constexpr (has_next) dma(cur ^ 1, kt + 1);
#endif
#endif
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (has_next) __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
    };
#ifdef H3_CLOCK_PROBE
    const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
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

#ifdef H3_CLOCK_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        g_h3_clock[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime() - ck0;
        g_h3_clock[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif
    // ---- epilogue: C/D of 16x16x32: col = lane & 15, row = 4 (lane >> 4) + e ----
    float *cbase = g.c + (int64_t)row0 * g.ldc + col0;
    const int rows_valid = min(g.m - row0, TM);
    __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
        cbase, 0, (int)((int64_t)rows_valid * g.ldc * 4), 0x00020000);
    const uint32_t ldc_b = (uint32_t)g.ldc * 4;
    uint32_t cvoff[4];
    float bv[4], sb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cl = wn * 64 + j * 16 + rr;
        const bool ok = col0 + cl < g.n;
        cvoff[j] = ok ? (uint32_t)(wm * (TM / 2) + 4 * kg) * ldc_b + (uint32_t)cl * 4 : 0x7fffffffu;
        bv[j] = (g.bias != nullptr && ok) ? g.bias[col0 + cl] : 0.f;
        sb[j] = ok ? g.inv_b[col0 + cl] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int rl = i * 16 + e;
            const uint32_t roff = (uint32_t)rl * ldc_b;
            const int grow = min(row0 + wm * (TM / 2) + 4 * kg + rl, g.m - 1);
            const float sa = g.inv_a[grow];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = fmaf(acc[i][j][e] * sa, sb[j], bv[j]);
                if (rows_valid == TM)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), crsrc, cvoff[j], roff, 0);
                else
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), crsrc, cvoff[j] + roff, 0, 0);
            }
        }
}

template __global__ void gemm_h3_kernel<128>(H3Args);
template __global__ void gemm_h3_kernel<64>(H3Args);

// ---- host side ----------------------------------------------------------------------------
static std::atomic<int> g_h3_mode{-1};      // -1: read GIST_GEMM_MODE on first use
// A sizing query "as if the mode were m" (gist_step_h3_workspace_bytes_mode) sets this for the calling
// thread only: launches of other threads keep seeing the process-wide mode.
thread_local int tl_mode_override = -1;

int h3_mode() {
    if (tl_mode_override >= 0) return tl_mode_override;
    int m = g_h3_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char *e = getenv("GIST_GEMM_MODE");
        // default: large projections as three bf16 pieces per operand (all 24 bits, six cross terms:
        // error at the fp32-MFMA kernel's level, gemm_b3.hip), everything else fp32 MFMA
        m = 2;
        if (e && (!strcmp(e, "f32") || !strcmp(e, "0"))) m = 0;
        else if (e && (!strcmp(e, "f16x3") || !strcmp(e, "1"))) m = 1;
        else if (e && (!strcmp(e, "bf16x3") || !strcmp(e, "2"))) m = 2;
        int expected = -1;
        g_h3_mode.compare_exchange_strong(expected, m, std::memory_order_relaxed);
        m = g_h3_mode.load(std::memory_order_relaxed);
    }
    return m;
}
void h3_mode_override(int mode) { tl_mode_override = mode; }

int64_t h3_kpad(int64_t k) { return ceil_div(k, H3_BK) * H3_BK; }

// Shapes the split path takes: enough 128x128 tiles to occupy the chip and enough flops to
// pay for the pre-pass.  Break-even measured at ~20 GFLOP when every call splits its own
// operands (scripts/h3_bench.py) and at ~4 GFLOP inside the step, which shares one split of an
// operand between the GEMMs that use it (bench.py --n-hidden 1024: 0.606 -> 0.551 ms/step).
// Everything else stays on the fp32 kernel.
static bool h3_shape_ok(int64_t m, int64_t n, int64_t k, double default_min_gflop) {
    if (h3_mode() != 1) return false;
    const double t_gflop = tune(GIST_TUNE_H3_MIN_GFLOP), t_tiles = tune(GIST_TUNE_H3_MIN_TILES);
    const double min_gflop = t_gflop > 0.0 ? t_gflop : default_min_gflop;
    const int min_tiles = t_tiles > 0.0 ? (int)t_tiles : 64;
    // (an explicit tile threshold -- tests -- also lifts the minimum extents: the kernel itself
    // handles any m, n, k >= 1)
    if (t_tiles <= 0.0 && (m < 64 || n < 64 || k < 64)) return false;
    if (m < 1 || n < 1 || k < 1) return false;
    if (ceil_div(m, H3_T) * ceil_div(n, H3_T) < min_tiles) return false;
    if (2.0 * (double)m * (double)n * (double)k < min_gflop * 1e9) return false;
    if (h3_kpad(k) >= (1LL << 22)) return false;
    return true;
}
bool h3_eligible(int64_t m, int64_t n, int64_t k) { return h3_shape_ok(m, n, k, 16.0); }
bool h3_eligible_kept(int64_t m, int64_t n, int64_t k) { return h3_shape_ok(m, n, k, 4.0); }

// workspace: [inv_a: m floats][inv_b: n floats][max bits: m + n] padded to 256 B, then the splits
static inline int64_t h3_head_bytes(int64_t m, int64_t n) { return ceil_div((m + n) * 8, 256) * 256; }

int64_t h3_workspace_bytes(int64_t m, int64_t n, int64_t k) {
    if (!h3_eligible(m, n, k)) return 0;
    return h3_head_bytes(m, n) + (m + n) * h3_kpad(k) * 4;
}

int h3_gemm_presplit(const char *name, const uint32_t *sa, const float *inv_a, const uint32_t *sb,
                     const float *inv_b, const float *bias, float *c, int64_t ldc, int64_t m,
                     int64_t n, int64_t k, hipStream_t st) {
    const int64_t kpad = h3_kpad(k);
    static DeviceOnce once;
    int dev;
    if (once.needed(&dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_h3_kernel<128>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           2 * H3_BUF_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_h3_kernel<64>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * H3_BUF_BYTES);
        if (e != hipSuccess) {
            set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
            return GIST_ELAUNCH;
        }
        once.done(dev);
    }
    H3Args g;
    g.a = sa; g.lda = kpad; g.b = sb; g.ldb = kpad; g.inv_a = inv_a; g.inv_b = inv_b; g.bias = bias;
    g.c = c; g.ldc = ldc; g.m = (int)m; g.n = (int)n; g.kpad = (int)kpad;
    g.tiles_n = (int)ceil_div(n, H3_T);
    // 64-row A tiles when 128-row tiles would leave CUs without a second workgroup
    const int t_tm = (int)tune(GIST_TUNE_H3_TM);      // 0 = auto, 64 / 128 = forced
    const bool tm64 = t_tm ? t_tm == 64 : ceil_div(m, 128) * g.tiles_n < 512;
    g.tiles_m = (int)ceil_div(m, tm64 ? 64 : 128);
    const int64_t slot = timer_begin(tl_timer, 2, m, n, k, st);      // kind 2: the main kernel alone
    if (tm64)
        hipLaunchKernelGGL(gemm_h3_kernel<64>, dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(256),
                           2 * (64 * 128 + H3_IMG * 4), st, g);
    else
        hipLaunchKernelGGL(gemm_h3_kernel<128>, dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(256),
                           2 * H3_BUF_BYTES, st, g);
    timer_end(tl_timer, slot, st);
    return launch_status(name);
}

int h3_split_rows(const float *src, int64_t ld, int64_t rows, int64_t k, uint32_t *dst, float *inv,
                  hipStream_t st) {
    if (rows <= 0) return GIST_OK;
    hipLaunchKernelGGL(h3_split_rows_kernel, dim3((unsigned)rows), dim3(256), 0, st, src, ld, (int)k,
                       dst, h3_kpad(k), inv);
    return launch_status("h3_split_rows");
}

int h3_absmax(const float *src, int64_t ld, int64_t rows, int64_t cols, unsigned *out, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return GIST_OK;
    hipLaunchKernelGGL(h3_absmax_kernel, dim3((unsigned)ceil_div(rows, 16)), dim3(256), 0, st, src, ld,
                       (int)rows, (int)cols, out);
    return launch_status("h3_absmax");
}

int h3_dual_split(const H3Dual &d, hipStream_t st) {
    if (d.rows <= 0 || d.cols <= 0) return GIST_OK;
    const int64_t ldd_r = h3_kpad(d.cols), ldd_t = h3_kpad(d.rows);
    const int64_t gx = ceil_div(d.dst_r ? ldd_r : d.cols, 64), gy = ceil_div(d.dst_t ? ldd_t : d.rows, 64);
    hipLaunchKernelGGL(h3_dual_split_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, st, d,
                       ldd_r, ldd_t, d.p > 0.f ? 1.0f / (1.0f - d.p) : 1.0f);
    return launch_status("h3_dual_split");
}

// A: a_kc ? [m][k] : [k][m];  B: b_kc ? [n][k] : [k][n].  Returns 1 if the GEMM was issued,
// 0 if this call is not for the split path (caller falls back to fp32), < 0 on error.
int h3_gemm(const char *name, bool a_kc, bool b_kc, const float *a, int64_t lda, const float *b,
            int64_t ldb, const float *bias, float *c, int64_t ldc, int64_t m, int64_t n, int64_t k,
            void *ws, int64_t ws_bytes, hipStream_t st) {
    if (!h3_eligible(m, n, k)) return 0;
    if (!(aligned16(a) && aligned16(b) && lda % 4 == 0 && ldb % 4 == 0)) return 0;
    if (ws == nullptr || !aligned16(ws) || ws_bytes < h3_workspace_bytes(m, n, k)) return 0;
    const int64_t kpad = h3_kpad(k);
    float *inv_a = static_cast<float *>(ws), *inv_b = inv_a + m;
    unsigned *max_a = reinterpret_cast<unsigned *>(inv_b + n), *max_b = max_a + m;
    uint32_t *sa = reinterpret_cast<uint32_t *>(static_cast<char *>(ws) + h3_head_bytes(m, n));
    uint32_t *sb = sa + m * kpad;
    if ((!a_kc || !b_kc) && hipMemsetAsync(max_a, 0, (size_t)(m + n) * 4, st) != hipSuccess) {
        set_error("%s: hipMemsetAsync failed", name);
        return GIST_ELAUNCH;
    }
    // k-contiguous source [rows][k]: one fused kernel; [k][rows] source: column maxima, then the
    // transposing split
    auto split = [&](bool kc, const float *src, int64_t ld, int64_t rows, uint32_t *dst,
                     unsigned *mx, float *iv) {
        if (kc) {
            hipLaunchKernelGGL(h3_split_rows_kernel, dim3((unsigned)rows), dim3(256), 0, st, src, ld,
                               (int)k, dst, kpad, iv);
        } else {
            hipLaunchKernelGGL(h3_colmax_kernel,
                               dim3((unsigned)ceil_div(rows, 64), (unsigned)ceil_div(k, 256)),
                               dim3(256), 0, st, src, ld, (int)k, (int)rows, mx);
            hipLaunchKernelGGL(h3_split_t_kernel,
                               dim3((unsigned)ceil_div(rows, 64), (unsigned)ceil_div(kpad, 64)),
                               dim3(256), 0, st, src, ld, (int)k, (int)rows, dst, kpad, mx, iv);
        }
    };
    split(a_kc, a, lda, m, sa, max_a, inv_a);
    split(b_kc, b, ldb, n, sb, max_b, inv_b);

    const int rc = h3_gemm_presplit(name, sa, inv_a, sb, inv_b, bias, c, ldc, m, n, k, st);
    return rc == GIST_OK ? 1 : rc;
}

}  // namespace gist

extern "C" int gist_gemm_set_mode(int mode) {
    GIST_REQUIRE(mode >= 0 && mode <= 2,
                 "gist_gemm_set_mode: mode must be 0 (fp32 MFMA), 1 (f16x3 split) or 2 (bf16x3 split)");
    gist::g_h3_mode.store(mode, std::memory_order_relaxed);
    return GIST_OK;
}

extern "C" int gist_gemm_get_mode(void) { return gist::h3_mode(); }

#ifdef H3_CLOCK_PROBE
extern "C" int gist_h3_clock_read(unsigned long long *out, int64_t n_blocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gist::g_h3_clock), n_blocks * 4 * 8);
}
#endif
