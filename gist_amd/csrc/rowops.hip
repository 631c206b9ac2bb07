// Row-wise (HBM-bound) pieces of one ISTSAGELayer step on gfx950:
// LayerNorm(no affine)+ReLU forward/backward, dropout, bias gradient (column sum),
// softmax cross-entropy, Adam, argmax accuracy.  Reference lines are cited at the
// C-ABI declarations in include/gist_hip.h.
//
// Each kernel streams its rows once or twice with 16-B accesses where alignment
// allows; rows are owned by one wave (d <= 1024) or one 256-thread workgroup.
#include <stdlib.h>

#include "common.h"
#include "class_dw_body.h"
#include "adam_body.h"

namespace gist {

// (wave_sum: common.h)
// Sum over the TPR threads that own one row (TPR = 64: a wave; TPR = 256: the block).
template <int TPR>
__device__ __forceinline__ float row_sum(float v, float *red) {
    v = wave_sum(v);
    if constexpr (TPR == 64) {
        return v;
    } else {
        const int w = threadIdx.x >> 6;
        __syncthreads();                      // red[] may still be read from a previous call
        if ((threadIdx.x & 63) == 0) red[w] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    }
}

// ---------------------------------------------------------------------------
// dropout's generator: counter based (splitmix64 finaliser of the seed-mixed element index); one hash
// serves two consecutive elements (even index -> low word, odd -> high word)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float keep_scale(uint64_t seed, uint64_t idx, float p, float scale) {
    const uint64_t h = splitmix64((idx >> 1) + seed * 0x9E3779B97F4A7C15ULL);
    const uint32_t w = (idx & 1) ? (uint32_t)(h >> 32) : (uint32_t)h;
    const float u = (float)(w >> 8) * (1.0f / 16777216.0f);
    return u >= p ? scale : 0.f;
}

// the mask of gist_dropout_f32 for the 4 consecutive element indices idx0 .. idx0 + 3 (sm = seed *
// golden ratio): two hashes when idx0 is even, three otherwise
__device__ __forceinline__ void drop_quad(float4 &v, uint64_t idx0, uint64_t sm, float p, float scale) {
    const float inv = 1.0f / 16777216.0f;
    const uint64_t pair = idx0 >> 1;
    const uint64_t h0 = splitmix64(pair + sm), h1 = splitmix64(pair + 1 + sm);
    if ((idx0 & 1) == 0) {
        v.x *= ((float)((uint32_t)h0 >> 8) * inv >= p) ? scale : 0.f;
        v.y *= ((float)((uint32_t)(h0 >> 32) >> 8) * inv >= p) ? scale : 0.f;
        v.z *= ((float)((uint32_t)h1 >> 8) * inv >= p) ? scale : 0.f;
        v.w *= ((float)((uint32_t)(h1 >> 32) >> 8) * inv >= p) ? scale : 0.f;
    } else {
        const uint64_t h2 = splitmix64(pair + 2 + sm);
        v.x *= ((float)((uint32_t)(h0 >> 32) >> 8) * inv >= p) ? scale : 0.f;
        v.y *= ((float)((uint32_t)h1 >> 8) * inv >= p) ? scale : 0.f;
        v.z *= ((float)((uint32_t)(h1 >> 32) >> 8) * inv >= p) ? scale : 0.f;
        v.w *= ((float)((uint32_t)h2 >> 8) * inv >= p) ? scale : 0.f;
    }
}

// Dropout folded into a producer (gist_ln_relu_fwd_drop_f32): what the producer stores at row r,
// column c of `out` is multiplied by the mask gist_dropout_f32 would apply to element index
// offset + r * mask_ld + c; the value before the mask goes to out2 (the aggregation's source).
struct DropOut {
    float *out2; int64_t ldo2;      // undropped copy (NULL: none)
    float p, scale;                 // p = 0: no mask (out = the plain value)
    uint64_t sm, offset;            // seed * golden ratio; counter base
    int64_t mask_ld;
    // the pre-norm input still as split-K slabs of its projection (gist_ln_relu_fwd_slabs_f32): row r, column c =
    // sum_s slabs[s * slab_stride + r * d + c] + bias[c], formed here in gist_gemm's own order (n_slabs = 0: y as is)
    const float *slabs; int64_t slab_stride; int n_slabs; const float *bias;
};

// ---------------------------------------------------------------------------
// LayerNorm (no affine, biased variance) + ReLU, forward.  y <- yhat in place.
// ---------------------------------------------------------------------------
template <int TPR, int VEC, bool DROP = false>
__global__ __launch_bounds__(256) void ln_relu_fwd_kernel(float *__restrict__ y, int64_t ldy,
                                                          float *__restrict__ out, int64_t ldo,
                                                          float *__restrict__ rstd_out,
                                                          int n_rows, int d, int use_lynorm,
                                                          int relu, float eps, DropOut dr) {
    __shared__ float red[4];
    constexpr int RPB = 256 / TPR;
    const int row = blockIdx.x * RPB + threadIdx.x / TPR;
    const int t = threadIdx.x % TPR;
    const bool live = row < n_rows;
    float *yr = y + (int64_t)(live ? row : 0) * ldy;
    float *orow = out + (int64_t)(live ? row : 0) * ldo;
    float *o2row = nullptr;
    uint64_t ibase = 0;
    if constexpr (DROP) {
        if (dr.out2 != nullptr) o2row = dr.out2 + (int64_t)(live ? row : 0) * dr.ldo2;
        ibase = dr.offset + (uint64_t)(live ? row : 0) * (uint64_t)dr.mask_ld;
    }
    // the store of one quad / element of the output row: plain, or undropped copy + masked value
    auto put4 = [&](int c, float4 w) {
        if constexpr (DROP) {
            if (o2row != nullptr) *reinterpret_cast<float4 *>(o2row + c) = w;
            if (dr.p > 0.f) drop_quad(w, ibase + (uint64_t)c, dr.sm, dr.p, dr.scale);
        }
        *reinterpret_cast<float4 *>(orow + c) = w;
    };
    auto put1 = [&](int c, float v) {
        if constexpr (DROP) {
            if (o2row != nullptr) o2row[c] = v;
            if (dr.p > 0.f) {
                const uint64_t idx = ibase + (uint64_t)c;
                const uint64_t h = splitmix64((idx >> 1) + dr.sm);
                const uint32_t w = (idx & 1) ? (uint32_t)(h >> 32) : (uint32_t)h;
                v *= ((float)(w >> 8) * (1.0f / 16777216.0f) >= dr.p) ? dr.scale : 0.f;
            }
        }
        orow[c] = v;
    };
    if (dr.n_slabs > 0 && live) {
        // y <- the projection's result; every thread forms exactly the columns it reads back below
        const float *sp = dr.slabs + (int64_t)row * d;
        for (int c = t * VEC; c < d; c += TPR * VEC) {
            if constexpr (VEC == 4) {
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int k = 0; k < dr.n_slabs; ++k) {
                    const float4 q = *reinterpret_cast<const float4 *>(sp + (int64_t)k * dr.slab_stride + c);
                    a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
                }
                if (dr.bias) { a.x += dr.bias[c]; a.y += dr.bias[c + 1]; a.z += dr.bias[c + 2]; a.w += dr.bias[c + 3]; }
                *reinterpret_cast<float4 *>(yr + c) = a;
            } else {
                float a = 0.f;
                for (int k = 0; k < dr.n_slabs; ++k) a += sp[(int64_t)k * dr.slab_stride + c];
                if (dr.bias) a += dr.bias[c];
                yr[c] = a;
            }
        }
    }
    float mean = 0.f, rstd = 1.f;
#ifndef GIST_LN_STREAMING      // dev A/B build flag: always take the streaming path
    if constexpr (TPR == 256 && VEC == 4) {
        // Rows of up to 4096 floats (one workgroup per row): the row is read ONCE into registers
        // (4 x 16 B per thread) instead of three times; same per-thread order of every sum, so
        // the results are bit-identical to the streaming path below.
        if (d <= 4096) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = t * 4 + 1024 * u;
                v[u] = (live && c < d) ? *reinterpret_cast<const float4 *>(yr + c)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (use_lynorm) {
                float s = 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (t * 4 + 1024 * u < d) s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
                mean = row_sum<TPR>(s, red) / (float)d;
                float q = 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (t * 4 + 1024 * u < d) {
                        const float a = v[u].x - mean, b = v[u].y - mean, e = v[u].z - mean, f = v[u].w - mean;
                        q += (a * a + b * b) + (e * e + f * f);
                    }
                const float var = row_sum<TPR>(q, red) / (float)d;
                rstd = 1.0f / sqrtf(var + eps);
                if (live && t == 0 && rstd_out) rstd_out[row] = rstd;
            }
            if (!live) return;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = t * 4 + 1024 * u;
                if (c >= d) break;
                float4 w = v[u];
                w.x = (w.x - mean) * rstd; w.y = (w.y - mean) * rstd;
                w.z = (w.z - mean) * rstd; w.w = (w.w - mean) * rstd;
                if (use_lynorm) *reinterpret_cast<float4 *>(yr + c) = w;
                if (relu) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f);
                            w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
                put4(c, w);
            }
            return;
        }
    }
#endif
    if (use_lynorm) {
        float s = 0.f;
        if (live)
            for (int c = t * VEC; c < d; c += TPR * VEC) {
                if constexpr (VEC == 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(yr + c);
                    s += (v.x + v.y) + (v.z + v.w);
                } else {
                    s += yr[c];
                }
            }
        mean = row_sum<TPR>(s, red) / (float)d;
        float q = 0.f;
        if (live)
            for (int c = t * VEC; c < d; c += TPR * VEC) {
                if constexpr (VEC == 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(yr + c);
                    const float a = v.x - mean, b = v.y - mean, e = v.z - mean, f = v.w - mean;
                    q += (a * a + b * b) + (e * e + f * f);
                } else {
                    const float a = yr[c] - mean;
                    q += a * a;
                }
            }
        const float var = row_sum<TPR>(q, red) / (float)d;
        rstd = 1.0f / sqrtf(var + eps);
        if (live && t == 0 && rstd_out) rstd_out[row] = rstd;
    }
    if (!live) return;
    for (int c = t * VEC; c < d; c += TPR * VEC) {
        if constexpr (VEC == 4) {
            float4 v = *reinterpret_cast<const float4 *>(yr + c);
            v.x = (v.x - mean) * rstd; v.y = (v.y - mean) * rstd;
            v.z = (v.z - mean) * rstd; v.w = (v.w - mean) * rstd;
            if (use_lynorm) *reinterpret_cast<float4 *>(yr + c) = v;
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            put4(c, v);
        } else {
            float v = (yr[c] - mean) * rstd;
            if (use_lynorm) yr[c] = v;
            if (relu) v = fmaxf(v, 0.f);
            put1(c, v);
        }
    }
}

// ---------------------------------------------------------------------------
// backward: g = d_out * [yhat > 0]; dy = rstd * (g - mean(g) - yhat * mean(g*yhat))
// ---------------------------------------------------------------------------
template <int TPR, int VEC>
__global__ __launch_bounds__(256) void ln_relu_bwd_kernel(
    const float *__restrict__ d_out, int64_t ldg, const float *yhat, int64_t ldy,
    const float *__restrict__ rstd_in, float *dy, int64_t lddy, int n_rows, int d,
    int use_lynorm, int relu, float *__restrict__ rowmax) {
    __shared__ float red[4];
    constexpr int RPB = 256 / TPR;
    const int row = blockIdx.x * RPB + threadIdx.x / TPR;
    const int t = threadIdx.x % TPR;
    const bool live = row < n_rows;
    const float *gr = d_out + (int64_t)(live ? row : 0) * ldg;
    const float *yr = yhat + (int64_t)(live ? row : 0) * ldy;
    float *dr = dy + (int64_t)(live ? row : 0) * lddy;
    float m1 = 0.f, m2 = 0.f, rstd = 1.f;
#ifndef GIST_LN_STREAMING
    if constexpr (TPR == 256 && VEC == 4) {
        if (d <= 4096) {      // one read of d_out and yhat into registers (see the forward kernel)
            float4 gq[4], yq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = t * 4 + 1024 * u;
                const bool in = live && c < d;
                gq[u] = in ? *reinterpret_cast<const float4 *>(gr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                yq[u] = in ? *reinterpret_cast<const float4 *>(yr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (relu) {
                    gq[u].x = yq[u].x > 0.f ? gq[u].x : 0.f; gq[u].y = yq[u].y > 0.f ? gq[u].y : 0.f;
                    gq[u].z = yq[u].z > 0.f ? gq[u].z : 0.f; gq[u].w = yq[u].w > 0.f ? gq[u].w : 0.f;
                }
            }
            if (use_lynorm) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (t * 4 + 1024 * u < d) {
                        s1 += (gq[u].x + gq[u].y) + (gq[u].z + gq[u].w);
                        s2 += (gq[u].x * yq[u].x + gq[u].y * yq[u].y) + (gq[u].z * yq[u].z + gq[u].w * yq[u].w);
                    }
                m1 = row_sum<TPR>(s1, red) / (float)d;
                m2 = row_sum<TPR>(s2, red) / (float)d;
                if (live) rstd = rstd_in[row];
            }
            if (!live) return;
            float mx = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = t * 4 + 1024 * u;
                if (c >= d) break;
                float4 o = gq[u];
                if (use_lynorm) {
                    o.x = rstd * (o.x - m1 - yq[u].x * m2); o.y = rstd * (o.y - m1 - yq[u].y * m2);
                    o.z = rstd * (o.z - m1 - yq[u].z * m2); o.w = rstd * (o.w - m1 - yq[u].w * m2);
                }
                *reinterpret_cast<float4 *>(dr + c) = o;
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
            }
            if (rowmax != nullptr) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
                __syncthreads();
                if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
                __syncthreads();
                if (t == 0) rowmax[row] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            }
            return;
        }
    }
#endif
    if (use_lynorm) {
        float s1 = 0.f, s2 = 0.f;
        if (live)
            for (int c = t * VEC; c < d; c += TPR * VEC) {
                if constexpr (VEC == 4) {
                    const float4 g = *reinterpret_cast<const float4 *>(gr + c);
                    const float4 yv = *reinterpret_cast<const float4 *>(yr + c);
                    const float g0 = (!relu || yv.x > 0.f) ? g.x : 0.f;
                    const float g1 = (!relu || yv.y > 0.f) ? g.y : 0.f;
                    const float g2 = (!relu || yv.z > 0.f) ? g.z : 0.f;
                    const float g3 = (!relu || yv.w > 0.f) ? g.w : 0.f;
                    s1 += (g0 + g1) + (g2 + g3);
                    s2 += (g0 * yv.x + g1 * yv.y) + (g2 * yv.z + g3 * yv.w);
                } else {
                    const float yv = yr[c];
                    const float g = (!relu || yv > 0.f) ? gr[c] : 0.f;
                    s1 += g;
                    s2 += g * yv;
                }
            }
        m1 = row_sum<TPR>(s1, red) / (float)d;
        m2 = row_sum<TPR>(s2, red) / (float)d;
        if (live) rstd = rstd_in[row];
    }
    if (!live) return;
    float mx = 0.f;      // max |dy| of the row (scale of the split projection operand, gemm_h3.hip)
    for (int c = t * VEC; c < d; c += TPR * VEC) {
        if constexpr (VEC == 4) {
            const float4 g = *reinterpret_cast<const float4 *>(gr + c);
            const float4 yv = *reinterpret_cast<const float4 *>(yr + c);
            float4 o;
            o.x = (!relu || yv.x > 0.f) ? g.x : 0.f;
            o.y = (!relu || yv.y > 0.f) ? g.y : 0.f;
            o.z = (!relu || yv.z > 0.f) ? g.z : 0.f;
            o.w = (!relu || yv.w > 0.f) ? g.w : 0.f;
            if (use_lynorm) {
                o.x = rstd * (o.x - m1 - yv.x * m2); o.y = rstd * (o.y - m1 - yv.y * m2);
                o.z = rstd * (o.z - m1 - yv.z * m2); o.w = rstd * (o.w - m1 - yv.w * m2);
            }
            *reinterpret_cast<float4 *>(dr + c) = o;
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
        } else {
            const float yv = yr[c];
            float o = (!relu || yv > 0.f) ? gr[c] : 0.f;
            if (use_lynorm) o = rstd * (o - m1 - yv * m2);
            dr[c] = o;
            mx = fmaxf(mx, fabsf(o));
        }
    }
    if (rowmax != nullptr) {      // uniform per launch; a row's threads are all live here
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if constexpr (TPR == 256) {
            __syncthreads();
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
            __syncthreads();
            mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        }
        if (t == 0) rowmax[row] = mx;
    }
}

// ---------------------------------------------------------------------------
// backward + bias-gradient partials in one pass (d <= 1024, 16-byte rows): a workgroup owns
// kLnCsRows = 16 consecutive rows, wave w the rows r0 + w + 4 j (j = 0..3; all four read before the
// first is reduced).  dy of a row is formed exactly as in ln_relu_bwd_kernel<64, 4>; every lane also
// adds what it stores into per-column sums (rows in j order), the four waves' sums meet in LDS in wave
// order, and the workgroup writes col_partials[blockIdx.x][0..d): the bias gradient is the sum of the
// chunks' rows in chunk order (gist_adam_segments_f32 forms it; gist_colsum_chunks_f32 on its own).
// U = ceil(d / 256) quads per lane.
// ---------------------------------------------------------------------------
constexpr int kLnCsRows = 16;

template <int U>
__device__ __forceinline__ void ln_relu_bwd_cs_block(
    const float *__restrict__ d_out, int64_t ldg, const float *yhat, int64_t ldy,
    const float *__restrict__ rstd_in, float *dy, int64_t lddy, int n_rows, int d, int use_lynorm,
    int relu, float *__restrict__ col_partials, const int bx) {
    __shared__ __attribute__((aligned(16))) float part[3][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = bx * kLnCsRows;
    float4 gq[4][U], yq[4][U];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = r0 + wave + 4 * j;
        const bool live = row < n_rows;
        const float *gr = d_out + (int64_t)(live ? row : 0) * ldg;
        const float *yr = yhat + (int64_t)(live ? row : 0) * ldy;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = lane * 4 + 256 * u;
            const bool in = live && c < d;
            gq[j][u] = in ? *reinterpret_cast<const float4 *>(gr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            yq[j][u] = in ? *reinterpret_cast<const float4 *>(yr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float4 cs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cs[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = r0 + wave + 4 * j;
        const bool live = row < n_rows;                      // wave-uniform
        float m1 = 0.f, m2 = 0.f, rstd = 1.f;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (relu) {
                gq[j][u].x = yq[j][u].x > 0.f ? gq[j][u].x : 0.f; gq[j][u].y = yq[j][u].y > 0.f ? gq[j][u].y : 0.f;
                gq[j][u].z = yq[j][u].z > 0.f ? gq[j][u].z : 0.f; gq[j][u].w = yq[j][u].w > 0.f ? gq[j][u].w : 0.f;
            }
        if (use_lynorm) {
            float s1 = 0.f, s2 = 0.f;
            if (live) {                                      // same per-lane order as ln_relu_bwd_kernel<64, 4>
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (lane * 4 + 256 * u < d) {
                        const float4 g = gq[j][u], yv = yq[j][u];
                        s1 += (g.x + g.y) + (g.z + g.w);
                        s2 += (g.x * yv.x + g.y * yv.y) + (g.z * yv.z + g.w * yv.w);
                    }
            }
            m1 = wave_sum(s1) / (float)d;
            m2 = wave_sum(s2) / (float)d;
            if (live) rstd = rstd_in[row];
        }
        if (!live) continue;
        float *dr = dy + (int64_t)row * lddy;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = lane * 4 + 256 * u;
            if (c >= d) break;
            float4 o = gq[j][u];
            if (use_lynorm) {
                const float4 yv = yq[j][u];
                o.x = rstd * (o.x - m1 - yv.x * m2); o.y = rstd * (o.y - m1 - yv.y * m2);
                o.z = rstd * (o.z - m1 - yv.z * m2); o.w = rstd * (o.w - m1 - yv.w * m2);
            }
            *reinterpret_cast<float4 *>(dr + c) = o;
            cs[u].x += o.x; cs[u].y += o.y; cs[u].z += o.z; cs[u].w += o.w;
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) *reinterpret_cast<float4 *>(&part[wave - 1][lane * 4 + 256 * u]) = cs[u];
    }
    __syncthreads();
    if (wave == 0) {
        float *pr = col_partials + (int64_t)bx * d;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = lane * 4 + 256 * u;
            if (c >= d) break;
            float4 o = cs[u];
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const float4 q = *reinterpret_cast<const float4 *>(&part[w][c]);
                o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
            }
            *reinterpret_cast<float4 *>(pr + c) = o;
        }
    }
}

template <int U>
__global__ __launch_bounds__(256) void ln_relu_bwd_cs_kernel(
    const float *__restrict__ d_out, int64_t ldg, const float *yhat, int64_t ldy,
    const float *__restrict__ rstd_in, float *dy, int64_t lddy, int n_rows, int d, int use_lynorm,
    int relu, float *__restrict__ col_partials) {
    ln_relu_bwd_cs_block<U>(d_out, ldg, yhat, ldy, rstd_in, dy, lddy, n_rows, d, use_lynorm, relu, col_partials,
                            (int)blockIdx.x);
}

// The same workgroups and, IN FRONT of them in the grid, those of the class layer's weight gradient
// (class_dw_body.h: dW = dlogits^T . Z as 128-row slabs).  In the fused step the two are independent -- the slabs need
// dlogits and the class layer's input, this backward the reverse aggregation of the class layer's dZ -- and each alone
// fills a fraction of the chip for ~6 us (n / 16 = 128 workgroups here, K / 64 x n / 128 there): one launch instead of two.
template <int U>
__global__ __launch_bounds__(256) void ln_relu_bwd_cs_dw_kernel(
    const float *__restrict__ d_out, int64_t ldg, const float *yhat, int64_t ldy,
    const float *__restrict__ rstd_in, float *dy, int64_t lddy, int n_rows, int d, int use_lynorm,
    int relu, float *__restrict__ col_partials, ClassDwArgs w, int n_dw, int dw_gx) {
    const int b = (int)blockIdx.x;
    if (b < n_dw) {
        class_dw_block(w, b % dw_gx, b / dw_gx);
        return;
    }
    ln_relu_bwd_cs_block<U>(d_out, ldg, yhat, ldy, rstd_in, dy, lddy, n_rows, d, use_lynorm, relu, col_partials,
                            b - n_dw);
}

__global__ __launch_bounds__(256) void class_dw_only_kernel(ClassDwArgs a) { class_dw_block(a, (int)blockIdx.x, (int)blockIdx.y); }

// column sums of g per 16-row chunk, in the order of the fused producers: INTERLEAVED (the LayerNorm
// backward: rows r0 + w + 4 j summed over j for w = 0..3, then ((w0 + w1) + w2) + w3) or sequential (the
// class layer's dZ kernel).  Fallback producer for shapes the fused kernels do not take.
template <bool INTERLEAVED>
__global__ void colsum_rows16_kernel(const float *__restrict__ g, int64_t ldg, int n_rows, int d,
                                     float *__restrict__ partials) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d) return;
    const int r0 = blockIdx.y * 16;
    auto at = [&](int r) { return r < n_rows ? g[(int64_t)r * ldg + c] : 0.f; };
    float out;
    if constexpr (INTERLEAVED) {
        float a[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            a[w] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) a[w] += at(r0 + w + 4 * j);
        }
        out = ((a[0] + a[1]) + a[2]) + a[3];
    } else {
        out = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) out += at(r0 + r);
    }
    partials[(int64_t)blockIdx.y * d + c] = out;
}

// out[j] = the chunks' sum in the order gist_adam_segments_f32 forms it: four partial sums over the
// chunks q, q + 4, ... (ascending), added in q order
__global__ __launch_bounds__(256) void colsum_chunks_kernel(const float *__restrict__ partials, int chunks, int d,
                                                           float *__restrict__ out) {
    __shared__ float part[3][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + e;
    float acc = 0.f;
    if (c < d) {
        int k = q;
        for (; k + 28 < chunks; k += 32) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = partials[(int64_t)(k + 4 * u) * d + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += t[u];
        }
        for (; k < chunks; k += 4) acc += partials[(int64_t)k * d + c];
    }
    if (q > 0) part[q - 1][e] = acc;
    __syncthreads();
    if (q == 0 && c < d) out[c] = ((acc + part[0][e]) + part[1][e]) + part[2][e];
}

// ---------------------------------------------------------------------------
// dropout: counter based (splitmix64 finaliser of seed-mixed element index)
// ---------------------------------------------------------------------------
// Fast path (d % 4 == 0, 16-B aligned rows, even offset): one thread = 4 consecutive
// elements of one row = two hashes, one 16-B load and store, one 64-bit division.
__global__ void dropout_vec4_kernel(float *__restrict__ z, int64_t ldz, int64_t n_rows, int64_t d,
                                    float p, float scale, uint64_t seed, uint64_t offset) {
    const int64_t quads = n_rows * (d >> 2);
    const uint64_t sm = seed * 0x9E3779B97F4A7C15ULL;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads;
         q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e0 = q << 2;
        const int64_t r = e0 / d, c = e0 - r * d;
        float4 *ptr = reinterpret_cast<float4 *>(z + r * ldz + c);
        float4 v = *ptr;
        const uint64_t pair = (offset + (uint64_t)e0) >> 1;
        const uint64_t h0 = splitmix64(pair + sm), h1 = splitmix64(pair + 1 + sm);
        const float inv = 1.0f / 16777216.0f;
        v.x *= ((float)((uint32_t)h0 >> 8) * inv >= p) ? scale : 0.f;
        v.y *= ((float)((uint32_t)(h0 >> 32) >> 8) * inv >= p) ? scale : 0.f;
        v.z *= ((float)((uint32_t)h1 >> 8) * inv >= p) ? scale : 0.f;
        v.w *= ((float)((uint32_t)(h1 >> 32) >> 8) * inv >= p) ? scale : 0.f;
        *ptr = v;
    }
}

__global__ void dropout_kernel(float *__restrict__ z, int64_t ldz, int64_t n_rows, int64_t d,
                               float p, float scale, uint64_t seed, uint64_t offset) {
    const int64_t total = n_rows * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / d, c = i - r * d;
        float *q = z + r * ldz + c;
        *q = *q * keep_scale(seed, offset + (uint64_t)i, p, scale);
    }
}

// ---------------------------------------------------------------------------
// dZ of a narrow layer (the class layer: dY is [n, C], C <= 64) with its dropout mask:
//   dz[i][j] = keep(i, j) * sum_c dy[i][c] * w[c][j]
// The product is 2nCK flops against 4nK bytes of output: store bound.  A block owns 16 rows x
// 1024 columns; dy's 16 rows sit in LDS ([c][row]: one broadcast ds_read_b128 per 4 rows), a
// thread keeps 16 rows x 4 columns of sums, reads each w[c][j..j+3] once and applies the mask
// of gist_dropout_f32 (same generator, same element index) to what it stores.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void narrow_nn_drop_kernel(
    const float *__restrict__ dy, int64_t lddy, const float *__restrict__ w, int64_t ldw,
    float *__restrict__ dz, int64_t lddz, int n_rows, int n_cols, int kc, float p, float scale,
    uint64_t seed, uint64_t offset, float *__restrict__ dy_col_partials) {
    __shared__ __attribute__((aligned(16))) float sdy[64][16];
    const int r0 = blockIdx.y * 16;
    const int c0 = blockIdx.x * 1024 + threadIdx.x * 4;
    for (int t = threadIdx.x; t < 64 * 16; t += 256) {
        const int c = t >> 4, r = t & 15;
        sdy[c][r] = (c < kc && r0 + r < n_rows) ? dy[(int64_t)(r0 + r) * lddy + c] : 0.f;
    }
    __syncthreads();
    // column sums of this 16-row chunk of dy (the layer's bias gradient in 16-row chunks), rows in order
    if (dy_col_partials != nullptr && blockIdx.x == 0 && threadIdx.x < kc) {
        float sacc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += sdy[threadIdx.x][r];
        dy_col_partials[(int64_t)blockIdx.y * kc + threadIdx.x] = sacc;
    }
    if (c0 >= n_cols) return;
    float4 acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fma_row = [&](int c, const float4 &wv) {      // acc[r] += dy[r][c] * w[c][j..j+3]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 d4 = *reinterpret_cast<const float4 *>(&sdy[c][4 * q]);
            const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 &a = acc[4 * q + u];
                a.x = fmaf(dv[u], wv.x, a.x); a.y = fmaf(dv[u], wv.y, a.y);
                a.z = fmaf(dv[u], wv.z, a.z); a.w = fmaf(dv[u], wv.w, a.w);
            }
        }
    };
    int c = 0;
    for (; c + 4 <= kc; c += 4) {      // four w rows in flight: the loop is L2-latency bound otherwise
        float4 wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            wv[u] = *reinterpret_cast<const float4 *>(w + (int64_t)(c + u) * ldw + c0);
#pragma unroll
        for (int u = 0; u < 4; ++u) fma_row(c + u, wv[u]);
    }
    for (; c < kc; ++c)
        fma_row(c, *reinterpret_cast<const float4 *>(w + (int64_t)c * ldw + c0));
    const uint64_t sm = seed * 0x9E3779B97F4A7C15ULL;
    const float inv24 = 1.0f / 16777216.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (r0 + r >= n_rows) break;
        float4 v = acc[r];
        if (p > 0.f) {      // offset and n_cols are even, c0 % 4 == 0: two hashes cover the quad
            const uint64_t pair = (offset + (uint64_t)(r0 + r) * (uint64_t)n_cols + (uint64_t)c0) >> 1;
            const uint64_t h0 = splitmix64(pair + sm), h1 = splitmix64(pair + 1 + sm);
            v.x *= ((float)((uint32_t)h0 >> 8) * inv24 >= p) ? scale : 0.f;
            v.y *= ((float)((uint32_t)(h0 >> 32) >> 8) * inv24 >= p) ? scale : 0.f;
            v.z *= ((float)((uint32_t)h1 >> 8) * inv24 >= p) ? scale : 0.f;
            v.w *= ((float)((uint32_t)(h1 >> 32) >> 8) * inv24 >= p) ? scale : 0.f;
        }
        *reinterpret_cast<float4 *>(dz + (int64_t)(r0 + r) * lddz + c0) = v;
    }
}

// ---------------------------------------------------------------------------
// column sum (bias gradient), two deterministic stages
// ---------------------------------------------------------------------------
constexpr int kColsumRows = 64;

// stage 1: workgroup = 64 columns x 4 row lanes over a chunk of 64 rows; a wave reads 64
// consecutive floats of one row (256 B), four independent row reads in flight per thread,
// row lanes combined through LDS in fixed order.
template <bool WITH_MAX>
__global__ __launch_bounds__(256) void colsum_stage1_kernel(const float *__restrict__ g,
                                                            int64_t ldg, int n_rows, int d,
                                                            float *__restrict__ partials,
                                                            float *__restrict__ pmax) {
    __shared__ float red[3][64];
    __shared__ float redm[3][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * kColsumRows;
    const int r1 = min(n_rows, r0 + kColsumRows);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, mx = 0.f;
    if (c < d) {
        const float *p = g + c;
        int r = r0 + rl;
        for (; r + 12 < r1; r += 16) {
            const float a0 = p[(int64_t)r * ldg], a1 = p[(int64_t)(r + 4) * ldg];
            const float a2 = p[(int64_t)(r + 8) * ldg], a3 = p[(int64_t)(r + 12) * ldg];
            s0 += a0; s1 += a1; s2 += a2; s3 += a3;
            if constexpr (WITH_MAX)
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(a0), fabsf(a1)), fmaxf(fabsf(a2), fabsf(a3))));
        }
        for (; r < r1; r += 4) {
            const float a0 = p[(int64_t)r * ldg];
            s0 += a0;
            if constexpr (WITH_MAX) mx = fmaxf(mx, fabsf(a0));
        }
    }
    const float s = (s0 + s1) + (s2 + s3);
    if (rl > 0) {
        red[rl - 1][cl] = s;
        if constexpr (WITH_MAX) redm[rl - 1][cl] = mx;
    }
    __syncthreads();
    if (rl == 0 && c < d) {
        partials[(int64_t)blockIdx.y * d + c] = ((s + red[0][cl]) + red[1][cl]) + red[2][cl];
        if constexpr (WITH_MAX)
            pmax[(int64_t)blockIdx.y * d + c] =
                fmaxf(fmaxf(mx, redm[0][cl]), fmaxf(redm[1][cl], redm[2][cl]));
    }
}

__global__ void colsum_stage2_kernel(const float *__restrict__ partials, int chunks, int d,
                                     float *__restrict__ out, const float *__restrict__ pmax,
                                     float *__restrict__ outmax) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d) return;
    if (outmax != nullptr) {
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
        int k = 0;
        for (; k + 3 < chunks; k += 4) {
            m0 = fmaxf(m0, pmax[(int64_t)k * d + c]);
            m1 = fmaxf(m1, pmax[(int64_t)(k + 1) * d + c]);
            m2 = fmaxf(m2, pmax[(int64_t)(k + 2) * d + c]);
            m3 = fmaxf(m3, pmax[(int64_t)(k + 3) * d + c]);
        }
        for (; k < chunks; ++k) m0 = fmaxf(m0, pmax[(int64_t)k * d + c]);
        outmax[c] = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < chunks; k += 4) {
        s0 += partials[(int64_t)k * d + c];
        s1 += partials[(int64_t)(k + 1) * d + c];
        s2 += partials[(int64_t)(k + 2) * d + c];
        s3 += partials[(int64_t)(k + 3) * d + c];
    }
    for (; k < chunks; ++k) s0 += partials[(int64_t)k * d + c];
    out[c] = (s0 + s1) + (s2 + s3);
}

// ---------------------------------------------------------------------------
// softmax cross entropy (mean over masked rows) + gradient; one wave per row
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// slabs != NULL: the logits are still the split-K partial sums of the class layer's projection (dense
// [n_slabs][n_rows][n_classes]); the row's logits = slabs summed in slab order + bias are formed here
// exactly as the split-K reduce pass would (gemm.hip) and stored to `logits` first.
__global__ __launch_bounds__(256) void xent_grad_kernel(
    float *logits, int64_t ldl, const int32_t *__restrict__ labels,
    const uint8_t *__restrict__ mask, float inv_count, float *__restrict__ d_logits, int64_t ldg,
    float *__restrict__ row_nll, int n_rows, int n_classes, const float *__restrict__ slabs,
    int64_t slab_stride, int n_slabs, const float *__restrict__ bias) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    float *lr = logits + (int64_t)row * ldl;
    float *gr = d_logits + (int64_t)row * ldg;
    const bool on = mask ? mask[row] != 0 : true;
    if (slabs != nullptr) {
        for (int c = lane; c < n_classes; c += 64) {
            const int64_t i = (int64_t)row * n_classes + c;
            float sacc = 0.f;
            for (int k = 0; k < n_slabs; ++k) sacc += slabs[k * slab_stride + i];
            if (bias) sacc += bias[c];
            lr[c] = sacc;                  // read back below by the same lane only
        }
    }
    float mx = -INFINITY;
    for (int c = lane; c < n_classes; c += 64) mx = fmaxf(mx, lr[c]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < n_classes; c += 64) se += expf(lr[c] - mx);
    se = wave_sum(se);
    const int lab = labels[row];
    const float inv = 1.f / se;
    for (int c = lane; c < ldg; c += 64) {
        float gval = 0.f;
        if (on && c < n_classes) gval = (expf(lr[c] - mx) * inv - (c == lab ? 1.f : 0.f)) * inv_count;
        gr[c] = gval;
    }
    if (lane == 0) row_nll[row] = on ? -((lr[lab] - mx) - logf(se)) : 0.f;
}

__global__ __launch_bounds__(256) void xent_loss_kernel(const float *__restrict__ row_nll,
                                                        int n_rows, float inv_count,
                                                        float *__restrict__ loss) {
    __shared__ float red[4];
    loss_reduce_256(row_nll, n_rows, inv_count, loss, red);
}

// ---------------------------------------------------------------------------
// Adam (torch.optim.Adam semantics, coupled weight decay) over a flat arena
// ---------------------------------------------------------------------------
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                            float *__restrict__ m, float *__restrict__ v, int64_t n, float beta1,
                            float beta2, float eps, float wd, float step_size,
                            float inv_bc2_sqrt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float pv = p[i];
        float gv = g[i];
        if (wd != 0.f) gv = fmaf(wd, pv, gv);
        const float mv = m[i] + (1.f - beta1) * (gv - m[i]);        // lerp_ like torch
        const float vv = beta2 * v[i] + (1.f - beta2) * gv * gv;
        m[i] = mv;
        v[i] = vv;
        const float denom = sqrtf(vv) * inv_bc2_sqrt + eps;
        p[i] = pv - step_size * (mv / denom);
    }
}

// Adam over the arena with DEFERRED gradient reductions (gist_adam_segments_f32): device code in adam_body.h.
// One 256-thread workgroup per virtual block; the last workgroup reduces the step's loss when row_nll != NULL.
__global__ __launch_bounds__(256) void adam_segments_kernel(AdamArgs A) {
    __shared__ float red[4];
    const int64_t n_blocks = adam_arena_blocks(A.n) + A.segs.n_ded;
    if ((int64_t)blockIdx.x >= n_blocks) {
        if (A.row_nll != nullptr) loss_reduce_256(A.row_nll, A.n_loss_rows, A.inv_count, A.loss, red);
        return;
    }
    adam_virtual_block(A, blockIdx.x, threadIdx.x);
}

// The deferred gradient sums WITHOUT the optimiser (gist_grad_segments_finish_f32): the same device code with the update
// left out.  Grid: the inline segments' own arena chunks, then the dedicated blocks.
__global__ __launch_bounds__(256) void grad_finish_kernel(AdamArgs A) {
    const int b = (int)blockIdx.x;
    if (b >= A.segs.n_fin) {
        adam_virtual_block<false>(A, adam_arena_blocks(A.n) + (b - A.segs.n_fin), threadIdx.x);
        return;
    }
    int sg = -1;                                           // uniform: the inline segment this block serves
    for (int s = 0; s < A.segs.n; ++s)
        if (A.segs.fin_first[s] >= 0 && A.segs.fin_first[s] <= b && (sg < 0 || A.segs.fin_first[s] > A.segs.fin_first[sg]))
            sg = s;
    if (sg < 0) return;
    adam_virtual_block<false>(A, A.segs.begin[sg] / 1024 + (b - A.segs.fin_first[sg]), threadIdx.x, sg);
}

// ---------------------------------------------------------------------------
// accuracy: first-max argmax like numpy; integer atomics => deterministic
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void argmax_correct_kernel(
    const float *__restrict__ logits, int64_t ldl, const int32_t *__restrict__ labels,
    const uint8_t *__restrict__ mask, int32_t *__restrict__ correct, int n_rows, int n_classes) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    if (mask && !mask[row]) return;
    const int lane = threadIdx.x & 63;
    const float *lr = logits + (int64_t)row * ldl;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    for (int c = lane; c < n_classes; c += 64) {
        const float v = lr[c];
        if (v > best) { best = v; arg = c; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oa = __shfl_xor(arg, off);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (lane == 0 && arg == labels[row]) atomicAdd(correct, 1);
}

}  // namespace gist

// ===========================================================================
using namespace gist;

namespace gist {
static int ln_relu_fwd_ex(const char *name, float *y, int64_t ldy, float *out, int64_t ldo, float *rstd,
                          int64_t n_rows, int64_t d, int use_lynorm, int relu, float eps,
                          const DropOut *drop, hipStream_t st) {
    if (n_rows < 0 || d < 0) { set_error("%s: negative size", name); return GIST_EINVAL; }
    if (n_rows == 0 || d == 0) return GIST_OK;
    if (!y || !out) { set_error("%s: null pointer", name); return GIST_EINVAL; }
    if (ldy < d || ldo < d) { set_error("%s: leading dimension < d", name); return GIST_EINVAL; }
    if (n_rows >= (1LL << 31) || d >= (1LL << 31)) { set_error("%s: size >= 2^31", name); return GIST_EINVAL; }
    bool v4 = d % 4 == 0 && ldy % 4 == 0 && ldo % 4 == 0 && aligned16(y) && aligned16(out);
    DropOut dr{};
    const bool dropping = drop != nullptr && (drop->p > 0.f || drop->out2 != nullptr);
    if (drop != nullptr && drop->n_slabs > 0) {
        if (drop->slabs == nullptr || drop->slab_stride < n_rows * d) { set_error("%s: bad slabs", name); return GIST_EINVAL; }
        dr.slabs = drop->slabs; dr.slab_stride = drop->slab_stride; dr.n_slabs = drop->n_slabs; dr.bias = drop->bias;
        v4 = v4 && aligned16(drop->slabs) && drop->slab_stride % 4 == 0;
    }
    if (dropping) {
        const DropOut keep = dr;
        dr = *drop;
        dr.slabs = keep.slabs; dr.slab_stride = keep.slab_stride; dr.n_slabs = keep.n_slabs; dr.bias = keep.bias;
        if (dr.out2 != nullptr) {
            if (dr.ldo2 < d) { set_error("%s: leading dimension < d", name); return GIST_EINVAL; }
            v4 = v4 && dr.ldo2 % 4 == 0 && aligned16(dr.out2);
        }
    }
    const bool wide = d > 1024;
    const unsigned grid = (unsigned)(wide ? n_rows : ceil_div(n_rows, 4));
#define L(TPR, V)                                                                                   \
    do {                                                                                            \
        if (dropping)                                                                               \
            hipLaunchKernelGGL((ln_relu_fwd_kernel<TPR, V, true>), dim3(grid), dim3(256), 0, st, y, \
                               ldy, out, ldo, rstd, (int)n_rows, (int)d, use_lynorm, relu, eps, dr); \
        else                                                                                        \
            hipLaunchKernelGGL((ln_relu_fwd_kernel<TPR, V, false>), dim3(grid), dim3(256), 0, st, y, \
                               ldy, out, ldo, rstd, (int)n_rows, (int)d, use_lynorm, relu, eps, dr); \
    } while (0)
    if (wide) { if (v4) L(256, 4); else L(256, 1); }
    else { if (v4) L(64, 4); else L(64, 1); }
#undef L
    return launch_status(name);
}
}  // namespace gist

extern "C" int gist_ln_relu_fwd_f32(float *y, int64_t ldy, float *out, int64_t ldo, float *rstd,
                                    int64_t n_rows, int64_t d, int use_lynorm, int relu,
                                    float eps, gist_stream_t stream) {
    return gist::ln_relu_fwd_ex("gist_ln_relu_fwd_f32", y, ldy, out, ldo, rstd, n_rows, d, use_lynorm, relu,
                                eps, nullptr, gist::as_stream(stream));
}

extern "C" int gist_ln_relu_fwd_drop_f32(float *y, int64_t ldy, float *out, int64_t ldo, float *out2,
                                         int64_t ldo2, float *rstd, int64_t n_rows, int64_t d,
                                         int use_lynorm, int relu, float eps, float p, uint64_t seed,
                                         uint64_t offset, int64_t mask_ld, gist_stream_t stream) {
    GIST_REQUIRE(p >= 0.f && p < 1.f, "gist_ln_relu_fwd_drop_f32: p must be in [0,1)");
    GIST_REQUIRE(mask_ld >= d || p == 0.f, "gist_ln_relu_fwd_drop_f32: mask_ld < d");
    gist::DropOut dr{};
    dr.out2 = out2; dr.ldo2 = ldo2; dr.p = p; dr.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.f;
    dr.sm = seed * 0x9E3779B97F4A7C15ULL; dr.offset = offset; dr.mask_ld = mask_ld;
    return gist::ln_relu_fwd_ex("gist_ln_relu_fwd_drop_f32", y, ldy, out, ldo, rstd, n_rows, d, use_lynorm,
                                relu, eps, &dr, gist::as_stream(stream));
}

extern "C" int gist_ln_relu_fwd_slabs_f32(float *y, int64_t ldy, const float *slabs, int64_t slab_stride,
                                          int32_t n_slabs, const float *bias, float *out, int64_t ldo, float *out2,
                                          int64_t ldo2, float *rstd, int64_t n_rows, int64_t d, int use_lynorm,
                                          int relu, float eps, float p, uint64_t seed, uint64_t offset,
                                          int64_t mask_ld, gist_stream_t stream) {
    GIST_REQUIRE(p >= 0.f && p < 1.f, "gist_ln_relu_fwd_slabs_f32: p must be in [0,1)");
    GIST_REQUIRE(mask_ld >= d || p == 0.f, "gist_ln_relu_fwd_slabs_f32: mask_ld < d");
    GIST_REQUIRE(n_slabs >= 0 && n_slabs < 4096, "gist_ln_relu_fwd_slabs_f32: bad n_slabs");
    GIST_REQUIRE(n_slabs == 0 || slabs != nullptr, "gist_ln_relu_fwd_slabs_f32: null slabs");
    gist::DropOut dr{};
    dr.out2 = out2; dr.ldo2 = ldo2; dr.p = p; dr.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.f;
    dr.sm = seed * 0x9E3779B97F4A7C15ULL; dr.offset = offset; dr.mask_ld = mask_ld;
    dr.slabs = slabs; dr.slab_stride = slab_stride; dr.n_slabs = n_slabs; dr.bias = n_slabs > 0 ? bias : nullptr;
    return gist::ln_relu_fwd_ex("gist_ln_relu_fwd_slabs_f32", y, ldy, out, ldo, rstd, n_rows, d, use_lynorm,
                                relu, eps, &dr, gist::as_stream(stream));
}

namespace gist {
// + rowmax (NULL or [n_rows]): max |dy| per row, for the split projection path (gemm_h3.hip)
int ln_relu_bwd_ex(const float *d_out, int64_t ldg, const float *yhat, int64_t ldy, const float *rstd,
                   float *dy, int64_t lddy, int64_t n_rows, int64_t d, int use_lynorm, int relu,
                   float *rowmax, hipStream_t st) {
    GIST_REQUIRE(n_rows >= 0 && d >= 0, "gist_ln_relu_bwd_f32: negative size");
    if (n_rows == 0 || d == 0) return GIST_OK;
    GIST_REQUIRE(d_out && yhat && dy, "gist_ln_relu_bwd_f32: null pointer");
    GIST_REQUIRE(!use_lynorm || rstd, "gist_ln_relu_bwd_f32: rstd is NULL with use_lynorm");
    GIST_REQUIRE(ldg >= d && ldy >= d && lddy >= d, "gist_ln_relu_bwd_f32: leading dimension < d");
    GIST_REQUIRE(n_rows < (1LL << 31) && d < (1LL << 31), "gist_ln_relu_bwd_f32: size >= 2^31");
    const bool v4 = d % 4 == 0 && ldg % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 &&
                    aligned16(d_out) && aligned16(yhat) && aligned16(dy);
    const bool wide = d > 1024;
    const unsigned grid = (unsigned)(wide ? n_rows : ceil_div(n_rows, 4));
#define L(TPR, V)                                                                               \
    hipLaunchKernelGGL((ln_relu_bwd_kernel<TPR, V>), dim3(grid), dim3(256), 0, st, d_out, ldg,   \
                       yhat, ldy, rstd, dy, lddy, (int)n_rows, (int)d, use_lynorm, relu, rowmax)
    if (wide) { if (v4) L(256, 4); else L(256, 1); }
    else { if (v4) L(64, 4); else L(64, 1); }
#undef L
    return launch_status("gist_ln_relu_bwd_f32");
}
}  // namespace gist

extern "C" int gist_ln_relu_bwd_f32(const float *d_out, int64_t ldg, const float *yhat,
                                    int64_t ldy, const float *rstd, float *dy, int64_t lddy,
                                    int64_t n_rows, int64_t d, int use_lynorm, int relu,
                                    gist_stream_t stream) {
    return gist::ln_relu_bwd_ex(d_out, ldg, yhat, ldy, rstd, dy, lddy, n_rows, d, use_lynorm, relu,
                                nullptr, gist::as_stream(stream));
}

extern "C" int64_t gist_row_chunks16(int64_t n_rows) { return n_rows <= 0 ? 0 : gist::ceil_div(n_rows, 16); }

namespace gist {
int colsum_rows16(const float *g, int64_t ldg, int64_t n_rows, int64_t d, float *partials, bool interleaved,
                  hipStream_t st) {
    if (n_rows <= 0 || d <= 0) return GIST_OK;
    const dim3 grid((unsigned)ceil_div(d, 256), (unsigned)ceil_div(n_rows, 16));
    if (interleaved)
        hipLaunchKernelGGL(colsum_rows16_kernel<true>, grid, dim3(256), 0, st, g, ldg, (int)n_rows, (int)d, partials);
    else
        hipLaunchKernelGGL(colsum_rows16_kernel<false>, grid, dim3(256), 0, st, g, ldg, (int)n_rows, (int)d, partials);
    return launch_status("gist_colsum_rows16");
}

// ln_relu_bwd_ex + col_partials [ceil(n_rows / 16)][d]: one kernel when d <= 1024 and the rows are
// 16-byte aligned, else the plain backward followed by the chunk sums of dy (same format, same order)
static int ln_relu_bwd_colsum_impl(const float *d_out, int64_t ldg, const float *yhat, int64_t ldy, const float *rstd,
                                   float *dy, int64_t lddy, int64_t n_rows, int64_t d, int use_lynorm, int relu,
                                   float *col_partials, const ClassDwArgs *dw, hipStream_t st) {
    GIST_REQUIRE(col_partials != nullptr, "gist_ln_relu_bwd_colsum_f32: null col_partials");
    const bool v4 = d % 4 == 0 && ldg % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 && aligned16(d_out) &&
                    aligned16(yhat) && aligned16(dy) && aligned16(col_partials);
    if (!(v4 && d <= 1024 && n_rows > 0 && d > 0 && d_out && yhat && dy && (!use_lynorm || rstd) &&
          ldg >= d && ldy >= d && lddy >= d && n_rows < (1LL << 31) - 16)) {
        if (dw != nullptr) {      // (no shared grid for this shape: the slabs as their own launch)
            hipLaunchKernelGGL(class_dw_only_kernel, dim3((unsigned)(dw->k / 64), (unsigned)ceil_div(dw->n_rows, kDwRows)),
                               dim3(256), 0, st, *dw);
            const int rc0 = launch_status("gist_ln_relu_bwd_colsum_class_dw_f32");
            if (rc0 != GIST_OK) return rc0;
        }
        const int rc = ln_relu_bwd_ex(d_out, ldg, yhat, ldy, rstd, dy, lddy, n_rows, d, use_lynorm, relu,
                                      nullptr, st);
        if (rc != GIST_OK) return rc;
        return colsum_rows16(dy, lddy, n_rows, d, col_partials, true, st);
    }
    const unsigned grid = (unsigned)ceil_div(n_rows, kLnCsRows);
    const int U = (int)ceil_div(d, 256);
    if (dw != nullptr) {
        const int dw_gx = dw->k / 64, n_dw = dw_gx * (int)ceil_div(dw->n_rows, kDwRows);
#define L(UU)                                                                                                        \
    hipLaunchKernelGGL((ln_relu_bwd_cs_dw_kernel<UU>), dim3(grid + (unsigned)n_dw), dim3(256), 0, st, d_out, ldg, yhat, \
                       ldy, rstd, dy, lddy, (int)n_rows, (int)d, use_lynorm, relu, col_partials, *dw, n_dw, dw_gx)
        switch (U) {
            case 1: L(1); break;
            case 2: L(2); break;
            case 3: L(3); break;
            default: L(4); break;
        }
#undef L
        return launch_status("gist_ln_relu_bwd_colsum_class_dw_f32");
    }
#define L(UU)                                                                                         \
    hipLaunchKernelGGL((ln_relu_bwd_cs_kernel<UU>), dim3(grid), dim3(256), 0, st, d_out, ldg, yhat, ldy, \
                       rstd, dy, lddy, (int)n_rows, (int)d, use_lynorm, relu, col_partials)
    switch (U) {
        case 1: L(1); break;
        case 2: L(2); break;
        case 3: L(3); break;
        default: L(4); break;
    }
#undef L
    return launch_status("gist_ln_relu_bwd_colsum_f32");
}

int ln_relu_bwd_colsum(const float *d_out, int64_t ldg, const float *yhat, int64_t ldy, const float *rstd,
                       float *dy, int64_t lddy, int64_t n_rows, int64_t d, int use_lynorm, int relu,
                       float *col_partials, hipStream_t st) {
    return ln_relu_bwd_colsum_impl(d_out, ldg, yhat, ldy, rstd, dy, lddy, n_rows, d, use_lynorm, relu, col_partials,
                                   nullptr, st);
}

// ln_relu_bwd_colsum with the class layer's weight-gradient slabs formed in the same launch (class_dw_body.h)
int ln_relu_bwd_colsum_class_dw(const float *d_out, int64_t ldg, const float *yhat, int64_t ldy, const float *rstd,
                                float *dy, int64_t lddy, int64_t n_rows, int64_t d, int use_lynorm, int relu,
                                float *col_partials, const ClassDwArgs &dw, hipStream_t st) {
    return ln_relu_bwd_colsum_impl(d_out, ldg, yhat, ldy, rstd, dy, lddy, n_rows, d, use_lynorm, relu, col_partials,
                                   &dw, st);
}
}  // namespace gist

extern "C" int gist_ln_relu_bwd_colsum_f32(const float *d_out, int64_t ldg, const float *yhat, int64_t ldy,
                                           const float *rstd, float *dy, int64_t lddy, int64_t n_rows,
                                           int64_t d, int use_lynorm, int relu, float *col_partials,
                                           gist_stream_t stream) {
    return gist::ln_relu_bwd_colsum(d_out, ldg, yhat, ldy, rstd, dy, lddy, n_rows, d, use_lynorm, relu,
                                    col_partials, gist::as_stream(stream));
}

extern "C" int gist_ln_relu_bwd_colsum_class_dw_f32(const float *d_out, int64_t ldg, const float *yhat, int64_t ldy,
                                                    const float *rstd, float *dy, int64_t lddy, int64_t n_rows,
                                                    int64_t d, int use_lynorm, int relu, float *col_partials,
                                                    const float *d_logits, int64_t ld_dlogits, const float *z, int64_t ldz,
                                                    float *slabs, int64_t slab_bytes, int32_t *n_slabs,
                                                    int64_t n_rows_cls, int64_t n_classes, int64_t k,
                                                    gist_stream_t stream) {
    GIST_REQUIRE(n_slabs != nullptr, "gist_ln_relu_bwd_colsum_class_dw_f32: null n_slabs");
    gist::ClassDwArgs w{};
    int32_t ns = 0;
    const int rc = gist::class_dw_args("gist_ln_relu_bwd_colsum_class_dw_f32", d_logits, ld_dlogits, z, ldz, slabs, slab_bytes,
                                       n_rows_cls, n_classes, k, &w, &ns);
    if (rc != GIST_OK) return rc;
    *n_slabs = ns;
    return gist::ln_relu_bwd_colsum_class_dw(d_out, ldg, yhat, ldy, rstd, dy, lddy, n_rows, d, use_lynorm, relu,
                                             col_partials, w, gist::as_stream(stream));
}

extern "C" int gist_colsum_chunks_f32(const float *partials, int64_t chunks, int64_t d, float *out,
                                      gist_stream_t stream) {
    GIST_REQUIRE(chunks >= 0 && d >= 0, "gist_colsum_chunks_f32: negative size");
    if (d == 0) return GIST_OK;
    GIST_REQUIRE(out && (partials || chunks == 0), "gist_colsum_chunks_f32: null pointer");
    hipLaunchKernelGGL(gist::colsum_chunks_kernel, dim3((unsigned)gist::ceil_div(d, 64)), dim3(256), 0,
                       gist::as_stream(stream), partials, (int)chunks, (int)d, out);
    return gist::launch_status("gist_colsum_chunks_f32");
}

extern "C" int gist_dropout_f32(float *z, int64_t ldz, int64_t n_rows, int64_t d, float p,
                                uint64_t seed, uint64_t offset, gist_stream_t stream) {
    GIST_REQUIRE(n_rows >= 0 && d >= 0, "gist_dropout_f32: negative size");
    GIST_REQUIRE(p >= 0.f && p < 1.f, "gist_dropout_f32: p must be in [0,1)");
    if (n_rows == 0 || d == 0 || p == 0.f) return GIST_OK;
    GIST_REQUIRE(z && ldz >= d, "gist_dropout_f32: bad buffer");
    const int64_t total = n_rows * d;
    if (d % 4 == 0 && ldz % 4 == 0 && aligned16(z) && (offset & 1) == 0) {
        const int64_t quads = total / 4;
        const unsigned grid = (unsigned)(ceil_div(quads, 256) < 16384 ? ceil_div(quads, 256) : 16384);
        hipLaunchKernelGGL(dropout_vec4_kernel, dim3(grid), dim3(256), 0, as_stream(stream), z, ldz,
                           n_rows, d, p, 1.0f / (1.0f - p), seed, offset);
        return launch_status("gist_dropout_f32");
    }
    const unsigned grid = (unsigned)(ceil_div(total, 256) < 8192 ? ceil_div(total, 256) : 8192);
    hipLaunchKernelGGL(dropout_kernel, dim3(grid), dim3(256), 0, as_stream(stream), z, ldz, n_rows,
                       d, p, 1.0f / (1.0f - p), seed, offset);
    return launch_status("gist_dropout_f32");
}

namespace gist {
// + dy_col_partials (NULL or [ceil(m / 16)][k]): column sums of g per 16-row chunk, rows in order
int gemm_nn_dropout_ex(const char *name, const float *g, int64_t ldg, const float *w, int64_t ldw, float *z,
                       int64_t ldz, int64_t m, int64_t n, int64_t k, float p, uint64_t seed, uint64_t offset,
                       void *workspace, int64_t workspace_bytes, float *dy_col_partials, hipStream_t st) {
    if (!(p >= 0.f && p < 1.f)) { set_error("%s: p must be in [0,1)", name); return GIST_EINVAL; }
    if (m < 0 || n < 0 || k < 0) { set_error("%s: negative size", name); return GIST_EINVAL; }
    const bool narrow = k >= 1 && k <= 64 && m > 0 && n > 0 && g && w && z && n % 4 == 0 &&
                        ldw % 4 == 0 && ldz % 4 == 0 && aligned16(w) && aligned16(z) &&
                        (offset & 1) == 0 && ldg >= k && ldw >= n && ldz >= n &&
                        m < (1LL << 31) && n < (1LL << 31);
    if (!narrow) {      // any other shape: the projection kernel, then the mask in place
        int rc = gist_gemm_nn_f32(g, ldg, w, ldw, z, ldz, m, n, k, workspace, workspace_bytes, st);
        if (rc == GIST_OK && p != 0.f) rc = gist_dropout_f32(z, ldz, m, n, p, seed, offset, st);
        if (rc == GIST_OK && dy_col_partials != nullptr) rc = colsum_rows16(g, ldg, m, k, dy_col_partials, false, st);
        return rc;
    }
    hipLaunchKernelGGL(narrow_nn_drop_kernel, dim3((unsigned)ceil_div(n, 1024), (unsigned)ceil_div(m, 16)),
                       dim3(256), 0, st, g, ldg, w, ldw, z, ldz, (int)m, (int)n, (int)k, p,
                       p > 0.f ? 1.0f / (1.0f - p) : 1.0f, seed, offset, dy_col_partials);
    return launch_status(name);
}
}  // namespace gist

extern "C" int gist_gemm_nn_dropout_f32(const float *g, int64_t ldg, const float *w, int64_t ldw,
                                        float *z, int64_t ldz, int64_t m, int64_t n, int64_t k,
                                        float p, uint64_t seed, uint64_t offset, void *workspace,
                                        int64_t workspace_bytes, gist_stream_t stream) {
    return gist::gemm_nn_dropout_ex("gist_gemm_nn_dropout_f32", g, ldg, w, ldw, z, ldz, m, n, k, p, seed,
                                    offset, workspace, workspace_bytes, nullptr, gist::as_stream(stream));
}

extern "C" int gist_gemm_nn_dropout_colsum_f32(const float *g, int64_t ldg, const float *w, int64_t ldw,
                                               float *z, int64_t ldz, int64_t m, int64_t n, int64_t k,
                                               float p, uint64_t seed, uint64_t offset, void *workspace,
                                               int64_t workspace_bytes, float *g_col_partials,
                                               gist_stream_t stream) {
    GIST_REQUIRE(g_col_partials != nullptr, "gist_gemm_nn_dropout_colsum_f32: null g_col_partials");
    return gist::gemm_nn_dropout_ex("gist_gemm_nn_dropout_colsum_f32", g, ldg, w, ldw, z, ldz, m, n, k, p,
                                    seed, offset, workspace, workspace_bytes, g_col_partials,
                                    gist::as_stream(stream));
}

extern "C" int64_t gist_colsum_partials(int64_t n_rows) {
    return n_rows <= 0 ? 0 : ceil_div(n_rows, kColsumRows);
}

namespace gist {
// + pmax ([chunks, d] scratch) and outmax ([d]): max |g| per column (both NULL = sums only)
int colsum_ex(const float *g, int64_t ldg, int64_t n_rows, int64_t d, float *partials, float *out,
              float *pmax, float *outmax, hipStream_t st) {
    GIST_REQUIRE(n_rows >= 0 && d >= 0, "gist_colsum_f32: negative size");
    if (d == 0) return GIST_OK;
    GIST_REQUIRE(out, "gist_colsum_f32: null output");
    GIST_REQUIRE((pmax == nullptr) == (outmax == nullptr), "gist_colsum_f32: pmax/outmax mismatch");
    const int chunks = (int)gist_colsum_partials(n_rows);
    if (chunks > 0) {
        GIST_REQUIRE(g && partials && ldg >= d, "gist_colsum_f32: bad buffer");
        GIST_REQUIRE(chunks <= 65535, "gist_colsum_f32: too many rows");
        const dim3 grid((unsigned)ceil_div(d, 64), (unsigned)chunks);
        if (pmax)
            hipLaunchKernelGGL(colsum_stage1_kernel<true>, grid, dim3(256), 0, st, g, ldg,
                               (int)n_rows, (int)d, partials, pmax);
        else
            hipLaunchKernelGGL(colsum_stage1_kernel<false>, grid, dim3(256), 0, st, g, ldg,
                               (int)n_rows, (int)d, partials, pmax);
    }
    hipLaunchKernelGGL(colsum_stage2_kernel, dim3((unsigned)ceil_div(d, 64)), dim3(64), 0, st,
                       partials, chunks, (int)d, out, pmax, outmax);
    return launch_status("gist_colsum_f32");
}
}  // namespace gist

namespace gist {
int colsum_finish(const float *partials, int64_t chunks, int64_t d, float *out, hipStream_t st) {
    if (d <= 0) return GIST_OK;
    hipLaunchKernelGGL(colsum_stage2_kernel, dim3((unsigned)ceil_div(d, 64)), dim3(64), 0, st, partials,
                       (int)chunks, (int)d, out, nullptr, nullptr);
    return launch_status("gist_colsum_f32");
}
}  // namespace gist

extern "C" int gist_colsum_f32(const float *g, int64_t ldg, int64_t n_rows, int64_t d,
                               float *partials, float *out, gist_stream_t stream) {
    return gist::colsum_ex(g, ldg, n_rows, d, partials, out, nullptr, nullptr,
                           gist::as_stream(stream));
}

namespace gist {
// slabs (NULL: logits hold the values) / loss (NULL: the caller reduces row_loss later, e.g. inside
// gist_adam_segments_f32)
int softmax_xent_ex(const char *name, float *logits, int64_t ldl, const float *slabs, int64_t slab_stride,
                    int n_slabs, const float *bias, const int32_t *labels, const uint8_t *mask, int64_t count,
                    float *row_loss, float *loss, float *d_logits, int64_t ldg, int64_t n_rows,
                    int64_t n_classes, hipStream_t st) {
    if (!(n_rows > 0 && n_classes > 0)) { set_error("%s: empty input", name); return GIST_EINVAL; }
    if (!(logits && labels && d_logits && row_loss)) { set_error("%s: null pointer", name); return GIST_EINVAL; }
    if (!(ldl >= n_classes && ldg >= n_classes)) { set_error("%s: leading dimension < n_classes", name); return GIST_EINVAL; }
    if (count <= 0) { set_error("%s: count must be > 0", name); return GIST_EINVAL; }
    if (!(n_rows < (1LL << 31) && ldg < (1LL << 31))) { set_error("%s: size >= 2^31", name); return GIST_EINVAL; }
    if (slabs != nullptr && n_slabs < 1) { set_error("%s: n_slabs < 1", name); return GIST_EINVAL; }
    const float inv = 1.0f / (float)count;
    hipLaunchKernelGGL(xent_grad_kernel, dim3((unsigned)ceil_div(n_rows, 4)), dim3(256), 0, st, logits, ldl,
                       labels, mask, inv, d_logits, ldg, row_loss, (int)n_rows, (int)n_classes, slabs,
                       slab_stride, n_slabs, bias);
    if (loss != nullptr)
        hipLaunchKernelGGL(xent_loss_kernel, dim3(1), dim3(256), 0, st, row_loss, (int)n_rows, inv, loss);
    return launch_status(name);
}
}  // namespace gist

extern "C" int gist_softmax_xent_f32(const float *logits, int64_t ldl, const int32_t *labels,
                                     const uint8_t *mask, int64_t count, float *row_loss,
                                     float *loss, float *d_logits, int64_t ldg, int64_t n_rows,
                                     int64_t n_classes, gist_stream_t stream) {
    GIST_REQUIRE(loss != nullptr, "gist_softmax_xent_f32: null pointer");
    return gist::softmax_xent_ex("gist_softmax_xent_f32", const_cast<float *>(logits), ldl, nullptr, 0, 0,
                                 nullptr, labels, mask, count, row_loss, loss, d_logits, ldg, n_rows,
                                 n_classes, gist::as_stream(stream));
}

extern "C" int gist_softmax_xent_slabs_f32(float *logits, int64_t ldl, const float *slabs,
                                           int64_t slab_stride, int64_t n_slabs, const float *bias,
                                           const int32_t *labels, const uint8_t *mask, int64_t count,
                                           float *row_loss, float *loss, float *d_logits, int64_t ldg,
                                           int64_t n_rows, int64_t n_classes, gist_stream_t stream) {
    GIST_REQUIRE(n_slabs >= 0 && n_slabs < 4096, "gist_softmax_xent_slabs_f32: bad n_slabs");
    return gist::softmax_xent_ex("gist_softmax_xent_slabs_f32", logits, ldl, n_slabs > 0 ? slabs : nullptr,
                                 slab_stride, (int)n_slabs, bias, labels, mask, count, row_loss, loss,
                                 d_logits, ldg, n_rows, n_classes, gist::as_stream(stream));
}

extern "C" int gist_adam_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                             int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int64_t step, gist_stream_t stream) {
    GIST_REQUIRE(n >= 0, "gist_adam_f32: n < 0");
    if (n == 0) return GIST_OK;
    GIST_REQUIRE(param && grad && exp_avg && exp_avg_sq, "gist_adam_f32: null pointer");
    GIST_REQUIRE(step >= 1, "gist_adam_f32: step is 1-based");
    // bias corrections in double on the host, like torch's python scalar math
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    const unsigned grid = (unsigned)(ceil_div(n, 256) < 16384 ? ceil_div(n, 256) : 16384);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, as_stream(stream), param, grad,
                       exp_avg, exp_avg_sq, n, beta1, beta2, eps, weight_decay, step_size,
                       inv_bc2_sqrt);
    return launch_status("gist_adam_f32");
}

namespace gist {
int adam_segments_args(const char *name, float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int64_t step,
                       const gist_grad_segment *segments, int64_t n_segments, const float *row_loss,
                       int64_t n_loss_rows, int64_t loss_count, float *loss, AdamArgs *out) {
    GIST_REQUIRE(n >= 0, "%s: n < 0", name);
    GIST_REQUIRE(n_segments >= 0 && n_segments <= kAdamMaxSegs, "%s: too many segments", name);
    GIST_REQUIRE(n_segments == 0 || segments != nullptr, "%s: null segments", name);
    GIST_REQUIRE(row_loss == nullptr || (loss != nullptr && loss_count > 0 && n_loss_rows >= 0 &&
                                         n_loss_rows < (1LL << 31)),
                 "%s: bad loss arguments", name);
    GIST_REQUIRE(n == 0 || (param && grad && exp_avg && exp_avg_sq), "%s: null pointer", name);
    GIST_REQUIRE(step >= 1, "%s: step is 1-based", name);
    AdamArgs A{};
    AdamSegs &sg = A.segs;
    for (int64_t i = 0; i < n_segments; ++i) {
        const gist_grad_segment &q = segments[i];
        if (q.src == nullptr || q.end <= q.begin) continue;
        GIST_REQUIRE(q.begin >= 0 && q.end <= n && q.n_src >= 1 && q.stride >= q.end - q.begin,
                     "%s: bad segment %d", name, (int)i);
        for (int j = 0; j < sg.n; ++j)
            GIST_REQUIRE(q.end <= sg.begin[j] || q.begin >= sg.end[j], "%s: overlapping segments", name);
        sg.begin[sg.n] = q.begin; sg.end[sg.n] = q.end; sg.stride[sg.n] = q.stride;
        sg.n_src[sg.n] = q.n_src; sg.src[sg.n] = q.src;
        // many sources, few elements (a bias gradient in row chunks): dedicated blocks of 64 elements
        sg.ded_first[sg.n] = -1;
        sg.fin_first[sg.n] = -1;
        if (q.n_src > 16 && q.end - q.begin <= 65536) {
            sg.ded_first[sg.n] = sg.n_ded;
            sg.n_ded += (int)ceil_div(q.end - q.begin, 64);
        } else {      // (gist_grad_segments_finish_f32: the arena chunks this segment touches)
            GIST_REQUIRE((q.end - 1) / 1024 - q.begin / 1024 + 1 + sg.n_fin < (1LL << 30), "%s: segment too large", name);
            sg.fin_first[sg.n] = sg.n_fin;
            sg.n_fin += (int)((q.end - 1) / 1024 - q.begin / 1024 + 1);
        }
        ++sg.n;
    }
    // bias corrections in double on the host, like torch's python scalar math
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    A.p = param; A.g = grad; A.m = exp_avg; A.v = exp_avg_sq; A.n = n;
    A.beta1 = beta1; A.beta2 = beta2; A.eps = eps; A.wd = weight_decay;
    A.step_size = (float)((double)lr / bc1);
    A.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    A.row_nll = row_loss; A.n_loss_rows = (int)n_loss_rows;
    A.inv_count = row_loss ? 1.0f / (float)loss_count : 0.f;
    A.loss = loss;
    GIST_REQUIRE(adam_arena_blocks(n) + sg.n_ded < (1LL << 31) - 2, "%s: arena too large", name);
    *out = A;
    return GIST_OK;
}
}  // namespace gist

extern "C" int gist_adam_segments_f32(float *param, float *grad, float *exp_avg, float *exp_avg_sq,
                                      int64_t n, float lr, float beta1, float beta2, float eps,
                                      float weight_decay, int64_t step, const gist_grad_segment *segments,
                                      int64_t n_segments, const float *row_loss, int64_t n_loss_rows,
                                      int64_t loss_count, float *loss, gist_stream_t stream) {
    gist::AdamArgs A;
    const int rc = gist::adam_segments_args("gist_adam_segments_f32", param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2,
                                            eps, weight_decay, step, segments, n_segments, row_loss, n_loss_rows,
                                            loss_count, loss, &A);
    if (rc != GIST_OK) return rc;
    if (n == 0 && row_loss == nullptr) return GIST_OK;
    const int64_t blocks = gist::adam_arena_blocks(n) + A.segs.n_ded + (row_loss ? 1 : 0);
    hipLaunchKernelGGL(gist::adam_segments_kernel, dim3((unsigned)blocks), dim3(256), 0, gist::as_stream(stream), A);
    return gist::launch_status("gist_adam_segments_f32");
}

extern "C" int gist_grad_segments_finish_f32(float *grad, int64_t n, const gist_grad_segment *segments,
                                             int64_t n_segments, gist_stream_t stream) {
    gist::AdamArgs A;
    const int rc = gist::adam_segments_args("gist_grad_segments_finish_f32", grad, grad, grad, grad, n, 0.f, 0.9f, 0.999f,
                                            1e-8f, 0.f, 1, segments, n_segments, nullptr, 0, 0, nullptr, &A);
    if (rc != GIST_OK) return rc;
    const int64_t blocks = (int64_t)A.segs.n_fin + A.segs.n_ded;
    if (blocks == 0) return GIST_OK;
    hipLaunchKernelGGL(gist::grad_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, gist::as_stream(stream), A);
    return gist::launch_status("gist_grad_segments_finish_f32");
}

namespace gist {
// loss = mean of row_loss[0..n): the reduction gist_adam_segments_f32 performs, as a launch of its own
int loss_finish(const float *row_loss, int64_t n_rows, int64_t count, float *loss, hipStream_t st) {
    hipLaunchKernelGGL(xent_loss_kernel, dim3(1), dim3(256), 0, st, row_loss, (int)n_rows, 1.0f / (float)count, loss);
    return launch_status("gist_sage_step");
}
}  // namespace gist

extern "C" int gist_argmax_correct_i32(const float *logits, int64_t ldl, const int32_t *labels,
                                       const uint8_t *mask, int32_t *correct, int64_t n_rows,
                                       int64_t n_classes, gist_stream_t stream) {
    GIST_REQUIRE(n_rows >= 0 && n_classes > 0, "gist_argmax_correct_i32: bad size");
    if (n_rows == 0) return GIST_OK;
    GIST_REQUIRE(logits && labels && correct && ldl >= n_classes,
                 "gist_argmax_correct_i32: bad buffer");
    GIST_REQUIRE(n_rows < (1LL << 31), "gist_argmax_correct_i32: size >= 2^31");
    hipLaunchKernelGGL(argmax_correct_kernel, dim3((unsigned)ceil_div(n_rows, 4)), dim3(256), 0,
                       as_stream(stream), logits, ldl, labels, mask, correct, (int)n_rows,
                       (int)n_classes);
    return launch_status("gist_argmax_correct_i32");
}
