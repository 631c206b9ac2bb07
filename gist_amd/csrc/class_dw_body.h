// The class layer's weight gradient as split-K slabs over 128-row chunks (classlayer.hip: gist_class_dw_slabs_f32) as a
// device function, so that its workgroups can share a grid with another kernel's (rowops.hip).
#pragma once
#include "common.h"

namespace gist {

using dw_f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kDwRows = 128;     // rows per slab

struct ClassDwArgs {
    const float *dlog; int64_t ldg;
    const float *z; int64_t ldz;
    float *slabs;                     // [ceil(n / 128)][C][K]
    int n_rows, n_classes, k;
};

// workgroup (bx, by) of a (K / 64, ceil(n / 128)) grid of 256-thread workgroups: wave w owns columns [64 bx + 16 w, +16)
// of slab by.  No LDS, no barrier: class_dw_kernel runs it as its own grid, ln_relu_bwd_cs_dw_kernel (rowops.hip) in
// one grid with the LayerNorm backward of the layer below.
__device__ __forceinline__ void class_dw_block(const ClassDwArgs &a, const int bx, const int by) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int C = a.n_classes, K = a.k;
    const int n0 = bx * 64 + wave * 16;
    const int rb = by * kDwRows;
    const int re = min(rb + kDwRows, a.n_rows);
    int cls[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) cls[t] = min(16 * t + r, C - 1);
    dw_f32x4 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = dw_f32x4{0.f, 0.f, 0.f, 0.f};
    // eight k steps (32 rows) per chunk, the next chunk's 32 loads in flight under the current chunk's 24 MFMAs
    constexpr int U = 8;
    float bz[2][U], av[2][U][3];
    auto load = [&](float (&bzz)[U], float (&avv)[U][3], int k0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = k0 + 4 * u + q;
            const bool live = rr < re;
            const int rc = live ? rr : re - 1;
            const float zv = a.z[(int64_t)rc * a.ldz + n0 + r];
            bzz[u] = live ? zv : 0.f;
#pragma unroll
            for (int t = 0; t < 3; ++t) avv[u][t] = a.dlog[(int64_t)rc * a.ldg + cls[t]];
        }
    };
    auto compute = [&](const float (&bzz)[U], const float (&avv)[U][3]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(avv[u][t], bzz[u], acc[t], 0, 0, 0);
    };
    const int n_chunks = (re - rb + 4 * U - 1) / (4 * U);
    if (n_chunks > 0) load(bz[0], av[0], rb);
    for (int c = 0; c < n_chunks; c += 2) {
        if (c + 1 < n_chunks) load(bz[1], av[1], rb + (c + 1) * 4 * U);
        compute(bz[0], av[0]);
        if (c + 2 < n_chunks) load(bz[0], av[0], rb + (c + 2) * 4 * U);
        if (c + 1 < n_chunks) compute(bz[1], av[1]);
    }
    float *slab = a.slabs + (int64_t)by * C * K;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 16 * t + 4 * q + i;
            if (c < C) slab[(int64_t)c * K + n0 + r] = acc[t][i];
        }
}


// host side (classlayer.hip): checks the shapes, fills the arguments; *n_slabs = ceil(n_rows / 128)
int class_dw_args(const char *name, const float *d_logits, int64_t ldg, const float *z, int64_t ldz, float *slabs,
                  int64_t slab_bytes, int64_t n_rows, int64_t n_classes, int64_t k, ClassDwArgs *out, int32_t *n_slabs);

}  // namespace gist
