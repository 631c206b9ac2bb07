// Adam over a flat arena with DEFERRED gradient reductions: the device code of gist_adam_segments_f32, in a header
// because two kernels run it -- adam_segments_kernel (rowops.hip) and adam_extract_kernel (subgraph.hip: the
// optimiser's workgroups and the NEXT batch's extraction in one grid).
//
// Inside a listed segment [begin, end) the gradient of element i is not grads[i] but the sum of n_src arrays
// src[s * stride + (i - begin)] -- the split-K slabs of a weight-gradient projection, or the per-row-chunk column
// sums of a bias gradient -- formed here (and written back to grads) instead of by a reduce launch of its own.
// Work is cut into VIRTUAL blocks of 256 threads that never synchronise with each other, so that a workgroup of any
// multiple of 256 threads can run several of them:
//   * arena blocks own 1024 consecutive elements; the segments that touch a block's range are found once per
//     block (uniform), so outside them the loop is adam_kernel's.  Slab segments (few sources) are summed
//     inline in source order, four loads in flight;
//   * chunk-sum segments (`ded`: many sources, few elements -- a bias gradient in 16-row chunks) are skipped by
//     the arena blocks; DEDICATED blocks take 64 elements each, four lanes per element over the sources q, q + 4,
//     ... (eight loads in flight), the four partial sums added in q order (the order gist_colsum_chunks_f32
//     uses too) -- a thread that walked 128 chunks for each of its 4 elements set the kernel's duration (76 us at
//     h = 512 against 6 for the plain kernel).  The four lanes of an element sit in one wave (lane = 16 q + e) and
//     meet through shuffles: no LDS, no barrier;
//   * one more block (a whole workgroup: it uses a barrier) reduces the step's loss when row_nll != NULL.
#pragma once
#include "common.h"

namespace gist {

constexpr int kAdamMaxSegs = 2 * GIST_MAX_LAYERS;
struct AdamSegs {
    int n;
    int n_src[kAdamMaxSegs];
    int ded_first[kAdamMaxSegs];      // first dedicated block of the segment (-1: inline)
    int n_ded;                         // dedicated blocks in total
    // gist_grad_segments_finish_f32 (the sums WITHOUT the optimiser update): an inline segment's own arena chunks,
    // fin_first[s] = its first block in that launch's grid (-1: dedicated), n_fin = such blocks in total
    int fin_first[kAdamMaxSegs];
    int n_fin;
    int64_t begin[kAdamMaxSegs], end[kAdamMaxSegs], stride[kAdamMaxSegs];
    const float *src[kAdamMaxSegs];
};
struct AdamArgs {
    float *p, *g, *m, *v;
    int64_t n;
    float beta1, beta2, eps, wd, step_size, inv_bc2_sqrt;
    AdamSegs segs;
    const float *row_nll; int n_loss_rows; float inv_count; float *loss;
};
// virtual blocks of a launch: arena chunks, then dedicated blocks (the loss block is a workgroup of its own)
__host__ __device__ inline int64_t adam_arena_blocks(int64_t n) { return (n + 1023) / 1024; }

__device__ __forceinline__ float adam_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// loss = inv_count * sum_i row_nll[i] by threads 0 .. 255 of the workgroup, fixed-order tree (deterministic); every
// thread of the workgroup must call it (one barrier)
__device__ __forceinline__ void loss_reduce_256(const float *__restrict__ row_nll, int n_rows,
                                                float inv_count, float *__restrict__ loss, float *red) {
    float s = 0.f;
    if (threadIdx.x < 256)
        for (int i = threadIdx.x; i < n_rows; i += 256) s += row_nll[i];
    s = adam_wave_sum(s);
    if (threadIdx.x < 256 && (threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (((red[0] + red[1]) + red[2]) + red[3]) * inv_count;
}

// sum over sources q, q + 4, q + 8, ... in ascending order (q = 0..3)
__device__ __forceinline__ float chunk_partial4(const float *__restrict__ src, int64_t stride, int n_src, int q) {
    float acc = 0.f;
    int k = q;
    for (; k + 28 < n_src; k += 32) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[(int64_t)(k + 4 * u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += t[u];
    }
    for (; k < n_src; k += 4) acc += src[(int64_t)k * stride];
    return acc;
}

__device__ __forceinline__ void adam_update(float *__restrict__ p, float *__restrict__ m, float *__restrict__ v,
                                            int64_t i, float gv, float beta1, float beta2, float eps, float wd,
                                            float step_size, float inv_bc2_sqrt) {
    const float pv = p[i];
    if (wd != 0.f) gv = fmaf(wd, pv, gv);
    const float mv = m[i] + (1.f - beta1) * (gv - m[i]);        // lerp_ like torch
    const float vv = beta2 * v[i] + (1.f - beta2) * gv * gv;
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) * inv_bc2_sqrt + eps;
    p[i] = pv - step_size * (mv / denom);
}

// virtual block vb (arena chunks first, then the dedicated blocks), thread vt of its 256; no barrier inside.
// kUpdate = false (gist_grad_segments_finish_f32): the segment sums are formed and written to the gradient arena
// in the same order, the optimiser update is left out; only_seg >= 0 restricts an arena chunk to that segment
template <bool kUpdate = true>
__device__ __forceinline__ void adam_virtual_block(const AdamArgs &A, int64_t vb, int vt, int only_seg = -1) {
    const AdamSegs &segs = A.segs;
    const int64_t n_chunks = adam_arena_blocks(A.n);
    if (vb >= n_chunks) {
        const int idx = (int)(vb - n_chunks);
        if (idx >= segs.n_ded) return;
        int sg = -1;                                       // uniform: the segment this block serves
        for (int s = 0; s < segs.n; ++s)                   // (the last one that starts at or before idx)
            if (segs.ded_first[s] >= 0 && segs.ded_first[s] <= idx &&
                (sg < 0 || segs.ded_first[s] > segs.ded_first[sg]))
                sg = s;
        if (sg < 0) return;
        // wave w of the block: elements 16 w .. 16 w + 15 of its 64, lane = 16 q + e
        const int lane = vt & 63, w = vt >> 6;
        const int e = 16 * w + (lane & 15), q = lane >> 4;
        const int64_t i = segs.begin[sg] + (int64_t)(idx - segs.ded_first[sg]) * 64 + e;
        const bool live = i < segs.end[sg];
        float acc = 0.f;
        if (live) acc = chunk_partial4(segs.src[sg] + (i - segs.begin[sg]), segs.stride[sg], segs.n_src[sg], q);
        const float a1 = __shfl(acc, (lane & 15) + 16), a2 = __shfl(acc, (lane & 15) + 32), a3 = __shfl(acc, (lane & 15) + 48);
        if (q == 0 && live) {
            const float gv = ((acc + a1) + a2) + a3;
            A.g[i] = gv;
            if (kUpdate) adam_update(A.p, A.m, A.v, i, gv, A.beta1, A.beta2, A.eps, A.wd, A.step_size, A.inv_bc2_sqrt);
        }
        return;
    }
    const int64_t lo = vb * 1024;
    const int64_t hi = lo + 1024 < A.n ? lo + 1024 : A.n;
    unsigned touch = 0;                                   // uniform: segments intersecting [lo, hi)
    for (int s = 0; s < segs.n; ++s)
        if (segs.begin[s] < hi && segs.end[s] > lo && (only_seg < 0 || s == only_seg)) touch |= 1u << s;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = lo + vt + 256 * u;
        if (i >= hi) break;
        float gv;
        int sg = -1;
        for (unsigned t = touch; t; t &= t - 1) {
            const int s = __builtin_ctz(t);
            if (i >= segs.begin[s] && i < segs.end[s]) sg = s;
        }
        if (sg >= 0) {
            if (segs.ded_first[sg] >= 0) continue;        // a dedicated block owns this element
            const float *src = segs.src[sg] + (i - segs.begin[sg]);
            const int64_t stride = segs.stride[sg];
            const int ns = segs.n_src[sg];
            float acc = 0.f;
            int k = 0;
            for (; k + 4 <= ns; k += 4) {                 // source order, four loads in flight
                const float t0 = src[(int64_t)k * stride], t1 = src[(int64_t)(k + 1) * stride];
                const float t2 = src[(int64_t)(k + 2) * stride], t3 = src[(int64_t)(k + 3) * stride];
                acc += t0; acc += t1; acc += t2; acc += t3;
            }
            for (; k < ns; ++k) acc += src[(int64_t)k * stride];
            gv = acc;
            A.g[i] = acc;
        } else {
            if (!kUpdate) continue;
            gv = A.g[i];
        }
        if (kUpdate) adam_update(A.p, A.m, A.v, i, gv, A.beta1, A.beta2, A.eps, A.wd, A.step_size, A.inv_bc2_sqrt);
    }
}

// what gist_adam_segments_f32 checks and derives from its arguments (rowops.hip), shared with the fused launch
int adam_segments_args(const char *name, float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int64_t step,
                       const gist_grad_segment *segments, int64_t n_segments, const float *row_loss,
                       int64_t n_loss_rows, int64_t loss_count, float *loss, AdamArgs *out);

}  // namespace gist
