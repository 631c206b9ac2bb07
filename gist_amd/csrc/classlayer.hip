// The class layer of the training step as two kernels instead of four launches (round 4).
//
// The last SAGE layer projects [h | ah] (K = 2 * hidden columns) onto C <= 48 classes
// (cluster_gcn/modules.py:299-308: no LayerNorm, no activation), the loss is the mean CE of the rows
// (cluster_gcn_ist_distrib.py:413), and the backward needs dZ = dropout-mask(dlogits . W) and
// dW = dlogits^T . Z.  As separate launches that was: a skinny projection GEMM (8-9 us), the CE kernel
// (5 us), the narrow dZ kernel (11 us) and a skinny transposed GEMM for dW (13-19 us on the generic
// kernel: 41 output rows) -- ~40 us of a 320-us step for 0.5 GFLOP.  Every one of them is tiny next to the
// machine; what they cost is their fixed part.
//
// class_layer_kernel: one workgroup per 16 batch rows (the chunk size of the bias gradient's partial
// sums).  (A) logits = Z W^T + b on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains): the four waves split K,
// operands come straight from memory as 16-byte loads (lane l holds row l % 16, k = k0 + 4 (l / 16) .. +3
// of both operands: the SAME k permutation on both sides, so the four MFMA steps of a block consume the
// four components), partial tiles are added in wave order through LDS; (B) softmax / CE / dlogits per
// row by 16 lanes, the chunk's column sums of dlogits (the bias gradient's partial) in row order;
// (C) dZ = dlogits . W with dlogits as the A operand from LDS and W's rows from memory, the dropout mask
// of the layer's input (gist_dropout_f32's generator, same element index) applied to what is stored.
// class_dw_kernel: dW = dlogits^T . Z as split-K slabs over 128-row chunks (the optimiser sums them:
// gist_adam_segments_f32), one 16-column tile per wave, all C classes.
#include "common.h"
#include "class_dw_body.h"

namespace gist {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kRows = 16;        // rows per workgroup
constexpr int kCpad = 48;        // classes padded to three 16-wide MFMA tiles

struct ClassArgs {
    const float *z; int64_t ldz;
    const float *w; int64_t ldw;
    const float *bias;
    const int32_t *labels;
    float inv_count;
    float *logits; int64_t ldl;
    float *dlog; int64_t ldg;
    float *row_nll;
    float *dz; int64_t lddz;          // NULL: forward + loss only
    float p, scale; uint64_t sm, offset;
    float *col_partials;              // [ceil(n / 16)][C] or NULL
    int n_rows, n_classes, k;
};

__global__ __launch_bounds__(256) void class_layer_kernel(ClassArgs a) {
    __shared__ float part[4][kRows][kCpad];     // per-wave partial logits
    __shared__ float dls[kRows][kCpad + 4];     // dlogits of the chunk (zero beyond C / beyond n_rows)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int r0 = blockIdx.x * kRows;
    const int C = a.n_classes, K = a.k;
    // ---- (A) partial logits of this wave's quarter of K ------------------------------------------
    // Blocks of 16 k; the operands of FOUR blocks are loaded together and the next four are in flight while the
    // current four feed the matrix cores (a block's 12 MFMAs take 384 cycles, a load from L2 ~1000: without the
    // run-ahead the loop is one memory latency per block -- measured 24 us for the kernel, against 9 like this).
    {
        const int row = min(r0 + r, a.n_rows - 1);
        const float *zp = a.z + (int64_t)row * a.ldz + 4 * q;
        const float *wp[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) wp[t] = a.w + (int64_t)min(16 * t + r, C - 1) * a.ldw + 4 * q;
        f32x4 acc[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int kq = K / 4, kb = wave * kq;
        constexpr int U = 4;
        float4 za0[U], wb0[U][3], za1[U], wb1[U][3];
        auto load = [&](float4 (&za)[U], float4 (&wb)[U][3], int k0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                za[u] = *reinterpret_cast<const float4 *>(zp + k0 + 16 * u);
#pragma unroll
                for (int t = 0; t < 3; ++t) wb[u][t] = *reinterpret_cast<const float4 *>(wp[t] + k0 + 16 * u);
            }
        };
        auto compute = [&](const float4 (&za)[U], const float4 (&wb)[U][3]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float zs[4] = {za[u].x, za[u].y, za[u].z, za[u].w};
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const float ws = s_ == 0 ? wb[u][t].x : s_ == 1 ? wb[u][t].y : s_ == 2 ? wb[u][t].z : wb[u][t].w;
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(zs[s_], ws, acc[t], 0, 0, 0);
                    }
                }
            }
        };
        const int n_chunks = kq / (16 * U);                 // (k % 256 == 0: whole chunks; else the tail loop below)
        if (n_chunks > 0) load(za0, wb0, kb);
        for (int c = 0; c < n_chunks; c += 2) {
            if (c + 1 < n_chunks) load(za1, wb1, kb + (c + 1) * 16 * U);
            compute(za0, wb0);
            if (c + 2 < n_chunks) load(za0, wb0, kb + (c + 2) * 16 * U);
            if (c + 1 < n_chunks) compute(za1, wb1);
        }
        for (int k0 = kb + n_chunks * 16 * U; k0 < kb + kq; k0 += 16) {      // at most three blocks
            const float4 za = *reinterpret_cast<const float4 *>(zp + k0);
            float4 wb[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) wb[t] = *reinterpret_cast<const float4 *>(wp[t] + k0);
            const float zs[4] = {za.x, za.y, za.z, za.w};
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const float ws = s_ == 0 ? wb[t].x : s_ == 1 ? wb[t].y : s_ == 2 ? wb[t].z : wb[t].w;
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(zs[s_], ws, acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[wave][4 * q + i][16 * t + r] = acc[t][i];
    }
    __syncthreads();
    // ---- (B) logits, softmax, CE, dlogits: 16 lanes per row, classes c, c + 16, c + 32 ---------------
    {
        const int row = tid >> 4, cl = tid & 15;
        const bool live = r0 + row < a.n_rows;
        float lg[3];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int c = cl + 16 * t;
            float v = ((part[0][row][c] + part[1][row][c]) + part[2][row][c]) + part[3][row][c];
            if (c < C) {
                v += a.bias ? a.bias[c] : 0.f;
                mx = fmaxf(mx, v);
            } else {
                v = -INFINITY;
            }
            lg[t] = v;
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        float se = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) se += (cl + 16 * t < C) ? expf(lg[t] - mx) : 0.f;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) se += __shfl_xor(se, off);
        const int lab = live ? a.labels[r0 + row] : -1;
        const float inv = 1.f / se;
        float at_label = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int c = cl + 16 * t;
            float g = 0.f;
            if (live && c < C) {
                g = (expf(lg[t] - mx) * inv - (c == lab ? 1.f : 0.f)) * a.inv_count;
                a.logits[(int64_t)(r0 + row) * a.ldl + c] = lg[t];
                if (c == lab) at_label = lg[t];
            }
            dls[row][c] = g;
            if (live && c < a.ldg) a.dlog[(int64_t)(r0 + row) * a.ldg + c] = g;
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) at_label += __shfl_xor(at_label, off);
        if (live && cl == 0) a.row_nll[r0 + row] = -((at_label - mx) - logf(se));
    }
    __syncthreads();
    if (a.col_partials != nullptr && tid < C) {      // bias gradient of the chunk: rows in order
        float s = 0.f;
#pragma unroll
        for (int rr = 0; rr < kRows; ++rr) s += dls[rr][tid];
        a.col_partials[(int64_t)blockIdx.x * C + tid] = s;
    }
    if (a.dz == nullptr) return;
    // ---- (C) dZ[16, K] = dlogits[16, C] . W[C, K], masked: wave w takes column tiles w, w + 4, ... in pairs ----
    // (always the 12 k steps of 48 padded classes: dlogits are zero beyond C and the W row index is clamped, so the
    // steps need no guards -- a run-time step count made every step a branch with its own s_waitcnt vmcnt(0), one
    // memory latency per step: 10 us for this phase instead of 4)
    float af[12];
#pragma unroll
    for (int s = 0; s < 12; ++s) af[s] = dls[r][4 * s + q];
    const float *wrow[12];
#pragma unroll
    for (int s = 0; s < 12; ++s) wrow[s] = a.w + (int64_t)min(4 * s + q, C - 1) * a.ldw + r;
    const int n_tiles = K / 16;
    // tile pairs wave, wave + 4, ...: the 24 W loads of the next pair are in flight under the current pair's MFMAs
    float b0[2][12], b1[2][12];
    auto loadb = [&](float (&p0)[12], float (&p1)[12], int nt) {
        const int n0 = nt * 16;
#pragma unroll
        for (int s_ = 0; s_ < 12; ++s_) { p0[s_] = wrow[s_][n0]; p1[s_] = wrow[s_][n0 + 16]; }
    };
    auto tile = [&](const float (&p0)[12], const float (&p1)[12], int nt) {
        const int n0 = nt * 16;
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s_ = 0; s_ < 12; ++s_) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s_], p0[s_], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s_], p1[s_], d1, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 4 * q + i;
            if (row >= a.n_rows) continue;
            float v0 = d0[i], v1 = d1[i];
            if (a.p > 0.f) {
                const uint64_t idx = a.offset + (uint64_t)row * (uint64_t)K + (uint64_t)(n0 + r);
                v0 *= drop_keep(idx, a.sm, a.p, a.scale);
                v1 *= drop_keep(idx + 16, a.sm, a.p, a.scale);
            }
            float *o = a.dz + (int64_t)row * a.lddz + n0 + r;
            o[0] = v0;
            o[16] = v1;
        }
    };
    int nt = 2 * wave;                                   // (K % 32 == 0: every pair is complete)
    if (nt < n_tiles) loadb(b0[0], b1[0], nt);
    for (; nt < n_tiles; nt += 16) {
        if (nt + 8 < n_tiles) loadb(b0[1], b1[1], nt + 8);
        tile(b0[0], b1[0], nt);
        if (nt + 16 < n_tiles) loadb(b0[0], b1[0], nt + 16);
        if (nt + 8 < n_tiles) tile(b0[1], b1[1], nt + 8);
    }
}

// grid (K / 64, ceil(n / 128)): class_dw_body.h
__global__ __launch_bounds__(256) void class_dw_kernel(ClassDwArgs a) { class_dw_block(a, (int)blockIdx.x, (int)blockIdx.y); }

}  // namespace

// Measured (rocprofv3, Reddit-like batch, C = 41): the fused kernel 19-20 us at K = 1024 against 8.2 + 5.0 + 11.4 for
// projection + CE + narrow dZ, 12 against 5.4 + 5.0 + 10.8 at K = 512; at K = 2048 it is 31-37 against 29 (one workgroup
// per 16 rows fills half the chip, and the three launches do not have that limit): taken up to K = 1024.
bool class_layer_takes(int64_t n_rows, int64_t n_classes, int64_t k, int64_t ldz, int64_t ldw, const float *z,
                       const float *w) {
    return n_rows > 0 && n_rows < (1LL << 31) - 64 && n_classes >= 1 && n_classes <= kCpad && k >= 64 &&
           k % 64 == 0 && k <= 1024 && ldz % 4 == 0 && ldw % 4 == 0 && ldz >= k && ldw >= k && aligned16(z) &&
           aligned16(w);
}

int64_t class_dw_slabs(int64_t n_rows) { return ceil_div(n_rows, kDwRows); }

}  // namespace gist

using namespace gist;

extern "C" int gist_class_layer_takes(int64_t n_rows, int64_t n_classes, int64_t k, int64_t ldz, int64_t ldw,
                                      const float *z, const float *w) {
    return class_layer_takes(n_rows, n_classes, k, ldz, ldw, z, w) ? 1 : 0;
}

extern "C" int64_t gist_class_dw_slab_bytes(int64_t n_rows, int64_t n_classes, int64_t k) {
    if (n_rows <= 0 || n_classes <= 0 || k <= 0) return 0;
    return class_dw_slabs(n_rows) * n_classes * k * 4;
}

extern "C" int gist_class_layer_f32(const float *z, int64_t ldz, const float *w, int64_t ldw, const float *bias,
                                    const int32_t *labels, int64_t count, float *logits, int64_t ldl,
                                    float *d_logits, int64_t ldg, float *row_loss, float *dz, int64_t lddz,
                                    float p, uint64_t seed, uint64_t offset, float *dlogits_col_partials,
                                    int64_t n_rows, int64_t n_classes, int64_t k, gist_stream_t stream) {
    GIST_REQUIRE(z && w && labels && logits && d_logits && row_loss, "gist_class_layer_f32: null pointer");
    GIST_REQUIRE(class_layer_takes(n_rows, n_classes, k, ldz, ldw, z, w),
                 "gist_class_layer_f32: shape not taken (see gist_class_layer_takes)");
    GIST_REQUIRE(count > 0 && ldl >= n_classes && ldg >= n_classes && ldg <= 64, "gist_class_layer_f32: bad leading dimension / count");
    GIST_REQUIRE(p >= 0.f && p < 1.f, "gist_class_layer_f32: p must be in [0,1)");
    GIST_REQUIRE(dz == nullptr || lddz >= k, "gist_class_layer_f32: bad lddz");
    ClassArgs a{};
    a.z = z; a.ldz = ldz; a.w = w; a.ldw = ldw; a.bias = bias; a.labels = labels;
    a.inv_count = 1.0f / (float)count;
    a.logits = logits; a.ldl = ldl; a.dlog = d_logits; a.ldg = ldg; a.row_nll = row_loss;
    a.dz = dz; a.lddz = lddz; a.p = p; a.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.f;
    a.sm = seed * 0x9E3779B97F4A7C15ULL; a.offset = offset;
    a.col_partials = dlogits_col_partials;
    a.n_rows = (int)n_rows; a.n_classes = (int)n_classes; a.k = (int)k;
    hipLaunchKernelGGL(class_layer_kernel, dim3((unsigned)ceil_div(n_rows, kRows)), dim3(256), 0,
                       as_stream(stream), a);
    return launch_status("gist_class_layer_f32");
}

namespace gist {
int class_dw_args(const char *name, const float *d_logits, int64_t ldg, const float *z, int64_t ldz, float *slabs,
                  int64_t slab_bytes, int64_t n_rows, int64_t n_classes, int64_t k, ClassDwArgs *out, int32_t *n_slabs) {
    GIST_REQUIRE(d_logits && z && slabs && n_slabs, "%s: null pointer", name);
    GIST_REQUIRE(n_rows > 0 && n_rows < (1LL << 31) - 256 && n_classes >= 1 && n_classes <= kCpad && k >= 64 &&
                     k % 64 == 0 && ldg >= n_classes && ldz >= k,
                 "%s: bad shape", name);
    const int64_t ns = class_dw_slabs(n_rows);
    GIST_REQUIRE(slab_bytes >= ns * n_classes * k * 4, "%s: slab buffer too small", name);
    ClassDwArgs a{};
    a.dlog = d_logits; a.ldg = ldg; a.z = z; a.ldz = ldz; a.slabs = slabs;
    a.n_rows = (int)n_rows; a.n_classes = (int)n_classes; a.k = (int)k;
    *out = a;
    *n_slabs = (int32_t)ns;
    return GIST_OK;
}
}  // namespace gist

extern "C" int gist_class_dw_slabs_f32(const float *d_logits, int64_t ldg, const float *z, int64_t ldz,
                                       float *slabs, int64_t slab_bytes, int32_t *n_slabs, int64_t n_rows,
                                       int64_t n_classes, int64_t k, gist_stream_t stream) {
    GIST_REQUIRE(n_slabs != nullptr, "gist_class_dw_slabs_f32: null n_slabs");
    ClassDwArgs a{};
    int32_t ns = 0;
    const int rc = class_dw_args("gist_class_dw_slabs_f32", d_logits, ldg, z, ldz, slabs, slab_bytes, n_rows, n_classes, k, &a, &ns);
    if (rc != GIST_OK) return rc;
    hipLaunchKernelGGL(class_dw_kernel, dim3((unsigned)(k / 64), (unsigned)ns), dim3(256), 0, as_stream(stream), a);
    *n_slabs = ns;
    return launch_status("gist_class_dw_slabs_f32");
}
