// Shared helpers for libgist_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/gist_hip.h"

namespace gist {

constexpr int kWave = 64;

void set_error(const char *fmt, ...);

static inline hipStream_t as_stream(gist_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Kernel launches issued by this process through the library (gist_launch_count): a measurement aid -- the launch count
// of a step, next to the cost of as many EMPTY launches (gist_empty_launches), is the floor of a launch-bound step.
extern std::atomic<uint64_t> g_launches;

// Checks the launch itself (not completion): calls stay asynchronous.
static inline int launch_status(const char *what) {
    g_launches.fetch_add(1, std::memory_order_relaxed);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return GIST_ELAUNCH;
    }
    return GIST_OK;
}

#define GIST_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            gist::set_error(__VA_ARGS__);  \
            return GIST_EINVAL;            \
        }                                  \
    } while (0)

// HIP-event timer of the native step driver (step.hip).  While gist_sage_step runs with a timer
// armed, tl_timer points at it so that a kernel below an entry point (the split GEMM's main
// kernel) can be bracketed on its own: slot = timer_begin(...); launch; timer_end(slot).
extern thread_local gist_timer *tl_timer;
int64_t timer_begin(gist_timer *t, int kind, int64_t m, int64_t n, int64_t k, hipStream_t s);
void timer_end(gist_timer *t, int64_t slot, hipStream_t s);

// dropout's counter-based generator (rowops.hip; also applied inside the split pre-pass)
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// gemm_h3.hip: building blocks of the split projection path for a caller that manages the
// split operands itself (the step driver).  Split operand = [rows][h3_kpad(k)] 32-bit words
// + one inverse scale (power of two) per row.
int64_t h3_kpad(int64_t k);
bool h3_eligible(int64_t m, int64_t n, int64_t k);        // a call that splits its own operands
bool h3_eligible_kept(int64_t m, int64_t n, int64_t k);   // operands split once, kept by the step
struct H3Dual {                       // one read of src[rows, cols] -> up to two split operands
    const float *src; int64_t ld; int64_t rows, cols;
    float p; uint64_t seed, offset;   // dropout applied on the fly (p = 0: none), gist_dropout_f32's stream
    int fixed_shift;                  // uniform scale 2^fixed_shift ...
    const unsigned *amax;             // ... or, if not NULL, from the tensor's max |x| (float bits)
    const float *rowmax, *colmax;     // if not NULL: per-row / per-column maxima instead
    uint32_t *dst_r; float *inv_r;    // [rows][kpad(cols)]: k = columns of src   (NULL: skip)
    uint32_t *dst_t; float *inv_t;    // [cols][kpad(rows)]: k = rows of src      (NULL: skip)
};
int h3_dual_split(const H3Dual &d, hipStream_t st);
int h3_split_rows(const float *src, int64_t ld, int64_t rows, int64_t k, uint32_t *dst, float *inv,
                  hipStream_t st);
int h3_absmax(const float *src, int64_t ld, int64_t rows, int64_t cols, unsigned *out, hipStream_t st);
int h3_gemm_presplit(const char *name, const uint32_t *sa, const float *inv_a, const uint32_t *sb,
                     const float *inv_b, const float *bias, float *c, int64_t ldc, int64_t m,
                     int64_t n, int64_t k, hipStream_t st);

// gemm_b3.hip: the bf16x3 split projection path (all 24 operand bits, no scales).  Split operand =
// [rows][b3_kpad(k)] elements of 6 bytes (three bf16 pieces per element, 16-byte chunks per 8 k).
int h3_mode();                        // 0 fp32 MFMA, 1 f16x3, 2 bf16x3 (gist_gemm_set_mode)
void h3_mode_override(int mode);      // >= 0: h3_mode() of THIS thread returns it (sizing queries); -1: off
int64_t b3_kpad(int64_t k);
bool b3_eligible(int64_t m, int64_t n, int64_t k);
bool b3_eligible_kept(int64_t m, int64_t n, int64_t k);
struct B3Dual {                       // one read of src[rows, cols] -> up to two split operands
    const float *src; int64_t ld; int64_t rows, cols;
    float p; uint64_t seed, offset;   // dropout applied on the fly (p = 0: none), gist_dropout_f32's stream
    uint16_t *dst_r;                  // [rows][kpad(cols)]: k = columns of src   (NULL: skip)
    uint16_t *dst_t;                  // [cols][kpad(rows)]: k = rows of src      (NULL: skip)
    float *col_partials;              // [ceil(rows / 64)][cols] column sums per 64-row chunk of src (NULL: none):
                                      // the first stage of gist_colsum_f32, taken from the tile the split reads anyway
    bool vec4;                        // set by b3_dual_split: 16-byte loads allowed
};
int b3_dual_split(const B3Dual &d, hipStream_t st);
int64_t b3_slab_bytes(int64_t m, int64_t n, int64_t k);   // fp32 slabs of a split-K call (0: one k slice)
int b3_gemm_presplit(const char *name, const uint16_t *sa, const uint16_t *sb, const float *bias, float *c,
                     int64_t ldc, int64_t m, int64_t n, int64_t k, float *slabs, int64_t slab_bytes,
                     hipStream_t st, int *deferred = nullptr);
// gemm.hip: gist_gemm_{nt,nn,tn}_f32 (layout 0, 1, 2) whose split-K partial sums stay as dense slabs
// [*n_slabs][m][n] at `slabs` for the consumer to sum in slab order (+ bias); *n_slabs = 1: c is final
int gemm_slabs(int layout, const float *a, int64_t lda, const float *b, int64_t ldb, const float *bias, float *c,
               int64_t ldc, int64_t m, int64_t n, int64_t k, void *slabs, int64_t slab_bytes, int *n_slabs,
               hipStream_t st);
int64_t gemm_f32_slab_bytes(int64_t m, int64_t n, int64_t k, bool tn = false);      // tn: a weight gradient (A and B k-major)
// gemm.hip: dz = dy . w (NN) and dW = dy^T . z (TN, slabs) in one launch of the fp32 kernel's tiles
bool gemm_dual_takes(int64_t m, int64_t n1, int64_t k1, int64_t lddy, int64_t ldw, int64_t ldz, int64_t lddz,
                     const float *dy, const float *w, const float *z, const float *dz);
int gemm_dual_nn_tn(const char *name, const float *dy, int64_t lddy, const float *w, int64_t ldw, float *dz,
                    int64_t lddz, const float *z, int64_t ldz, float *dw, int64_t lddw, int64_t m, int64_t n1,
                    int64_t k1, void *slabs, int64_t slab_bytes, int *n_slabs, hipStream_t st);
// gemm.hip: c[m, n] (ldc) = sum of `splits` dense slabs [m][n] + bias
int splitk_reduce(const char *name, const float *slabs, int64_t slab, int splits, const float *bias, float *c,
                  int64_t ldc, int64_t m, int64_t n, hipStream_t st);

// spmm.hip: dropout folded into an aggregation (gist_spmm_csr_drop_f32); element (row, c) of y has mask
// index y_base + row * ld + c, of x src_base + row * ld + c (sm = seed * golden ratio)
struct SpmmDrop {
    int mode;                 // 0 none, 1 forward (mask what is stored), 2 backward (mask x and the old y)
    float p, scale;
    uint64_t sm, y_base, src_base;
    int64_t ld;
};
// LayerNorm + ReLU backward of the layer BELOW in the store of a reverse aggregation (mode 2, d <= 256: a wave holds a
// whole row): what would be stored is d_out of that layer's output; dy = rstd . (g - mean(g) - yhat . mean(g . yhat)),
// g = d_out . [yhat > 0], goes to `dy` (may be `yhat` itself), nothing to y; col_partials[unit][0..d) = the column sums of
// the dy rows one workgroup stored (spmm_lnb_units(n_row_blocks) rows: the bias gradient's partial sums)
struct SpmmLnBwd {
    const float *yhat; int64_t ldy;
    const float *rstd;            // NULL: no LayerNorm (dy = g)
    float *dy; int64_t lddy;
    float *col_partials;
    int relu;
};
int spmm_drop(const int32_t *rowptr, const int32_t *col, const float *x, int64_t ldx, float *y, int64_t ldy,
              int64_t n_rows, int64_t d, const float *out_scale, const float *src_scale, int accumulate,
              const int32_t *row_blocks, int64_t n_row_blocks, const SpmmDrop &dr, hipStream_t st,
              const void *prepared = nullptr, const SpmmLnBwd *ln = nullptr);
// can a mode-2 call carry SpmmLnBwd, and how many partial rows does it write?
bool spmm_lnb_takes(int64_t d, int64_t ldx, int64_t ldy, const float *x, const float *y, const int32_t *row_blocks,
                    const void *prepared);
int64_t spmm_lnb_units(int64_t n_row_blocks);
bool spmm_drop_takes(int mode, int64_t d, int64_t ldx, int64_t ldy, const float *x, const float *y,
                     const int32_t *row_blocks);
#ifdef __HIPCC__
// Sum over the 64 lanes of a wave, the same bits in every lane.  Within a row of 16 lanes on the DPP path of the vector unit
// (lane ^ 1, lane ^ 2 as quad permutes; the other quad of the half row and the other half row as mirrors -- every lane of a
// quad / half row holds the same partial sum by then), the four rows through the scalar unit: 4 + 4 + 3 instructions, none of
// them on the LDS pipe (the __shfl_xor butterfly is six dependent ds_bpermute round trips).
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0xB1>(v);       // quad_perm [1, 0, 3, 2]
    v += dpp_move<0x4E>(v);       // quad_perm [2, 3, 0, 1]
    v += dpp_move<0x141>(v);      // row_half_mirror
    v += dpp_move<0x140>(v);      // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}
// gist_dropout_f32's keep/scale factor of element idx, of VEC consecutive elements, of an aligned quad
__device__ __forceinline__ float drop_keep(uint64_t idx, uint64_t sm, float p, float scale) {
    const uint64_t h = splitmix64((idx >> 1) + sm);
    const uint32_t w = (idx & 1) ? (uint32_t)(h >> 32) : (uint32_t)h;
    return ((float)(w >> 8) * (1.0f / 16777216.0f) >= p) ? scale : 0.f;
}
template <int VEC>
__device__ __forceinline__ void drop_vec(float (&v)[VEC], uint64_t idx0, const SpmmDrop &dr) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] *= drop_keep(idx0 + k, dr.sm, dr.p, dr.scale);
}
__device__ __forceinline__ void drop_f4(float4 &v, uint64_t idx0, const SpmmDrop &dr) {
    const float inv = 1.0f / 16777216.0f;
    if ((idx0 & 1) == 0) {                 // the usual case: two hashes cover the quad
        const uint64_t pair = idx0 >> 1;
        const uint64_t h0 = splitmix64(pair + dr.sm), h1 = splitmix64(pair + 1 + dr.sm);
        v.x *= ((float)((uint32_t)h0 >> 8) * inv >= dr.p) ? dr.scale : 0.f;
        v.y *= ((float)((uint32_t)(h0 >> 32) >> 8) * inv >= dr.p) ? dr.scale : 0.f;
        v.z *= ((float)((uint32_t)h1 >> 8) * inv >= dr.p) ? dr.scale : 0.f;
        v.w *= ((float)((uint32_t)(h1 >> 32) >> 8) * inv >= dr.p) ? dr.scale : 0.f;
    } else {
        v.x *= drop_keep(idx0, dr.sm, dr.p, dr.scale); v.y *= drop_keep(idx0 + 1, dr.sm, dr.p, dr.scale);
        v.z *= drop_keep(idx0 + 2, dr.sm, dr.p, dr.scale); v.w *= drop_keep(idx0 + 3, dr.sm, dr.p, dr.scale);
    }
}

#endif
// spmm_mfma.hip: the block-dense aggregation kernel behind gist_spmm_csr_blocked_f32 / _prepared_f32
// (prepared = NULL: every workgroup builds its block's counts itself)
int launch_spmm_mfma(const int32_t *rowptr, const int32_t *col, const float *x, int64_t ldx, float *y,
                     int64_t ldy, int64_t n_rows, int64_t d, const float *out_scale, const float *src_scale,
                     int accumulate, const int32_t *row_blocks, int64_t n_row_blocks, const void *prepared,
                     hipStream_t st, const SpmmDrop *dr = nullptr, bool pairs = true);
int64_t spmm_blocks_bytes(int64_t n_blocks);
// Pairs of sibling blocks (spmm_mfma.hip): false while a caller that KNOWS its batch has none is issuing launches
// (gist_sage_step with plan->sibling_parts == 0): the prepare kernel then looks for none and no pairs launch follows
// an aggregation.  Everybody else: true.
extern thread_local bool tl_spmm_pairs;
// spmm_dense32.hip: the block-dense aggregation on the fp32 matrix cores, operands from memory (prepared blocks only)
bool spmm_dense32_takes(int64_t d, int64_t ldx, int64_t ldy);
int launch_spmm_dense32(const int32_t *rowptr, const int32_t *col, const float *x, int64_t ldx, float *y,
                        int64_t ldy, int64_t n_rows, int64_t d, const float *out_scale, const float *src_scale,
                        int accumulate, const int32_t *row_blocks, int64_t n_row_blocks, const void *prepared,
                        hipStream_t st, const SpmmDrop *dr = nullptr);
bool spmm_prepared_takes(int64_t d, int64_t ldx, int64_t ldy, const float *x, const float *y);   // spmm.hip
int launch_spmm_blocks_prepare(const int32_t *rowptr, const int32_t *col, const int32_t *rowptr2,
                               const int32_t *col2, int64_t n_rows, const int32_t *row_blocks,
                               int64_t n_row_blocks, void *prepared, void *prepared2, hipStream_t st, bool pairs = true);

// rowops.hip: the C-ABI kernels with the extra outputs the split projection path consumes
int ln_relu_bwd_ex(const float *d_out, int64_t ldg, const float *yhat, int64_t ldy, const float *rstd,
                   float *dy, int64_t lddy, int64_t n_rows, int64_t d, int use_lynorm, int relu,
                   float *rowmax, hipStream_t st);
int colsum_ex(const float *g, int64_t ldg, int64_t n_rows, int64_t d, float *partials, float *out,
              float *pmax, float *outmax, hipStream_t st);
// second stage alone: out[d] = sum over `chunks` rows of partials[chunks][d], fixed order
int colsum_finish(const float *partials, int64_t chunks, int64_t d, float *out, hipStream_t st);
// rowops.hip: the fused producers of "column sums per 16-row chunk" and their consumers
int ln_relu_bwd_colsum(const float *d_out, int64_t ldg, const float *yhat, int64_t ldy, const float *rstd,
                       float *dy, int64_t lddy, int64_t n_rows, int64_t d, int use_lynorm, int relu,
                       float *col_partials, hipStream_t st);
struct ClassDwArgs;
int ln_relu_bwd_colsum_class_dw(const float *d_out, int64_t ldg, const float *yhat, int64_t ldy, const float *rstd,
                                float *dy, int64_t lddy, int64_t n_rows, int64_t d, int use_lynorm, int relu,
                                float *col_partials, const ClassDwArgs &dw, hipStream_t st);
int gemm_nn_dropout_ex(const char *name, const float *g, int64_t ldg, const float *w, int64_t ldw, float *z,
                       int64_t ldz, int64_t m, int64_t n, int64_t k, float p, uint64_t seed, uint64_t offset,
                       void *workspace, int64_t workspace_bytes, float *dy_col_partials, hipStream_t st);
int softmax_xent_ex(const char *name, float *logits, int64_t ldl, const float *slabs, int64_t slab_stride,
                    int n_slabs, const float *bias, const int32_t *labels, const uint8_t *mask, int64_t count,
                    float *row_loss, float *loss, float *d_logits, int64_t ldg, int64_t n_rows,
                    int64_t n_classes, hipStream_t st);
int colsum_rows16(const float *g, int64_t ldg, int64_t n_rows, int64_t d, float *partials, bool interleaved,
                  hipStream_t st);
// loss[0] = sum(row_loss[0..n_rows)) / count, the reduction gist_adam_segments_f32 performs, as its own launch
int loss_finish(const float *row_loss, int64_t n_rows, int64_t count, float *loss, hipStream_t st);

// Tuning hooks (gist_tuning_set, include/gist_hip.h): explicit process-wide overrides of the
// launchers' own choices, for sweeps and for tests that must reach both variants of a kernel.
// 0 = the launcher decides.  The library never reads the environment on a launch path.
double tune(int knob);

// Per-device one-time setup (hipFuncSetAttribute applies to the current device only).  Usage:
//   static DeviceOnce once;  int dev;  if (once.needed(&dev)) { ...set...; once.done(dev); }
// Racing threads may both run the setup, which is idempotent.
struct DeviceOnce {
    std::atomic<uint64_t> mask{0};
    bool needed(int *dev) {
        if (hipGetDevice(dev) != hipSuccess || *dev < 0 || *dev > 63) { *dev = -1; return true; }
        return ((mask.load(std::memory_order_acquire) >> *dev) & 1ULL) == 0;
    }
    void done(int dev) {
        if (dev >= 0) mask.fetch_or(1ULL << dev, std::memory_order_release);
    }
};

// XCDs of the device the library runs on (gfx950: 8); blockIdx -> tile maps deal work round-robin
// to XCDs the way the hardware dispatches consecutive workgroups.
constexpr int kXcds = 8;

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool aligned8(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace gist
