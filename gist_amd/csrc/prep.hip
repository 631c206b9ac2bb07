// Data preparation on the device: sklearn's StandardScaler as the
// reference applies it before training (cluster_gcn/cluster_gcn_ist_distrib.py:492-499,
// cluster_gcn/cluster_gcn.py:37-44): mean and POPULATION variance of every feature column over
// the TRAIN rows, accumulated in float64 like sklearn; scale = sqrt(var), 1 where it is 0;
// every row of the full feature matrix becomes f32(f32(x - mean) / scale) (sklearn's in-place
// `X -= mean_; X /= scale_` on a float32 array with float64 statistics).
// HBM-bound: two passes over the train rows for the statistics (sum, then centred sum of squares:
// no E[x^2] - mean^2 cancellation), one pass over all rows for the transform.
#include "common.h"

namespace gist {

constexpr int kStatRows = 256;      // rows per partial block

// partial[blk][c] = sum over the block's rows of (x - shift[c])^(1 or 2); 64 columns x 4 row
// lanes per workgroup, fixed summation order -> deterministic
template <bool SQUARE>
__global__ __launch_bounds__(256) void col_stat_partial_kernel(
    const float *__restrict__ x, int64_t ld, const int32_t *__restrict__ rows, int64_t n_rows, int d,
    const double *__restrict__ shift, double *__restrict__ partial) {
    __shared__ double red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * kStatRows;
    const int64_t r1 = min(r0 + kStatRows, n_rows);
    double acc = 0.0;
    if (c < d) {
        const double sh = shift ? shift[c] : 0.0;
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            const int64_t row = rows ? (int64_t)rows[r] : r;
            const double v = (double)x[row * ld + c] - sh;
            acc += SQUARE ? v * v : v;
        }
    }
    red[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && c < d)
        partial[(int64_t)blockIdx.y * d + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) +
                                                (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// out[c] = (sum over blocks of partial[blk][c]) / count  (+ add[c]); fixed order
__global__ void col_stat_final_kernel(const double *__restrict__ partial, int64_t n_blocks, int d,
                                      double inv_count, const double *__restrict__ add,
                                      double *__restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d) return;
    double s = 0.0;
    for (int64_t b = 0; b < n_blocks; ++b) s += partial[b * d + c];
    out[c] = s * inv_count + (add ? add[c] : 0.0);
}

__global__ __launch_bounds__(256) void standardize_kernel(float *__restrict__ x, int64_t ld,
                                                          int64_t n_rows, int d,
                                                          const double *__restrict__ mean,
                                                          const double *__restrict__ var) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    if (c >= d) return;
    const double m = mean[c];
    const double v = var[c];
    const double s = v > 0.0 ? sqrt(v) : 1.0;              // sklearn: zero variance -> scale 1
    const int64_t r0 = (int64_t)blockIdx.y * 64;
    const int64_t r1 = min(r0 + 64, n_rows);
    for (int64_t r = r0 + (threadIdx.x >> 6); r < r1; r += 4) {
        float *p = x + r * ld + c;
        const float t = (float)((double)*p - m);
        *p = (float)((double)t / s);
    }
}

}  // namespace gist

extern "C" int64_t gist_standard_scaler_workspace_bytes(int64_t n_fit_rows, int64_t d) {
    if (n_fit_rows <= 0 || d <= 0) return 0;
    return gist::ceil_div(n_fit_rows, gist::kStatRows) * d * 8;
}

extern "C" int gist_standard_scaler_f32(float *x, int64_t ld, int64_t n_rows, int64_t d,
                                        const int32_t *fit_rows, int64_t n_fit_rows, double *mean,
                                        double *var, void *workspace, int64_t workspace_bytes,
                                        gist_stream_t stream) {
    using namespace gist;
    GIST_REQUIRE(n_rows >= 0 && d >= 0 && n_fit_rows >= 0, "gist_standard_scaler_f32: negative size");
    if (n_rows == 0 || d == 0) return GIST_OK;
    GIST_REQUIRE(x && mean && var, "gist_standard_scaler_f32: null pointer");
    GIST_REQUIRE(ld >= d, "gist_standard_scaler_f32: leading dimension < d");
    GIST_REQUIRE(n_fit_rows > 0, "gist_standard_scaler_f32: no rows to fit on");
    GIST_REQUIRE(d < (1LL << 31), "gist_standard_scaler_f32: d >= 2^31");
    GIST_REQUIRE(workspace && workspace_bytes >= gist_standard_scaler_workspace_bytes(n_fit_rows, d),
                 "gist_standard_scaler_f32: workspace too small");
    hipStream_t st = as_stream(stream);
    double *partial = static_cast<double *>(workspace);
    const int64_t nblk = ceil_div(n_fit_rows, kStatRows);
    GIST_REQUIRE(nblk <= 65535 * 16, "gist_standard_scaler_f32: too many fit rows");
    const dim3 grid((unsigned)ceil_div(d, 64), (unsigned)nblk);
    const int fin = (int)ceil_div(d, 256);
    hipLaunchKernelGGL((col_stat_partial_kernel<false>), grid, dim3(256), 0, st, x, ld, fit_rows,
                       n_fit_rows, (int)d, (const double *)nullptr, partial);
    hipLaunchKernelGGL(col_stat_final_kernel, dim3(fin), dim3(256), 0, st, partial, nblk, (int)d,
                       1.0 / (double)n_fit_rows, (const double *)nullptr, mean);
    hipLaunchKernelGGL((col_stat_partial_kernel<true>), grid, dim3(256), 0, st, x, ld, fit_rows,
                       n_fit_rows, (int)d, (const double *)mean, partial);
    hipLaunchKernelGGL(col_stat_final_kernel, dim3(fin), dim3(256), 0, st, partial, nblk, (int)d,
                       1.0 / (double)n_fit_rows, (const double *)nullptr, var);
    const dim3 g2((unsigned)ceil_div(d, 64), (unsigned)ceil_div(n_rows, 64));
    hipLaunchKernelGGL(standardize_kernel, g2, dim3(256), 0, st, x, ld, n_rows, (int)d, mean, var);
    return launch_status("gist_standard_scaler_f32");
}
