/* A plain-C consumer of the boundary that does DEVICE work (tests/test_cabi_device_gpu.py builds and runs it on the
 * GPU box): no Python, no torch in the process -- hipMalloc'ed buffers, the library's entry points on the NULL stream,
 * results against C loops.  A 300-node random multigraph (duplicate edges, self loops, one node without in-edges):
 *   gist_in_degree_norm_f32          vs 1 / in_degree, 0 for degree 0            (cluster_gcn/modules.py:239-243)
 *   gist_spmm_csr_f32, forward form  y = norm . sum over in-edges of x          (modules.py:223-226)
 *   gist_spmm_csr_f32, reversed + accumulate form: dX += A^T (norm . dAH)       (autograd of the same op)
 *   gist_gemm_nt_f32                 Y = A . W^T + bias                          (modules.py:233)
 * Returns 0 and prints "cabi device consumer ok". */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gist_hip.h"

#define N 300
#define D 40
#define NNZ 2400
#define M_OUT 24

static unsigned long long rng_state = 0x9E3779B97F4A7C15ULL;
static unsigned rnd(void) {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (unsigned)(rng_state >> 32);
}
static float frand(void) { return (float)(rnd() % 20001) / 10000.0f - 1.0f; }

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at line %d\n", (int)e_, __LINE__); return 100; } } while (0)
#define GIST_OK_(x) do { int r_ = (x); if (r_ != GIST_OK) { printf("gist error %d at line %d: %s\n", r_, __LINE__, gist_last_error()); return 101; } } while (0)

static void csr_from_edges(const int *row, const int *colv, int nnz, int n, int32_t *rowptr, int32_t *col) {
    int i, e;
    int *fill = (int *)calloc((size_t)n + 1, sizeof(int));
    memset(rowptr, 0, sizeof(int32_t) * (size_t)(n + 1));
    for (e = 0; e < nnz; ++e) rowptr[row[e] + 1]++;
    for (i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
    for (e = 0; e < nnz; ++e) { col[rowptr[row[e]] + fill[row[e]]] = colv[e]; fill[row[e]]++; }
    free(fill);
}

int main(void) {
    static int src[NNZ], dst[NNZ];
    static int32_t rowptr[N + 1], col[NNZ], t_rowptr[N + 1], t_col[NNZ];
    static float x[N * D], norm_ref[N], y_ref[N * D], dx_ref[N * D], dah[N * D], dx0[N * D];
    static float a[N * 2 * D], w[M_OUT * 2 * D], bias[M_OUT], yg_ref[N * M_OUT];
    static float got[N * 2 * D];
    int32_t *d_rowptr, *d_col, *d_t_rowptr, *d_t_col;
    float *d_x, *d_y, *d_norm, *d_dah, *d_dx, *d_a, *d_w, *d_bias, *d_yg;
    void *d_ws = NULL;
    int64_t ws_bytes;
    int e, i, j, k, n_dev = 0;
    double worst;

    if (gist_abi_version() != 16) return 1;
    HIP_OK(hipGetDeviceCount(&n_dev));
    if (n_dev < 1 || gist_device_count() < 1) { printf("no device\n"); return 2; }
    HIP_OK(hipSetDevice(0));

    for (e = 0; e < NNZ; ++e) {                 /* u -> v; node N-1 gets no in-edge; duplicates and self loops occur */
        src[e] = (int)(rnd() % N);
        dst[e] = (int)(rnd() % (N - 1));
        if (e % 97 == 0) src[e] = dst[e];       /* self loop */
        if (e % 53 == 1) { src[e] = src[e - 1]; dst[e] = dst[e - 1]; }   /* duplicate edge */
    }
    csr_from_edges(dst, src, NNZ, N, rowptr, col);          /* in-edge CSR: row = destination */
    csr_from_edges(src, dst, NNZ, N, t_rowptr, t_col);      /* reversed */
    for (i = 0; i < N * D; ++i) { x[i] = frand(); dah[i] = frand(); dx0[i] = frand(); }
    for (i = 0; i < N * 2 * D; ++i) a[i] = frand();
    for (i = 0; i < M_OUT * 2 * D; ++i) w[i] = frand() * 0.2f;
    for (i = 0; i < M_OUT; ++i) bias[i] = frand();

    /* C-loop references */
    for (i = 0; i < N; ++i) {
        int deg = rowptr[i + 1] - rowptr[i];
        norm_ref[i] = deg > 0 ? 1.0f / (float)deg : 0.0f;
        for (j = 0; j < D; ++j) {
            double s = 0.0;
            for (e = rowptr[i]; e < rowptr[i + 1]; ++e) s += x[col[e] * D + j];
            y_ref[i * D + j] = (float)(s * norm_ref[i]);
        }
    }
    for (i = 0; i < N; ++i)                     /* dx[u] = dx0[u] + sum over out-edges u -> v of norm[v] * dah[v] */
        for (j = 0; j < D; ++j) {
            double s = dx0[i * D + j];
            for (e = t_rowptr[i]; e < t_rowptr[i + 1]; ++e) s += (double)norm_ref[t_col[e]] * dah[t_col[e] * D + j];
            dx_ref[i * D + j] = (float)s;
        }
    for (i = 0; i < N; ++i)
        for (j = 0; j < M_OUT; ++j) {
            double s = bias[j];
            for (k = 0; k < 2 * D; ++k) s += (double)a[i * 2 * D + k] * w[j * 2 * D + k];
            yg_ref[i * M_OUT + j] = (float)s;
        }

    HIP_OK(hipMalloc((void **)&d_rowptr, sizeof rowptr)); HIP_OK(hipMalloc((void **)&d_col, sizeof col));
    HIP_OK(hipMalloc((void **)&d_t_rowptr, sizeof t_rowptr)); HIP_OK(hipMalloc((void **)&d_t_col, sizeof t_col));
    HIP_OK(hipMalloc((void **)&d_x, sizeof x)); HIP_OK(hipMalloc((void **)&d_y, sizeof x));
    HIP_OK(hipMalloc((void **)&d_norm, sizeof norm_ref)); HIP_OK(hipMalloc((void **)&d_dah, sizeof dah));
    HIP_OK(hipMalloc((void **)&d_dx, sizeof dx0)); HIP_OK(hipMalloc((void **)&d_a, sizeof a));
    HIP_OK(hipMalloc((void **)&d_w, sizeof w)); HIP_OK(hipMalloc((void **)&d_bias, sizeof bias));
    HIP_OK(hipMalloc((void **)&d_yg, sizeof yg_ref));
    HIP_OK(hipMemcpy(d_rowptr, rowptr, sizeof rowptr, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_col, col, sizeof col, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_t_rowptr, t_rowptr, sizeof t_rowptr, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_t_col, t_col, sizeof t_col, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_x, x, sizeof x, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_dah, dah, sizeof dah, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_dx, dx0, sizeof dx0, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_a, a, sizeof a, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_w, w, sizeof w, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_bias, bias, sizeof bias, hipMemcpyHostToDevice));
    ws_bytes = gist_gemm_workspace_bytes(N, M_OUT, 2 * D);
    if (ws_bytes > 0) HIP_OK(hipMalloc(&d_ws, (size_t)ws_bytes));

    /* 1. norm */
    GIST_OK_(gist_in_degree_norm_f32(d_rowptr, N, d_norm, NULL));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(got, d_norm, sizeof norm_ref, hipMemcpyDeviceToHost));
    for (i = 0; i < N; ++i) if (got[i] != norm_ref[i]) { printf("norm[%d] %g != %g\n", i, got[i], norm_ref[i]); return 10; }
    if (norm_ref[N - 1] != 0.0f) return 11;

    /* 2. forward aggregation, fused 1/deg */
    GIST_OK_(gist_spmm_csr_f32(d_rowptr, d_col, d_x, D, d_y, D, N, D, d_norm, NULL, 0, NULL));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(got, d_y, sizeof x, hipMemcpyDeviceToHost));
    worst = 0.0;
    for (i = 0; i < N * D; ++i) { double d_ = fabs((double)got[i] - y_ref[i]); if (d_ > worst) worst = d_; }
    if (worst > 1e-5) { printf("spmm forward: max error %g\n", worst); return 12; }

    /* 3. reversed graph, source scale, accumulate */
    GIST_OK_(gist_spmm_csr_f32(d_t_rowptr, d_t_col, d_dah, D, d_dx, D, N, D, NULL, d_norm, 1, NULL));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(got, d_dx, sizeof dx0, hipMemcpyDeviceToHost));
    worst = 0.0;
    for (i = 0; i < N * D; ++i) { double d_ = fabs((double)got[i] - dx_ref[i]); if (d_ > worst) worst = d_; }
    if (worst > 1e-5) { printf("spmm reversed + accumulate: max error %g\n", worst); return 13; }

    /* 4. projection */
    GIST_OK_(gist_gemm_nt_f32(d_a, 2 * D, d_w, 2 * D, d_bias, d_yg, M_OUT, N, M_OUT, 2 * D, d_ws, ws_bytes, NULL));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(got, d_yg, sizeof yg_ref, hipMemcpyDeviceToHost));
    worst = 0.0;
    for (i = 0; i < N * M_OUT; ++i) { double d_ = fabs((double)got[i] - yg_ref[i]); if (d_ > worst) worst = d_; }
    if (worst > 1e-4) { printf("gemm_nt: max error %g\n", worst); return 14; }

    /* 5. error behaviour on device arguments: a leading dimension below the width is refused, nothing launched */
    if (gist_spmm_csr_f32(d_rowptr, d_col, d_x, D - 1, d_y, D, N, D, NULL, NULL, 0, NULL) != GIST_EINVAL) return 15;

    hipFree(d_rowptr); hipFree(d_col); hipFree(d_t_rowptr); hipFree(d_t_col); hipFree(d_x); hipFree(d_y);
    hipFree(d_norm); hipFree(d_dah); hipFree(d_dx); hipFree(d_a); hipFree(d_w); hipFree(d_bias); hipFree(d_yg);
    if (d_ws) hipFree(d_ws);
    printf("cabi device consumer ok\n");
    return 0;
}
