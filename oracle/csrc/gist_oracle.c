/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement (plain C + OpenMP) of the integer/
 * gather parts of GIST's hot path.  Used by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg as the CHECKER / reported baseline; the product
 * (gist_amd/) never links or calls it.
 *
 * Reference call sites restated (paths relative to /root/reference):
 *   oracle_spmm_csr_f32      cluster_gcn/modules.py:223-226  update_all(copy_src,sum) (* norm)
 *                            backward of the same op = this function on the
 *                            transposed CSR with scale_src
 *   oracle_induced_count/fill cluster_gcn/partition_utils.py:20-25  g.subgraph(ids)
 *                            cluster_gcn/sampler.py:34
 *   oracle_transpose_csr     reverse graph used by autograd of update_all
 *
 * Build: gcc -O3 -fopenmp -shared -fPIC (see oracle/build.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* y[v,:] = (accumulate ? y[v,:] : 0) + out_scale[v] * sum_e src_scale[col[e]] * x[col[e],:]
 * out_scale / src_scale may be NULL (= 1).  Edge order inside a row is kept, so
 * the float summation order is the CSR order. */
void oracle_spmm_csr_f32(const int64_t *rowptr, const int64_t *col,
                         const float *x, int64_t ldx,
                         float *y, int64_t ldy,
                         int64_t n_rows, int64_t d,
                         const float *out_scale, const float *src_scale,
                         int accumulate)
{
#pragma omp parallel
    {
        float *acc = (float *)malloc(sizeof(float) * (size_t)(d > 0 ? d : 1));
#pragma omp for schedule(dynamic, 16)
        for (int64_t v = 0; v < n_rows; ++v) {
            for (int64_t j = 0; j < d; ++j) acc[j] = 0.0f;
            for (int64_t e = rowptr[v]; e < rowptr[v + 1]; ++e) {
                const int64_t u = col[e];
                const float *xr = x + u * ldx;
                if (src_scale) {
                    const float s = src_scale[u];
                    for (int64_t j = 0; j < d; ++j) acc[j] += s * xr[j];
                } else {
                    for (int64_t j = 0; j < d; ++j) acc[j] += xr[j];
                }
            }
            float *yr = y + v * ldy;
            const float os = out_scale ? out_scale[v] : 1.0f;
            if (accumulate) {
                for (int64_t j = 0; j < d; ++j) yr[j] += os * acc[j];
            } else {
                for (int64_t j = 0; j < d; ++j) yr[j] = os * acc[j];
            }
        }
        free(acc);
    }
}

/* Node-induced subgraph, pass 1: remap must hold -1 everywhere on entry; on exit
 * remap[ids[i]] = i (caller resets).  deg_out[i] = #neighbours of ids[i] that are
 * in the id set. */
void oracle_induced_count(const int64_t *rowptr, const int64_t *col,
                          const int64_t *ids, int64_t n_ids,
                          int64_t *remap, int64_t *deg_out)
{
    for (int64_t i = 0; i < n_ids; ++i) remap[ids[i]] = i;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n_ids; ++i) {
        const int64_t v = ids[i];
        int64_t c = 0;
        for (int64_t e = rowptr[v]; e < rowptr[v + 1]; ++e) c += (remap[col[e]] >= 0);
        deg_out[i] = c;
    }
}

/* pass 2: sub_rowptr = exclusive scan of deg_out (caller); writes relabelled
 * neighbours in the original edge order. */
void oracle_induced_fill(const int64_t *rowptr, const int64_t *col,
                         const int64_t *ids, int64_t n_ids,
                         const int64_t *remap, const int64_t *sub_rowptr,
                         int64_t *sub_col)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n_ids; ++i) {
        const int64_t v = ids[i];
        int64_t w = sub_rowptr[i];
        for (int64_t e = rowptr[v]; e < rowptr[v + 1]; ++e) {
            const int64_t r = remap[col[e]];
            if (r >= 0) sub_col[w++] = r;
        }
    }
}

/* CSR (n x n) -> CSR of the transpose, stable (rows of the result list their
 * entries in increasing source-row order). */
void oracle_transpose_csr(const int64_t *rowptr, const int64_t *col, int64_t n,
                          int64_t *t_rowptr, int64_t *t_col)
{
    const int64_t nnz = rowptr[n];
    memset(t_rowptr, 0, sizeof(int64_t) * (size_t)(n + 1));
    for (int64_t e = 0; e < nnz; ++e) t_rowptr[col[e] + 1]++;
    for (int64_t i = 0; i < n; ++i) t_rowptr[i + 1] += t_rowptr[i];
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    memcpy(cur, t_rowptr, sizeof(int64_t) * (size_t)n);
    for (int64_t v = 0; v < n; ++v)
        for (int64_t e = rowptr[v]; e < rowptr[v + 1]; ++e) t_col[cur[col[e]]++] = v;
    free(cur);
}
