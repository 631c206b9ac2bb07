/* A plain-C consumer of the boundary: include/gist_hip.h must compile as C99 and every
 * entry point must link from C (tests/test_cabi_exports.py builds and runs this; with no GPU
 * it only exercises host-side entry points and argument validation). */
#include <stdio.h>
#include <string.h>

#include "gist_hip.h"

int main(void) {
    gist_step_plan plan;
    memset(&plan, 0, sizeof plan);
    if (gist_abi_version() != 16) return 1;
    if (gist_gemm_workspace_bytes(2046, 41, 8192) <= 0) return 2;
    if (gist_colsum_partials(129) != 3) return 3;
    /* validation happens before any device work, so these are safe without a GPU */
    if (gist_spmm_csr_f32(NULL, NULL, NULL, 4, NULL, 4, 3, 4, NULL, NULL, 0, NULL) != GIST_EINVAL) return 4;
    if (strstr(gist_last_error(), "null pointer") == NULL) return 5;
    if (gist_sage_step(NULL, NULL, 1, 0, 0.01f, 0.9f, 0.999f, 1e-8f, 0.f, 1, 0, NULL) != GIST_EINVAL) return 6;
    plan.n_layers = 0;
    if (gist_sage_step(&plan, NULL, 1, 0, 0.01f, 0.9f, 0.999f, 1e-8f, 0.f, 1, 0, NULL) != GIST_EINVAL) return 7;
    if (gist_gemm_nt_f32(NULL, 8, NULL, 8, NULL, NULL, 8, 4, 4, 8, NULL, 0, NULL) != GIST_EINVAL) return 8;
    if (gist_extract_batch(NULL, NULL, NULL, NULL, NULL, 4, NULL, NULL, NULL, NULL, NULL, 0, NULL, NULL,
                           4, 4, NULL, 4, NULL, NULL, NULL) != GIST_EINVAL) return 9;
    /* take the address of every remaining entry point so a missing symbol fails the link */
    {
        void *syms[] = {(void *)gist_device_count, (void *)gist_in_degree_norm_f32,
                        (void *)gist_spmm_csr_blocked_f32, (void *)gist_gemm_nn_f32,
                        (void *)gist_gemm_tn_f32, (void *)gist_ln_relu_fwd_f32,
                        (void *)gist_ln_relu_bwd_f32, (void *)gist_dropout_f32,
                        (void *)gist_colsum_f32, (void *)gist_softmax_xent_f32,
                        (void *)gist_adam_f32, (void *)gist_argmax_correct_i32,
                        (void *)gist_induced_mark, (void *)gist_induced_unmark,
                        (void *)gist_fill_i32, (void *)gist_induced_rowptr,
                        (void *)gist_induced_fill, (void *)gist_gather_rows_f32,
                        (void *)gist_gather_i32, (void *)gist_block_gather_f32,
                        (void *)gist_block_scatter_f32, (void *)gist_mean_rows_f32,
                        (void *)gist_timer_create, (void *)gist_timer_destroy,
                        (void *)gist_timer_reset, (void *)gist_timer_count,
                        (void *)gist_timer_read, (void *)gist_class_layer_f32,
                        (void *)gist_class_dw_slabs_f32, (void *)gist_class_layer_takes,
                        (void *)gist_class_dw_slab_bytes, (void *)gist_spmm_csr_drop_prepared_f32,
                        (void *)gist_spmm_prepared_useful, (void *)gist_spmm_block_units_f32,
                        (void *)gist_spmm_block_image_bytes, (void *)gist_gemm_dual_takes,
                        (void *)gist_gemm_nn_tn_dual_f32, (void *)gist_adam_segments_extract_f32,
                        (void *)gist_sage_step_extracts_next, (void *)gist_extract_parts_desc_batch,
                        (void *)gist_ln_relu_bwd_colsum_class_dw_f32};
        size_t i;
        for (i = 0; i < sizeof syms / sizeof syms[0]; ++i)
            if (syms[i] == NULL) return 10;
    }
    printf("cabi consumer ok\n");
    return 0;
}
