/*
 * gist_hip.h -- C ABI of libgist_hip.so: the MI355X (gfx950) implementation of
 * GIST's hot path (GraphSAGE forward/backward over cluster sub-graphs, cluster
 * batch extraction, IST weight-block dispatch/sync).
 *
 * The reference (wolfecameron/GIST) is pure Python on DGL + PyTorch and has no
 * FFI of its own; each entry point below replaces the native work ONE reference
 * call site causes (cited as file:line relative to the reference root), so a
 * maintainer binds them with ctypes at exactly those call sites
 * (INTEGRATION.md shows the stubs).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer (HBM)
 *     unless a parameter is documented as host.
 *   - all matrices are row-major fp32 with an explicit leading dimension (in
 *     elements); graph indices are int32 on the device (reference: g.int(),
 *     cluster_gcn/cluster_gcn_ist_distrib.py:514).
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the default stream); nothing allocates, nothing synchronises, the
 *     caller owns every buffer (workspaces are passed in).
 *   - return value: 0 on success, a negative GIST_E* code otherwise; no C++
 *     exception crosses the boundary.  gist_last_error() returns a static
 *     string describing the last failure on the calling thread.
 */
#ifndef GIST_HIP_H_
#define GIST_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *gist_stream_t;

#define GIST_OK 0
#define GIST_EINVAL (-1)   /* bad argument (null pointer, negative size, misaligned) */
#define GIST_ELAUNCH (-2)  /* hipLaunch / runtime error */
#define GIST_ENOSPACE (-3) /* workspace too small */

const char *gist_last_error(void);
/* ABI version; bumped whenever a signature changes. */
int gist_abi_version(void);   /* currently 16 */
/* Number of visible HIP devices (>= 0) or a negative error. */
int gist_device_count(void);

/* ---------------------------------------------------------------------------
 * Neighbour aggregation
 * ------------------------------------------------------------------------- */

/* norm[v] = 1 / in_degree(v), 0 where the degree is 0.
 * Replaces ISTSAGELayer.get_norm, cluster_gcn/modules.py:239-243
 * (and ClusterIter.get_norm, cluster_gcn/sampler.py:72-76). */
int gist_in_degree_norm_f32(const int32_t *rowptr, int64_t n_rows, float *norm,
                            gist_stream_t stream);

/* y[v, 0:d] = (accumulate ? y[v,:] : 0)
 *           + out_scale[v] * sum_{e in [rowptr[v], rowptr[v+1])} src_scale[col[e]] * x[col[e], 0:d]
 * out_scale / src_scale may be NULL (= 1).  x and y may be column windows of wider
 * buffers (ldx, ldy); they must not overlap.
 * Forward: replaces g.update_all(fn.copy_src, fn.sum) followed by `* norm`,
 * cluster_gcn/modules.py:223-226 (out_scale = norm), writing straight into the
 * right half of the [h | ah] buffer that torch.cat builds at :227.
 * Backward: the same call on the CSR of the reversed graph with src_scale = norm
 * and accumulate = 1 is autograd's gradient of that op. */
int gist_spmm_csr_f32(const int32_t *rowptr, const int32_t *col,
                      const float *x, int64_t ldx, float *y, int64_t ldy,
                      int64_t n_rows, int64_t d,
                      const float *out_scale, const float *src_scale,
                      int accumulate, gist_stream_t stream);

/* Same result as gist_spmm_csr_f32, for row sets made of LOCALITY BLOCKS -- the METIS
 * parts a Cluster-GCN batch is the union of (cluster_gcn/partition_utils.py:21-24): rows
 * [row_blocks[b], row_blocks[b+1]) form block b (row_blocks = NULL: uniform 128-row blocks).
 * Each workgroup stages its block's X tile in LDS once and reads in-block neighbours from
 * LDS; only cross-block neighbours are gathered from L2/HBM.  Blocks larger than 128 rows
 * stay correct (rows beyond the first 128 are gathered globally). */
int gist_spmm_csr_blocked_f32(const int32_t *rowptr, const int32_t *col,
                              const float *x, int64_t ldx, float *y, int64_t ldy,
                              int64_t n_rows, int64_t d,
                              const float *out_scale, const float *src_scale, int accumulate,
                              const int32_t *row_blocks, int64_t n_row_blocks,
                              gist_stream_t stream);

/* The block structure of a row set, built ONCE for every aggregation over the same graph and blocks
 * (a training step aggregates over its batch two or three times forward and again backward): per
 * block the edge counts of its diagonal block (bf16, the matrix-core kernel's A operand) and the
 * ids of each row's neighbours outside the block.  gist_spmm_blocks_bytes(n_row_blocks) bytes
 * (row_blocks = NULL: ceil(n_rows / 128) uniform blocks), 16-byte aligned; valid until rowptr / col /
 * row_blocks change.  gist_spmm_csr_prepared_f32 == gist_spmm_csr_blocked_f32 on the same arguments
 * (same results up to the order of fp32 sums).  Round 4: for a width whose rows are not 16-byte aligned (the
 * F = 602 input layer) the prepared call runs the product on the fp32 matrix cores with its operands read
 * straight from memory (spmm_dense32.hip: 15.7 us where the row-split kernel takes 22). */
int64_t gist_spmm_blocks_bytes(int64_t n_row_blocks);                      /* host function */
int gist_spmm_blocks_prepare(const int32_t *rowptr, const int32_t *col, int64_t n_rows,
                             const int32_t *row_blocks, int64_t n_row_blocks,
                             void *prepared, int64_t prepared_bytes, gist_stream_t stream);
int gist_spmm_csr_prepared_f32(const int32_t *rowptr, const int32_t *col,
                               const float *x, int64_t ldx, float *y, int64_t ldy,
                               int64_t n_rows, int64_t d,
                               const float *out_scale, const float *src_scale, int accumulate,
                               const int32_t *row_blocks, int64_t n_row_blocks,
                               const void *prepared, gist_stream_t stream);

/* The same aggregation with gist_dropout_f32's mask folded in, so that nn.Dropout on the concatenated
 * [h | ah] (cluster_gcn/modules.py:230-231) and its backward cost no pass of their own.  Element
 * (row, c) of y has mask index y_offset + row * mask_ld + c, of x src_offset + row * mask_ld + c
 * (the operands are column windows of a [rows, mask_ld] tensor the mask is defined on).
 *   mode 1 (forward):  what is stored to y is multiplied by y's mask -- y = dropout(aggregate(x));
 *   mode 2 (backward): x is read through its mask and, with accumulate, the old y through y's --
 *                      the aggregation of a gradient whose dropout pass has not been run.
 * Bit-identical to running gist_dropout_f32 after (mode 1) / before (mode 2) the plain call.
 * row_blocks as in gist_spmm_csr_blocked_f32 (NULL: none).  Not every kernel can carry the mask
 * (gist_spmm_drop_takes, host function: 1 if this call is accepted; otherwise GIST_EINVAL). */
int gist_spmm_csr_drop_f32(const int32_t *rowptr, const int32_t *col,
                           const float *x, int64_t ldx, float *y, int64_t ldy,
                           int64_t n_rows, int64_t d,
                           const float *out_scale, const float *src_scale, int accumulate,
                           const int32_t *row_blocks, int64_t n_row_blocks,
                           int mode, float p, uint64_t seed, uint64_t y_offset, uint64_t src_offset,
                           int64_t mask_ld, gist_stream_t stream);
int gist_spmm_drop_takes(int mode, int64_t d, int64_t ldx, int64_t ldy, const float *x, const float *y,
                         int has_row_blocks);
/* gist_spmm_csr_drop_f32 on a row set whose block structure is prepared (gist_spmm_blocks_prepare; NULL: as
 * gist_spmm_csr_drop_f32).  Same results up to the order of fp32 sums. */
int gist_spmm_csr_drop_prepared_f32(const int32_t *rowptr, const int32_t *col,
                                    const float *x, int64_t ldx, float *y, int64_t ldy,
                                    int64_t n_rows, int64_t d,
                                    const float *out_scale, const float *src_scale, int accumulate,
                                    const int32_t *row_blocks, int64_t n_row_blocks,
                                    int mode, float p, uint64_t seed, uint64_t y_offset, uint64_t src_offset,
                                    int64_t mask_ld, const void *prepared, gist_stream_t stream);
/* The reverse aggregation of layer k + 1 (gist_spmm_csr_drop_f32 mode 2, accumulate, src_scale = 1 / deg) whose STORE is the
 * LayerNorm + ReLU backward of layer k (cluster_gcn/modules.py:232-237 differentiated), for rows of at most 256 floats with
 * locality blocks -- one wave holds a whole row in the LDS-staged kernel (config 2, --n-hidden <= 256; round 5, ABI 15).  What
 * would be written to y is d_out; dy = rstd . (g - mean(g) - yhat . mean(g . yhat)), g = d_out . [yhat > 0] (rstd = NULL: dy =
 * g), goes to dy (may be yhat itself), y is only read.  col_partials [gist_spmm_lnb_units(n_row_blocks)][d] receives the
 * column sums of the dy rows each workgroup stored: the bias gradient is the sum of those rows in order
 * (gist_adam_segments_f32 forms it).  dy equals gist_ln_relu_bwd_colsum_f32's to fp32 rounding.  Needs 0 < p < 1 (a step without
 * dropout keeps the separate launches), d % 4 == 0, 128 <= d <= 256.  The FORWARD counterpart (LayerNorm + ReLU of layer k
 * formed in the staging of layer k + 1's aggregation) was built and measured out: profiles/NEGATIVES.md. */
int gist_spmm_csr_drop_lnbwd_f32(const int32_t *rowptr, const int32_t *col, const float *x, int64_t ldx, const float *y,
                                 int64_t ldy, int64_t n_rows, int64_t d, const float *src_scale, const int32_t *row_blocks,
                                 int64_t n_row_blocks, float p, uint64_t seed, uint64_t y_offset, uint64_t src_offset,
                                 int64_t mask_ld, const float *yhat, int64_t ldyh, const float *rstd, float *dy, int64_t lddy,
                                 float *col_partials, int64_t partial_rows, int relu, gist_stream_t stream);
int64_t gist_spmm_lnb_units(int64_t n_row_blocks);      /* rows of col_partials the call writes (host function) */

/* 1 if an aggregation of this width runs on the kernel the prepared block structure is built for (the bf16x3
 * matrix-core kernel, from 1536 columns on): a caller that aggregates over a batch more than once should then
 * prepare its blocks -- and every other prepared call of the batch may use the structure too (the fp32 block-dense
 * kernel takes the widths whose rows are not 16-byte aligned).  Host function. */
int gist_spmm_prepared_useful(int64_t d, int64_t ldx, int64_t ldy, const float *x, const float *y);

/* Full-graph evaluation (cluster_gcn/utils.py:70-80) on a graph whose node ids are ordered by part: the edges between a
 * row block and a column block that share many edges (a dense off-diagonal block of the adjacency) as a counts x features
 * product on the bf16x3 matrix cores (the kernel of gist_spmm_csr_prepared_f32: exact fp32 sums) instead of one gathered
 * row per edge.  Rows must be 16-byte aligned multiples of 4 floats.  Unit u = units[4 u .. 4 u + 3] = (r0, r1, xs0, xs1):
 *   y[r0 .. r1, :d] (+)= out_scale[r] * C_u . x[xs0 .. xs1, :d]
 * with C_u the u-th image of `images`, each gist_spmm_block_image_bytes() long: bf16 edge counts (<= 256: exact), count of
 * (row r, source k) at ((k / 8) * 128 + r) * 8 + k % 8, followed by zeros.  The units of one call must have disjoint
 * output rows (<= 128 rows, <= 128 sources each): a caller with several column blocks per row block issues the j-th
 * pair of every row block together, j = 0, 1, ...  0.7 us of chip time per unit at D = 4096 (an fp32-MFMA form with
 * operands from memory was measured at 1.7 us per pair and removed: profiles/NEGATIVES.md). */
int64_t gist_spmm_block_image_bytes(void);
/* Edges from one block's rows into another block of the batch from which gist_spmm_blocks_prepare treats the two as a
 * pair (their off-diagonal block multiplied like a diagonal one): what a caller compares its part-to-part edge counts
 * with to set gist_step_plan.sibling_parts.  Host function. */
int32_t gist_spmm_pair_min_edges(void);
int gist_spmm_block_units_f32(const int32_t *units, int64_t n_units, const void *images, const float *x, int64_t ldx,
                              float *y, int64_t ldy, int64_t n_rows_y, int64_t d, const float *out_scale,
                              int accumulate, gist_stream_t stream);
/* The same products for CHAINS of units (ABI 14, round 5): chain c = units [chain_ptr[c], chain_ptr[c + 1]) of `units`, all with
 * the same output rows (r0, r1) -- a row block's diagonal block and every dense off-diagonal pair that writes it -- computed by
 * ONE workgroup per (chain, column-tile group) whose accumulators stay in registers across the chain:
 *   y[r0 .. r1, :d] (+)= out_scale[r] * sum over the chain's units u of C_u . x[xs0_u .. xs1_u, :d]
 * y is read (accumulate) and written once per chain instead of once per unit (gist_spmm_block_units_f32 moves 64 KB of X +
 * 128 KB of y per unit and 128-column tile; this form 64 KB of X + the unit's 32-KiB image).  Unit u's image is the u-th of
 * `images`; different chains must have disjoint output rows.  Same layout and alignment rules as above. */
int gist_spmm_block_chains_f32(const int32_t *chain_ptr, int64_t n_chains, const int32_t *units, const void *images,
                               const float *x, int64_t ldx, float *y, int64_t ldy, int64_t n_rows_y, int64_t d,
                               const float *out_scale, int accumulate, gist_stream_t stream);

/* ---------------------------------------------------------------------------
 * Data preparation (HOST function, host pointers)
 * ------------------------------------------------------------------------- */

/* k-way partition of a graph: the stand-in for dgl.transform.metis_partition
 * (cluster_gcn/partition_utils.py:11-18; METIS is not available offline).  rowptr/col: in-edge
 * CSR, t_rowptr/t_col: out-edge CSR (may be NULL for a symmetric graph); neighbours are the
 * union.  Multilevel (round 5): size-constrained label-propagation coarsening (chunked: the edge scans of a
 * chunk of the visiting order run on the host's worker threads, moves are applied in order: the result is a
 * function of (graph, k, seed) alone), greedy initial partition of the coarsest level, and per level while
 * uncoarsening n_passes strict-gain sweeps + localised k-way Fiduccia-Mattheyses searches with rollback (negative-gain
 * moves allowed, the best prefix kept); every part holds at most ceil((1 + imbalance) * n / k) nodes, at least
 * floor((1 - imbalance) * n / k) where the graph allows it, and none is empty.  part[v] in [0, k).  One-time
 * preparation whose result ClusterIter caches in the reference's ../data/{dataset}_{psize}.npy format.
 * Measured (profiles/r05_partitioner.json): the planted cut on the block models (153 k nodes / 1 500 parts in 2.6 s,
 * 1.71 M / 15 000 in ~10 s on 8 cores), 1.21 x the ideal cut on a torus mesh, below a partition built from the
 * ground-truth communities on a power-law community graph. */
int gist_partition_graph(const int32_t *rowptr, const int32_t *col,
                         const int32_t *t_rowptr, const int32_t *t_col, int64_t n, int32_t k,
                         uint64_t seed, int32_t n_passes, float imbalance, int32_t *part);
/* Wall seconds of the stages of the LAST gist_partition_graph call of this process: [0] input graph, [1] clustering,
 * [2] contraction, [3] visiting orders, [4] initial partition + refinement of the coarse levels, [5] refinement of the
 * input level, [6] balance repair, [8] number of levels, [9] vertices of the coarsest level (ABI 14; host function). */
int gist_partition_last_stats(double *out, int32_t n);

/* ---------------------------------------------------------------------------
 * Dense projection (fp32 MFMA, exact fp32 arithmetic)
 * ------------------------------------------------------------------------- */

/* All three forms: sizes < 2^31, leading dimensions < 2^22 elements (GIST_EINVAL otherwise).
 * 16-byte aligned operands with leading dimensions % 4 == 0 (every buffer the engine
 * allocates) take the LDS-DMA staging path; anything else is still exact, only slower.
 *
 * Bytes of workspace gist_gemm_* may need for the given output shape (split-K
 * partial sums); 0 is a valid answer.  Host function. */
int64_t gist_gemm_workspace_bytes(int64_t m, int64_t n, int64_t k);
/* 1 if a call of this shape splits its own operands in the current mode (the pre-split f16x3 / bf16x3 kernels:
 * large workspace, reduces its own k slices -- gist_gemm_slabs_f32 then returns one slab); host function. */
int gist_gemm_splits_operands(int64_t m, int64_t n, int64_t k);

/* Y[m,n] = A[m,k] . W[n,k]^T + bias[n]          (bias may be NULL)
 * Replaces self.linear(h), cluster_gcn/modules.py:233 (nn.Linear: W is [out, 2*in]). */
int gist_gemm_nt_f32(const float *a, int64_t lda, const float *w, int64_t ldw,
                     const float *bias, float *y, int64_t ldy,
                     int64_t m, int64_t n, int64_t k,
                     void *workspace, int64_t workspace_bytes, gist_stream_t stream);

/* Z[m,n] = G[m,k] . W[k,n]
 * Replaces autograd of nn.Linear wrt its input (dZ = dY . W), modules.py:233. */
int gist_gemm_nn_f32(const float *g, int64_t ldg, const float *w, int64_t ldw,
                     float *z, int64_t ldz, int64_t m, int64_t n, int64_t k,
                     void *workspace, int64_t workspace_bytes, gist_stream_t stream);

/* D[m,n] = G[k,m]^T . A[k,n]
 * Replaces autograd of nn.Linear wrt its weight (dW = dY^T . Z), modules.py:233. */
int gist_gemm_tn_f32(const float *g, int64_t ldg, const float *a, int64_t lda,
                     float *d, int64_t ldd, int64_t m, int64_t n, int64_t k,
                     void *workspace, int64_t workspace_bytes, gist_stream_t stream);

/* gist_gemm_{nt,nn,tn}_f32 (layout 0, 1, 2) on the fp32 kernel's split-K path WITHOUT the reduce pass:
 * when the call splits its k range, the partial sums stay as dense slabs [*n_slabs][m][n] at `slabs`
 * (gist_gemm_workspace_bytes(m, n, k) bytes) and c is not written; the CONSUMER sums them in slab
 * order and adds the bias (gist_softmax_xent_slabs_f32 for the class layer's logits,
 * gist_adam_segments_f32 for a weight gradient) -- one launch less per projection where the
 * consumer reads the values anyway.  *n_slabs = 1: c holds the finished result (bias included).
 * Same reference call sites as the three entry points above. */
int gist_gemm_slabs_f32(int layout, const float *a, int64_t lda, const float *b, int64_t ldb,
                        const float *bias, float *c, int64_t ldc, int64_t m, int64_t n, int64_t k,
                        void *slabs, int64_t slab_bytes, int32_t *n_slabs, gist_stream_t stream);

/* How the three entry points above form their products (GIST_GEMM_MODE = bf16x3 | f32 | f16x3 is
 * read once at first use; gist_gemm_set_mode overrides it; process-wide: set it before sizing
 * workspaces).  Inputs, outputs and accumulation are fp32 in every mode, and small or skinny shapes
 * always run on v_mfma_f32_32x32x2_f32.
 *   Mode 2, bf16x3 (the default): the same arithmetic in two kernels.  gist_gemm_nt_f32 / gist_gemm_nn_f32
 *     calls of >= 2 GFLOP with >= 256 output tiles of 64 x 64 that do not reach the thresholds below split
 *     their operands INSIDE the GEMM (convert on load: no pre-pass, no workspace; 1.1-1.3x the fp32 kernel
 *     there; gist_gemm_tn_f32 stays on the fp32 kernel -- tuning hook GIST_TUNE_B3C).  Shapes large enough to fill the chip (>= 128 workgroups of
 *     256 x 128 tiles x k slices, >= 16 GFLOP per call or >= 9 GFLOP inside gist_sage_step, workspace
 *     of gist_gemm_workspace_bytes) carry each fp32 operand as three bf16 pieces, x = b1 + b2 + b3
 *     EXACTLY (3 x 8 = all 24 significant bits, fp32's exponent range, no scales), and accumulate
 *     the six leading cross terms in fp32 on v_mfma_f32_16x16x32_bf16; what is dropped is below one
 *     fp32 rounding of a product.  Error against fp64 at mode 0's level on every operand class tested,
 *     adversarial ones included (tests/test_gemm_b3_gpu.py), at ~0.6x the time.
 *   Mode 0, f32: every shape on v_mfma_f32_32x32x2_f32 -- fp32 products, the arithmetic of the
 *     reference's nn.Linear.
 *   Mode 1, f16x3 (opt-in): each fp32 operand as two f16 halves under one power-of-two scale per
 *     operand row (22 of fp32's 24 significant bits), ah.bh + ah.bl + al.bh in fp32 on
 *     v_mfma_f32_16x16x32_f16 -- error at mode 0's level on the step's operands
 *     (tests/test_gemm_h3_gpu.py) at about half the time, but narrower operand arithmetic than fp32
 *     by construction.
 * In the split modes a NaN or Inf in an operand row makes the corresponding output row / column
 * non-finite (NaN where fp32 would give Inf) and leaves all other outputs unchanged.  All three
 * replace the same call, self.linear(h), cluster_gcn/modules.py:233, and its autograd. */
int gist_gemm_set_mode(int mode);
int gist_gemm_get_mode(void);

/* Tuning hooks: explicit, process-wide overrides of choices the launchers otherwise make
 * themselves (value 0 = the launcher decides; the default for every knob).  They exist for
 * sweeps (scripts/) and so that tests reach both variants of a kernel on one shape; the library
 * never reads the environment on a launch path.  No reference counterpart (the reference has no
 * native code); results are the same whatever the setting, within each kernel's tolerance. */
#define GIST_TUNE_H3_MIN_GFLOP 0  /* smallest projection (GFLOP) the split path takes           */
#define GIST_TUNE_H3_MIN_TILES 1  /* smallest number of 128x128 output tiles for the split path */
#define GIST_TUNE_H3_TM 2         /* A-tile height of the split GEMM: 64 or 128                 */
#define GIST_TUNE_GEMM_TILE 3     /* fp32 GEMM tile (64 or 128); needs GEMM_SPLITS too          */
#define GIST_TUNE_GEMM_SPLITS 4   /* fp32 GEMM split-K factor                                   */
#define GIST_TUNE_SPMM_CHUNK 5    /* rows per XCD chunk of the row-split SpMM                   */
#define GIST_TUNE_SPMM_SPLIT 6    /* row split of the LDS-staged SpMM (1..8)                    */
#define GIST_TUNE_SPMM_KERNEL 7   /* blocked SpMM: 1 = LDS gather kernel, 2 = block-dense bf16x3 MFMA kernel, 3 = fp32 block-dense kernel at every width (prepared calls) */
#define GIST_TUNE_B3C 8           /* convert-on-load bf16x3 GEMM: 1 = never, 2 = also below 0.25 GFLOP */
#define GIST_TUNE_CLASS_FUSED 9   /* class layer of the fused step: 1 = the four-launch sequence (gist_class_layer_f32 off), 2 = its dW slabs as their own launch (not in the LayerNorm backward's grid) */
#define GIST_TUNE_GEMM_DUAL 10    /* backward of a narrow hidden layer: 1 = dZ and dW as two launches (gist_gemm_nn_tn_dual_f32 off) */
#define GIST_TUNE_HOST_THREADS 11 /* worker threads of the host partitioner (0 = the container's CPU quota, at most 8; up to 64 on request); its RESULT does not depend on it */
#define GIST_TUNE_LNB_FUSED 12    /* 1 = the LayerNorm backward of a <= 256-wide hidden layer as its own launch (not in the store of the reverse aggregation above it) */
#define GIST_TUNE_B3C_SPLITS 13   /* k slices of the convert-on-load bf16x3 GEMM (0 = its own choice) */
#define GIST_TUNE_B3_TAIL 14      /* bf16x3 GEMM: 1 = no k slices for the tiles past the last full round of 256 (whole tiles) */
#define GIST_TUNE_COUNT 15
int gist_tuning_set(int knob, double value);
double gist_tuning_get(int knob);

/* Measurement aids.  gist_launch_count: kernel launches this process has issued through the library so far (a step's
 * launch count = the difference across it).  gist_empty_launches: n launches of a kernel that does nothing (grid x block
 * threads) on `stream` -- timed by the caller, the floor of a launch-bound sequence of n kernels on this box. */
uint64_t gist_launch_count(void);
int gist_empty_launches(int32_t n, int32_t grid, int32_t block, gist_stream_t stream);

/* ---------------------------------------------------------------------------
 * Row-wise epilogues of one ISTSAGELayer
 * ------------------------------------------------------------------------- */

/* In place on y[n_rows, d]: yhat = (y - mean) / sqrt(var + eps) per row if
 * use_lynorm (biased variance; nn.LayerNorm(out, elementwise_affine=False),
 * cluster_gcn/modules.py:209,234), then out = relu ? max(yhat,0) : yhat is written
 * to `out` (ldo) -- normally the LEFT half of the next layer's [h | ah] buffer.
 * y keeps yhat (needed by the backward); rstd[n_rows] receives 1/sqrt(var+eps)
 * (NULL: not stored -- inference, or !use_lynorm).  Replaces modules.py:234-236. */
int gist_ln_relu_fwd_f32(float *y, int64_t ldy, float *out, int64_t ldo,
                         float *rstd, int64_t n_rows, int64_t d,
                         int use_lynorm, int relu, float eps, gist_stream_t stream);

/* gist_ln_relu_fwd_f32 with the NEXT layer's dropout folded in: `out` (the left half of the next
 * layer's [h | ah]) receives dropout(h) under gist_dropout_f32's mask at element index
 * offset + row * mask_ld + c (mask_ld = the width of the tensor the mask is defined on, 2 * in of the
 * next layer), `out2` (NULL: none) the undropped h -- the source of the next layer's aggregation,
 * which modules.py:223-231 runs BEFORE the dropout.  p = 0: out = h.  Replaces modules.py:234-236 of
 * layer k and the left half of nn.Dropout, :230-231, of layer k + 1. */
int gist_ln_relu_fwd_drop_f32(float *y, int64_t ldy, float *out, int64_t ldo, float *out2, int64_t ldo2,
                              float *rstd, int64_t n_rows, int64_t d, int use_lynorm, int relu,
                              float eps, float p, uint64_t seed, uint64_t offset, int64_t mask_ld,
                              gist_stream_t stream);

/* gist_ln_relu_fwd_drop_f32 whose pre-norm input is still the split-K slabs of its projection
 * (gist_gemm_slabs_f32, layout NT: n_slabs dense [n_rows, d] arrays slab_stride floats apart): the kernel
 * forms y = sum of the slabs in slab order + bias (gist_gemm_nt_f32's own order: same bits) as it reads the
 * row, so self.linear(h) of modules.py:233 needs no reduce launch in front of :234-236.  n_slabs = 0: y is
 * read as is (slabs, bias ignored); p = 0 and out2 = NULL: no dropout (gist_ln_relu_fwd_f32). */
int gist_ln_relu_fwd_slabs_f32(float *y, int64_t ldy, const float *slabs, int64_t slab_stride,
                               int32_t n_slabs, const float *bias, float *out, int64_t ldo, float *out2,
                               int64_t ldo2, float *rstd, int64_t n_rows, int64_t d, int use_lynorm,
                               int relu, float eps, float p, uint64_t seed, uint64_t offset,
                               int64_t mask_ld, gist_stream_t stream);

/* gist_ln_relu_bwd_f32 that also leaves the bias gradient of the layer in chunks:
 * col_partials[gist_row_chunks16(n_rows)][d] = column sums of dy per 16 consecutive rows.  The sum of the
 * chunks in chunk order is db (gist_colsum_chunks_f32, or gist_adam_segments_f32 where it is consumed).
 * One kernel for d <= 1024 with 16-byte aligned rows; other shapes run the plain backward and a chunk-sum
 * pass (same layout, same summation order).  Replaces autograd of modules.py:233-236 wrt the bias. */
int64_t gist_row_chunks16(int64_t n_rows);                                 /* host function */
int gist_ln_relu_bwd_colsum_f32(const float *d_out, int64_t ldg, const float *yhat, int64_t ldy,
                                const float *rstd, float *dy, int64_t lddy, int64_t n_rows, int64_t d,
                                int use_lynorm, int relu, float *col_partials, gist_stream_t stream);
/* gist_ln_relu_bwd_colsum_f32 and gist_class_dw_slabs_f32 (below: the class layer's weight gradient as 128-row slabs of
 * d_logits^T . z) in ONE launch: in a training step the two are independent -- the slabs need the class layer's dlogits
 * and input, this backward the reverse aggregation of the class layer's dZ -- and each alone fills a fraction of the chip
 * for ~6 us, so their workgroups share a grid (the slabs' first).  Same results bit for bit as the two calls; shapes the
 * one-kernel backward does not take (d > 1024, unaligned rows) run the slabs as their own launch.  gist_sage_step uses it
 * when the class layer is gist_class_layer_f32 and a hidden layer follows below (tuning hook GIST_TUNE_CLASS_FUSED = 2
 * keeps the two launches). */
int gist_ln_relu_bwd_colsum_class_dw_f32(const float *d_out, int64_t ldg, const float *yhat, int64_t ldy,
                                         const float *rstd, float *dy, int64_t lddy, int64_t n_rows, int64_t d,
                                         int use_lynorm, int relu, float *col_partials,
                                         const float *d_logits, int64_t ld_dlogits, const float *z, int64_t ldz,
                                         float *slabs, int64_t slab_bytes, int32_t *n_slabs, int64_t n_rows_cls,
                                         int64_t n_classes, int64_t k, gist_stream_t stream);
/* out[j] = sum over chunks (in order) of partials[chunk][j]. */
int gist_colsum_chunks_f32(const float *partials, int64_t chunks, int64_t d, float *out,
                           gist_stream_t stream);

/* dy[n_rows,d] from d_out: undo relu (mask yhat > 0) and LayerNorm
 * (dy = rstd * (g - mean(g) - yhat * mean(g*yhat))).  dy may alias yhat.
 * Replaces autograd of modules.py:234-236. */
int gist_ln_relu_bwd_f32(const float *d_out, int64_t ldg, const float *yhat, int64_t ldy,
                         const float *rstd, float *dy, int64_t lddy,
                         int64_t n_rows, int64_t d, int use_lynorm, int relu,
                         gist_stream_t stream);

/* In place inverted dropout on z[n_rows, d]: keep with prob 1-p, scale by
 * 1/(1-p).  The mask is a pure function of (seed, offset + row*d + col) so
 * the backward regenerates it by calling the same function on the gradient.
 * Replaces nn.Dropout on the concatenated tensor, modules.py:230-231 (same
 * distribution, not torch's Philox stream -- SURVEY.md section 2.1). */
int gist_dropout_f32(float *z, int64_t ldz, int64_t n_rows, int64_t d, float p,
                     uint64_t seed, uint64_t offset, gist_stream_t stream);

/* z = dropout(g . w): gist_gemm_nn_f32 followed by gist_dropout_f32 on its output (same mask
 * stream: element index offset + row*n + col), as one call.  For a narrow reduction (k <= 64:
 * the class layer, whose dY is [rows, n_classes]) the product is store-bound and one kernel
 * forms the sums in fp32 FMAs and masks what it stores; every other shape runs the two kernels.
 * p = 0: plain product.  Replaces autograd of nn.Linear wrt its input followed by the
 * backward of nn.Dropout, modules.py:230-233. */
int gist_gemm_nn_dropout_f32(const float *g, int64_t ldg, const float *w, int64_t ldw,
                             float *z, int64_t ldz, int64_t m, int64_t n, int64_t k,
                             float p, uint64_t seed, uint64_t offset,
                             void *workspace, int64_t workspace_bytes, gist_stream_t stream);

/* gist_gemm_nn_dropout_f32 that also leaves g's column sums per 16 consecutive rows in
 * g_col_partials[gist_row_chunks16(m)][k] (the class layer's bias gradient in chunks: g = d_logits). */
int gist_gemm_nn_dropout_colsum_f32(const float *g, int64_t ldg, const float *w, int64_t ldw,
                                    float *z, int64_t ldz, int64_t m, int64_t n, int64_t k,
                                    float p, uint64_t seed, uint64_t offset,
                                    void *workspace, int64_t workspace_bytes, float *g_col_partials,
                                    gist_stream_t stream);

/* The CLASS LAYER of a training step in one launch (round 4): logits = z . w^T + bias (z [n_rows, k] = the layer's
 * dropped input [h | ah], w [n_classes, k]); the mean cross entropy of gist_softmax_xent_f32 (row_loss[n_rows] = -log p,
 * d_logits [n_rows, ldg] = (softmax - onehot) / count, zero in the pad columns; the caller reduces row_loss, e.g. inside
 * gist_adam_segments_f32); dz [n_rows, k] = (d_logits . w) under the dropout mask of the layer's input (p, seed, offset:
 * gist_dropout_f32's generator, element index offset + row * k + col; dz = NULL: forward and loss only); and
 * dlogits_col_partials[gist_row_chunks16(n_rows)][n_classes] = d_logits' column sums per 16 rows (the bias gradient in
 * chunks; NULL: skip).  fp32 FMA chains on v_mfma_f32_16x16x4_f32, one workgroup per 16 rows.
 * gist_class_layer_takes: 1 if the shape is taken (n_classes <= 48, k % 64 == 0, k <= 1024, 16-byte aligned operands).
 * Replaces nn.Linear of the last ISTSAGELayer (modules.py:233,299-308), nn.CrossEntropyLoss and their backward
 * (cluster_gcn_ist_distrib.py:411-415) wrt the layer's input. */
int gist_class_layer_takes(int64_t n_rows, int64_t n_classes, int64_t k, int64_t ldz, int64_t ldw,
                           const float *z, const float *w);
int gist_class_layer_f32(const float *z, int64_t ldz, const float *w, int64_t ldw, const float *bias,
                         const int32_t *labels, int64_t count, float *logits, int64_t ldl,
                         float *d_logits, int64_t ldg, float *row_loss, float *dz, int64_t lddz,
                         float p, uint64_t seed, uint64_t offset, float *dlogits_col_partials,
                         int64_t n_rows, int64_t n_classes, int64_t k, gist_stream_t stream);
/* The class layer's weight gradient dW = d_logits^T . z as *n_slabs = ceil(n_rows / 128) dense fp32 slabs
 * [n_classes][k] (one per 128 rows) for the consumer to sum in slab order (gist_adam_segments_f32);
 * gist_class_dw_slab_bytes = the bytes the slabs need.  Autograd of that nn.Linear wrt its weight. */
int64_t gist_class_dw_slab_bytes(int64_t n_rows, int64_t n_classes, int64_t k);
int gist_class_dw_slabs_f32(const float *d_logits, int64_t ldg, const float *z, int64_t ldz,
                            float *slabs, int64_t slab_bytes, int32_t *n_slabs, int64_t n_rows,
                            int64_t n_classes, int64_t k, gist_stream_t stream);

/* The two products of a hidden layer's backward that read the same gradient, in ONE launch (round 4):
 *   dz[m, n] = dy[m, k] . w[k, n]                       (autograd of nn.Linear wrt its input, modules.py:233)
 *   dW[k, n] = dy[m, k]^T . z[m, n]                     (wrt its weight; reduction over the m rows)
 * dW is left as *n_slabs dense fp32 slabs [k][n] at `slabs` for the consumer to sum in slab order
 * (gist_adam_segments_f32), or written to dw when *n_slabs = 1.  Both run the fp32 kernel's own 64 x 64 tiles
 * (v_mfma_f32_32x32x2_f32), so the results equal gist_gemm_nn_f32 / gist_gemm_slabs_f32 bit for bit; what the call
 * removes is the second launch: at the per-rank widths either product fills a fraction of the chip for 10-22 us.
 * gist_gemm_dual_takes: 1 if the shapes are taken (both products small enough that they share the chip: at most
 * 1280 workgroups together; 16-byte aligned operands, leading dimensions % 4 == 0; neither large enough for a
 * split-operand kernel). */
int gist_gemm_dual_takes(int64_t m, int64_t n, int64_t k, int64_t lddy, int64_t ldw, int64_t ldz, int64_t lddz,
                         const float *dy, const float *w, const float *z, const float *dz);
int gist_gemm_nn_tn_dual_f32(const float *dy, int64_t lddy, const float *w, int64_t ldw, float *dz, int64_t lddz,
                             const float *z, int64_t ldz, float *dw, int64_t lddw, int64_t m, int64_t n, int64_t k,
                             void *slabs, int64_t slab_bytes, int32_t *n_slabs, gist_stream_t stream);

/* out[j] = sum_i g[i, j], deterministic two-stage reduction.
 * `partials` must hold gist_colsum_partials(n_rows) * d floats.
 * Replaces autograd of nn.Linear wrt its bias (db), modules.py:233. */
int64_t gist_colsum_partials(int64_t n_rows);
int gist_colsum_f32(const float *g, int64_t ldg, int64_t n_rows, int64_t d,
                    float *partials, float *out, gist_stream_t stream);

/* ---------------------------------------------------------------------------
 * Loss and optimiser
 * ------------------------------------------------------------------------- */

/* Mean cross entropy over rows with mask != 0 (mask NULL = all rows):
 * loss[0] = mean_i -log softmax(logits[i])[labels[i]]; d_logits[n_rows, ldg]
 * = (softmax - onehot) / count for masked rows, 0 elsewhere and in the pad
 * columns [n_classes, ldg).  `count` is the number of masked rows (host int);
 * row_loss[n_rows] receives the per-row -log p (0 for unmasked rows) and is
 * reduced in a fixed order, so the loss is bitwise reproducible.
 * Replaces nn.CrossEntropyLoss + its backward,
 * cluster_gcn/cluster_gcn_ist_distrib.py:384,411-415; cluster_gcn/cluster_gcn.py:76,98-104. */
int gist_softmax_xent_f32(const float *logits, int64_t ldl, const int32_t *labels,
                          const uint8_t *mask, int64_t count, float *row_loss,
                          float *loss, float *d_logits, int64_t ldg, int64_t n_rows,
                          int64_t n_classes, gist_stream_t stream);

/* gist_softmax_xent_f32 whose logits are still split-K slabs (gist_gemm_slabs_f32: dense
 * [n_slabs][n_rows][n_classes]; n_slabs = 0: `logits` already holds the values): the kernel forms
 * logits = slabs summed in slab order + bias (bias may be NULL), stores them to `logits` and proceeds.
 * loss may be NULL: the mean of row_loss is then left to gist_adam_segments_f32 (one launch less). */
int gist_softmax_xent_slabs_f32(float *logits, int64_t ldl, const float *slabs, int64_t slab_stride,
                                int64_t n_slabs, const float *bias, const int32_t *labels,
                                const uint8_t *mask, int64_t count, float *row_loss, float *loss,
                                float *d_logits, int64_t ldg, int64_t n_rows, int64_t n_classes,
                                gist_stream_t stream);

/* One Adam step (coupled L2 like torch.optim.Adam) over a flat parameter arena.
 * step is 1-based.  Replaces optimizer.step(),
 * cluster_gcn/cluster_gcn_ist_distrib.py:405-407,417; cluster_gcn/cluster_gcn.py:78-80,105. */
int gist_adam_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                  int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int64_t step, gist_stream_t stream);

/* gist_adam_f32 with DEFERRED gradient reductions: inside segment s, the gradient of arena element i
 * in [begin, end) is sum_{q < n_src} src[q * stride + (i - begin)], summed in q order -- the split-K
 * slabs of a weight-gradient projection (gist_gemm_slabs_f32) or a bias gradient's row-chunk sums
 * (gist_ln_relu_bwd_colsum_f32, gist_gemm_nn_dropout_colsum_f32) -- formed inside the optimiser kernel,
 * which also writes it to grad[i].  Elements outside every segment use grad[i].  Segments must not
 * overlap; at most 2 * GIST_MAX_LAYERS; `segments` is HOST memory.  row_loss != NULL: loss[0] =
 * sum(row_loss[0..n_loss_rows)) / loss_count as well (the reduction gist_softmax_xent_f32 would launch).
 * Same reference call site as gist_adam_f32. */
typedef struct gist_grad_segment {
    int64_t begin, end;      /* element range of the arena                       */
    const float *src;        /* NULL: grad[i] as is                              */
    int64_t stride;          /* elements between consecutive sources             */
    int32_t n_src;
} gist_grad_segment;
int gist_adam_segments_f32(float *param, float *grad, float *exp_avg, float *exp_avg_sq,
                           int64_t n, float lr, float beta1, float beta2, float eps,
                           float weight_decay, int64_t step,
                           const gist_grad_segment *segments, int64_t n_segments,
                           const float *row_loss, int64_t n_loss_rows, int64_t loss_count, float *loss,
                           gist_stream_t stream);

/* The deferred gradient sums of gist_adam_segments_f32 WITHOUT the optimiser update (ABI 14): grad[i] = sum of the
 * segment's sources in the same order (so a later gist_adam_f32 over the arena is bitwise gist_adam_segments_f32),
 * elements outside every segment untouched.  For a loop that must see complete gradients between loss.backward() and
 * optimizer.step() (cluster_gcn/cluster_gcn.py:103-105): the backward-phase call of gist_sage_step ends with it. */
int gist_grad_segments_finish_f32(float *grad, int64_t n, const gist_grad_segment *segments, int64_t n_segments,
                                  gist_stream_t stream);

/* correct[0] += #{i : mask[i] && argmax_j logits[i,j] == labels[i]} (first max wins,
 * like numpy argmax).  Replaces calc_acc / calc_f1(micro), cluster_gcn/utils.py:47-67. */
int gist_argmax_correct_i32(const float *logits, int64_t ldl, const int32_t *labels,
                            const uint8_t *mask, int32_t *correct, int64_t n_rows,
                            int64_t n_classes, gist_stream_t stream);

/* ---------------------------------------------------------------------------
 * Cluster batch extraction (node-induced subgraph), device resident
 * ------------------------------------------------------------------------- */

/* remap[ids[i]] = i.  remap holds -1 everywhere else (caller initialises once
 * with gist_fill_i32 and calls gist_induced_unmark after the batch). */
int gist_induced_mark(const int32_t *ids, int64_t n_ids, int32_t *remap, gist_stream_t stream);
int gist_induced_unmark(const int32_t *ids, int64_t n_ids, int32_t *remap, gist_stream_t stream);
int gist_fill_i32(int32_t *p, int64_t n, int32_t value, gist_stream_t stream);

/* Stream-ordered host <-> device traffic as KERNELS on pinned host memory (hipHostMalloc: device-accessible at its host
 * address), ABI 14: the training loop's queue then holds launches only -- no copy command, no event object, and the host
 * never sits in a busy-waiting runtime call (the loop's host thread shares a CPU quota with everything else in its
 * container: profiles/r05_module_path.md).
 *   gist_copy_i32      dst[i] = src[i]; either side may be pinned host memory (the per-epoch part order,
 *                      cluster_gcn/sampler.py:55,92, built on the host once per epoch);
 *   gist_publish_i64   host_word[0] = *device_word (0 if NULL), then host_word[1] = tag: a progress mark the host
 *                      polls (the extraction's error word of gist_extract_parts_batch + the epoch it belongs to). */
int gist_copy_i32(const int32_t *src, int32_t *dst, int64_t n, gist_stream_t stream);
int gist_publish_i64(const int64_t *device_word, int64_t tag, int64_t *host_word, gist_stream_t stream);

/* sub_rowptr[0..n_ids] = exclusive scan of the induced degree of ids[i] in the
 * CSR (rowptr, col): #neighbours u of ids[i] with remap[u] >= 0. */
int gist_induced_rowptr(const int32_t *rowptr, const int32_t *col, const int32_t *ids,
                        int64_t n_ids, const int32_t *remap, int32_t *sub_rowptr,
                        gist_stream_t stream);

/* sub_col[sub_rowptr[i] ...] = remap[u] for the kept neighbours of ids[i], in the
 * original edge order.  sub_col_capacity guards the buffer (entries beyond it are
 * dropped and GIST_ENOSPACE cannot be reported asynchronously, so size it with
 * the full-degree sum of the batch).
 * The three calls together replace g.subgraph(node_ids),
 * cluster_gcn/partition_utils.py:20-25 and cluster_gcn/sampler.py:34. */
int gist_induced_fill(const int32_t *rowptr, const int32_t *col, const int32_t *ids,
                      int64_t n_ids, const int32_t *remap, const int32_t *sub_rowptr,
                      int32_t *sub_col, int64_t sub_col_capacity, gist_stream_t stream);

/* The whole cluster-batch extraction in ONE call (5 launches): mark, induced row
 * pointers of the in-edge AND out-edge CSR (+ norm = 1/in-degree, modules.py:239-243),
 * scans, fills, then feature/label gather into z0 (the left half of layer 0's [h | ah]
 * buffer) and remap reset.  Same result as the fine-grained calls above.  Replaces
 * get_subgraph + cluster.to(device), cluster_gcn/partition_utils.py:20-25 and
 * cluster_gcn_ist_distrib.py:409.  labels_all may be NULL. */
int gist_extract_batch(const int32_t *g_rowptr, const int32_t *g_col,
                       const int32_t *g_t_rowptr, const int32_t *g_t_col,
                       const int32_t *ids, int64_t n, int32_t *remap,
                       int32_t *rowptr, int32_t *col, int32_t *t_rowptr, int32_t *t_col,
                       int64_t col_capacity, float *norm,
                       const float *feat, int64_t ld_feat, int64_t n_feat,
                       float *z0, int64_t ldz0,
                       const int32_t *labels_all, int32_t *labels, gist_stream_t stream);

/* gist_extract_batch with layer 0's dropout folded into the feature gather: z0 receives
 * dropout(features) under gist_dropout_f32's mask (element index offset + i * mask_ld + c; mask_ld =
 * 2 * n_feat, the width of layer 0's [h | ah]) and x0 [n, n_feat] (ldx0) the features themselves, which
 * layer 0's aggregation reads (modules.py:223-231 aggregates before it drops).  p = 0: z0 = x0. */
int gist_extract_batch_drop(const int32_t *g_rowptr, const int32_t *g_col,
                            const int32_t *g_t_rowptr, const int32_t *g_t_col,
                            const int32_t *ids, int64_t n, int32_t *remap,
                            int32_t *rowptr, int32_t *col, int32_t *t_rowptr, int32_t *t_col,
                            int64_t col_capacity, float *norm,
                            const float *feat, int64_t ld_feat, int64_t n_feat,
                            float *z0, int64_t ldz0,
                            const int32_t *labels_all, int32_t *labels,
                            float *x0, int64_t ldx0, float p, uint64_t seed, uint64_t offset,
                            int64_t mask_ld, gist_stream_t stream);

/* The same extraction in ONE launch for batches that are unions of parts of a fixed partition -- what
 * ClusterIter yields (cluster_gcn/sampler.py:85-93): node_part[v] = (v's part, v's position in the part's id
 * list) (int32 pairs, static), part_slot[p] = (the batch of the current epoch that part p belongs to, -1:
 * none; the batch row of its first node) (int32 pairs, uploaded with the epoch's part order), so membership
 * needs no mark pass and ids[row0(p) + pos] must be the pos-th node of part p.  Both 8-byte aligned.  One
 * kernel counts, scans (a look-back over its own workgroups' totals: no barrier) and fills both CSRs while
 * a third group of workgroups gathers features and labels; x0 != NULL folds layer 0's dropout in as
 * gist_extract_batch_drop does.  Same result as gist_extract_batch, bit for bit.
 * scratch: gist_extract_parts_scratch_bytes(n_max) bytes, 8-byte aligned, ZEROED ONCE by the caller (it
 * carries the workgroups' published totals, tagged with a per-launch number; word [1] != 0 after a call =
 * a workgroup gave up waiting for a predecessor after ~1 s, results invalid).  Calls sharing a scratch must
 * be stream-ordered.  gist_extract_parts_supported(n_max): 1 if n_max is in range.  Host functions both. */
int64_t gist_extract_parts_scratch_bytes(int64_t n_max);
int gist_extract_parts_supported(int64_t n_max);
int gist_extract_parts_batch(const int32_t *g_rowptr, const int32_t *g_col,
                             const int32_t *g_t_rowptr, const int32_t *g_t_col,
                             const int32_t *ids, int64_t n, int64_t n_max,
                             const int32_t *node_part, const int32_t *part_slot, int32_t batch,
                             int32_t *rowptr, int32_t *col, int32_t *t_rowptr, int32_t *t_col,
                             int64_t col_capacity, float *norm,
                             const float *feat, int64_t ld_feat, int64_t n_feat,
                             float *z0, int64_t ldz0,
                             const int32_t *labels_all, int32_t *labels,
                             float *x0, int64_t ldx0, float p, uint64_t seed, uint64_t offset,
                             int64_t mask_ld, void *scratch, gist_stream_t stream);

/* gist_extract_parts_batch's arguments as a structure, and the optimiser launch that ALSO extracts the next batch
 * (round 4): gist_adam_segments_f32 and gist_extract_parts_batch(*next) in ONE grid -- after the backward pass nothing
 * reads the batch buffers, and the next batch depends on nothing the step computes, so its extraction (21 us of
 * dependent memory round trips for ~2000 rows) runs beside the optimiser instead of in front of the next step.  Same
 * results as the two calls.  Same reference call sites (sampler.py:85-93 / optimizer.step()). */
typedef struct gist_extract_parts_desc {
    const int32_t *g_rowptr, *g_col, *g_t_rowptr, *g_t_col;
    const int32_t *ids; int64_t n, n_max;
    const int32_t *node_part, *part_slot; int32_t batch;
    int32_t *rowptr, *col, *t_rowptr, *t_col; int64_t col_capacity; float *norm;
    const float *feat; int64_t ld_feat, n_feat; float *z0; int64_t ldz0;
    const int32_t *labels_all; int32_t *labels;
    float *x0; int64_t ldx0; float p; uint64_t seed, offset; int64_t mask_ld;
    void *scratch;
    /* layer 0's aggregation formed by the extraction itself (ABI 13; ah == NULL: not).  A batch is a union of WHOLE parts,
     * so the neighbours of a row inside its own part are the same in every batch: feat_intra[v] = sum of feat[u] over
     * v's in-neighbours u in v's part, computed once per run (gist_spmm_csr_f32 on the intra-part edges); the pass that
     * filters the row's in-edges adds the few kept neighbours in the batch's OTHER parts and writes
     * ah[i] = norm[i] . (feat_intra[ids[i]] + sum of those) -- with the mask of mask index offset + i * mask_ld + n_feat + c
     * when x0 != NULL -- to ah + i * ldz0.  Replaces the model's first g.update_all(copy_src, sum) * norm
     * (modules.py:223-226) for the input features; sums in another (fixed) order than gist_spmm_csr_f32. */
    const float *feat_intra; int64_t ld_intra;
    float *ah;
} gist_extract_parts_desc;
/* gist_extract_parts_batch with its arguments as the structure (the only form that carries feat_intra / ah). */
int gist_extract_parts_desc_batch(const gist_extract_parts_desc *desc, gist_stream_t stream);
int gist_adam_segments_extract_f32(float *param, float *grad, float *exp_avg, float *exp_avg_sq,
                                   int64_t n, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, int64_t step,
                                   const struct gist_grad_segment *segments, int64_t n_segments,
                                   const float *row_loss, int64_t n_loss_rows, int64_t loss_count, float *loss,
                                   const gist_extract_parts_desc *next, gist_stream_t stream);

/* dst[i, 0:d] = src[ids[i], 0:d]  -- the ndata['feat'] gather of g.subgraph
 * (partition_utils.py:23) written straight into the left half of layer 0's
 * [h | ah] buffer (ldd). */
int gist_gather_rows_f32(const float *src, int64_t lds, const int32_t *ids, int64_t n_ids,
                         int64_t d, float *dst, int64_t ldd, gist_stream_t stream);
int gist_gather_i32(const int32_t *src, const int32_t *ids, int64_t n_ids, int32_t *dst,
                    gist_stream_t stream);

/* ---------------------------------------------------------------------------
 * IST weight blocks (cluster_gcn/cluster_gcn_ist_distrib.py:100-367)
 * ------------------------------------------------------------------------- */

/* dst[i, j] = src[row_idx[i], col_idx[j]]   i < n_rows, j < n_cols
 * row_idx / col_idx may be NULL (= identity).  Replaces the slicing in
 * dispatch_model / ini_sync_dispatch_model, e.g. W[:, full_prev][next, :],
 * cluster_gcn_ist_distrib.py:203-226,291-313. */
int gist_block_gather_f32(const float *src, int64_t lds, const int32_t *row_idx,
                          const int32_t *col_idx, int64_t n_rows, int64_t n_cols,
                          float *dst, int64_t ldd, gist_stream_t stream);

/* dst[row_idx[i], col_idx[j]] = src[i, j].  Replaces the write-back in sync_model,
 * cluster_gcn_ist_distrib.py:106-133,136-195. */
int gist_block_scatter_f32(const float *src, int64_t lds, const int32_t *row_idx,
                           const int32_t *col_idx, int64_t n_rows, int64_t n_cols,
                           float *dst, int64_t ldd, gist_stream_t stream);

/* out[j] = (1/n_src) * sum_s src[s * stride + j], summed in s order.  Replaces
 * all_reduce(SUM)/num_subnet of the shared last-layer bias after the packed
 * all-gather, cluster_gcn_ist_distrib.py:38-41,103. */
int gist_mean_rows_f32(const float *src, int64_t stride, int64_t n_src, int64_t n,
                       float *out, gist_stream_t stream);

/* ---------------------------------------------------------------------------
 * Whole training iteration in one call (native step driver)
 * ------------------------------------------------------------------------- */

/* ---------------------------------------------------------------------------
 * Data preparation
 * ------------------------------------------------------------------------- */

/* sklearn.preprocessing.StandardScaler as the reference applies it before training
 * (cluster_gcn/cluster_gcn_ist_distrib.py:492-499, cluster_gcn/cluster_gcn.py:37-44), in place on
 * the device-resident feature matrix x[n_rows, d]: mean[c] and the POPULATION variance var[c] of
 * every column over the rows fit_rows[0..n_fit_rows) (the train nodes; NULL = the first n_fit_rows
 * rows), accumulated in float64 in a fixed order; then every row becomes
 * f32(f32(x - mean) / scale) with scale = sqrt(var), 1 where var == 0.  mean / var are outputs
 * (float64 [d], device).  Workspace: gist_standard_scaler_workspace_bytes (host function). */
int64_t gist_standard_scaler_workspace_bytes(int64_t n_fit_rows, int64_t d);
int gist_standard_scaler_f32(float *x, int64_t ld, int64_t n_rows, int64_t d,
                             const int32_t *fit_rows, int64_t n_fit_rows, double *mean, double *var,
                             void *workspace, int64_t workspace_bytes, gist_stream_t stream);

#define GIST_MAX_LAYERS 16

/* One SAGE layer's buffers (all device pointers, all preallocated by the caller). */
typedef struct gist_layer_desc {
    int64_t n_in, n_out;
    float *W, *b;          /* [n_out, 2*n_in], [n_out]                      */
    float *dW, *db;        /* gradients, same shapes                       */
    float *Z; int64_t ldz; /* [n_max, 2*n_in] = [h | ah]                   */
    float *Y; int64_t ldy; /* [n_max, n_out] pre-norm -> yhat -> dY        */
    float *rstd;           /* [n_max] (unused by the last layer)           */
} gist_layer_desc;

struct gist_timer;
typedef struct gist_step_plan {
    int32_t n_layers;              /* L+1 SAGE layers                               */
    int32_t use_layernorm;
    float p_drop;                  /* 0 = no dropout                                */
    uint64_t seed;
    gist_layer_desc layer[GIST_MAX_LAYERS];
    float *dlogits; int64_t ldc;   /* [n_max, ldc], ldc >= n_classes (padded)       */
    float *dZ;                     /* scratch [n_max * max_k(2*n_in_k, k>=1)]       */
    float *partials;               /* colsum scratch                                */
    float *row_loss, *loss;        /* [n_max], [1]                                  */
    void *workspace; int64_t workspace_bytes;   /* split-K scratch                  */
    void *workspace2; int64_t workspace2_bytes; /* reserved (second scratch), may be NULL    */
    float *params, *grads, *exp_avg, *exp_avg_sq; int64_t n_params;   /* flat arenas */
    /* resident training graph + the batch buffers the extraction fills */
    const int32_t *g_rowptr, *g_col, *g_t_rowptr, *g_t_col;
    const float *feat; int64_t ld_feat;
    const int32_t *labels_all;
    int32_t *remap;
    int32_t *rowptr, *col, *t_rowptr, *t_col; int64_t col_capacity;
    float *norm; int32_t *labels;
    struct gist_timer *timer;      /* NULL = no timing */
    /* Split projection path inside the step (gist_gemm_set_mode 1): with h3_workspace set the
     * step keeps the f16-split operands of its large projections itself -- activations are split
     * (with dropout applied on the fly) by one kernel per layer that emits both the forward and
     * the weight-gradient layout, weights once per step, gradients once per layer with the row
     * and column maxima taken from the LayerNorm-backward and bias-gradient kernels -- instead
     * of once per GEMM call.  n_max = rows the batch buffers were sized for; feat_absmax = an
     * upper bound of |feat| (0 = unknown: layer 0 then takes the per-call path).  A layer qualifies
     * when its three projections have >= 64 output tiles and >= 4 GFLOP each.  In mode 2 (bf16x3) the
     * same workspace holds the three-piece bf16 operands instead (6 bytes per element, no scales or
     * maxima; a layer qualifies with >= 128 workgroups of 256 x 128 tiles x k slices and >= 9 GFLOP per
     * projection).  Size
     * the workspace with gist_step_h3_workspace_bytes IN THE MODE the steps will run in; NULL / too
     * small / sized in another mode = per-call path. */
    int64_t n_max;
    float feat_absmax;
    void *h3_workspace; int64_t h3_workspace_bytes;
    /* Locality blocks of the CURRENT batch (set per call; the plan is host memory): rows
     * [row_blocks[b], row_blocks[b+1]) = the b-th METIS part of the batch, at most 128 rows each
     * (longer parts cut).  With it the wide aggregations run gist_spmm_csr_blocked_f32 (X tile of
     * a part staged in LDS once); NULL = gist_spmm_csr_f32. */
    const int32_t *row_blocks; int64_t n_row_blocks;
    /* With row_blocks: room for the batch's prepared block structure, both orientations
     * (2 * gist_spmm_blocks_bytes(n_row_blocks) bytes, 16-byte aligned).  The step then prepares the
     * blocks once after the extraction and its wide aggregations run gist_spmm_csr_prepared_f32;
     * NULL / too small = every aggregation builds what it needs itself. */
    void *spmm_prepared; int64_t spmm_prepared_bytes;
    /* Fused sequence (fuse != 0; every buffer optional, NULL = that fusion is off).  A training step is
     * launch-bound at small widths (per-rank h = 512: 41 launches, half of them on the ~4.5 us launch
     * floor), so the step removes launches that only move a result from one kernel to the next:
     *   - dropout is applied where its operand is produced or consumed: hsrc[k] = [n_max, n_in_k] holds
     *     the UNDROPPED input of layer k (the aggregation's source) while Z_k receives dropout([h | ah])
     *     from the feature gather / LayerNorm epilogue (left half) and the aggregation's store (right
     *     half); the gradient's mask is applied as the reverse aggregation reads it
     *     (gist_ln_relu_fwd_drop_f32, gist_extract_batch_drop, gist_spmm_csr_drop_f32);
     *   - bias gradients leave the LayerNorm backward / the class layer's dZ kernel as 16-row chunk sums
     *     in col_partials (sum over layers of gist_row_chunks16(n_max) * n_out floats) and split-K
     *     projections whose consumer reads the values anyway keep their slabs in fused_workspace
     *     (gist_step_fused_workspace_bytes): the class layer's logits are summed by the loss kernel, the
     *     weight gradients and the chunk sums by the optimiser (gist_adam_segments_f32), which also
     *     reduces the loss.
     * Results equal the un-fused sequence's up to the summation order of the bias gradients. */
    int32_t fuse;
    float *hsrc[GIST_MAX_LAYERS]; int64_t ld_hsrc[GIST_MAX_LAYERS];   /* ld % 4 == 0 and >= n_in + 2 keeps the 16-byte gathers */
    float *col_partials;
    void *fused_workspace; int64_t fused_workspace_bytes;
    /* One-launch extraction (gist_extract_parts_batch) when the batch is a union of parts: the static
     * node_part, the epoch's part_slot, this batch's index in the epoch (set per call) and the zeroed
     * scratch; any NULL = gist_extract_batch (5 launches). */
    const int32_t *node_part, *part_slot;
    int32_t batch_index;
    void *extract_scratch;
    /* GIST_STEP_EXTRACT_NEXT (round 4): the NEXT batch of the same epoch (same node_part / part_slot tables), extracted
     * into the batch buffers beside this step's optimiser launch (gist_adam_segments_extract_f32), with layer 0's
     * dropout mask at next_drop_offset (= the drop_offset of the next call) folded in under the rules of this call.
     * The next call then passes GIST_STEP_PREEXTRACTED instead of GIST_STEP_EXTRACT. */
    const int32_t *next_ids;
    int64_t next_n;
    int32_t next_batch_index;
    uint64_t next_drop_offset;
    /* the intra-part neighbour sums of the input features (gist_extract_parts_desc.feat_intra): when set, the one-launch
     * extraction forms layer 0's aggregation and the step skips that launch */
    const float *feat_intra; int64_t ld_feat_intra;
    /* Sibling parts (round 6, set per call like row_blocks): non-zero = two parts of this batch may share hundreds of
     * edges (one community cut in two): the batch's block structure is prepared WITH pairs (gist_spmm_blocks_prepare) and
     * every wide aggregation is followed by its pairs launch.  0 = the caller knows there are none (gist_amd: from the
     * part-to-part edge counts of the training graph, once per run): no search, no second launch.  Either value is
     * correct for any batch; 0 on a batch that has sibling parts is the slow path (their rows walk their edge lists). */
    int32_t sibling_parts;
} gist_step_plan;

/* Bytes of fused_workspace / floats of col_partials the plan's shapes need.  Host functions. */
int64_t gist_step_fused_workspace_bytes(const gist_step_plan *plan);
int64_t gist_step_col_partials_floats(const gist_step_plan *plan);
/* Slab bytes the fused sequence reserves for layer k's weight-gradient projection (k == n_layers: the
 * class layer's logits): what a caller issuing the same sequence op by op passes to gist_gemm_slabs_f32
 * to get the same split counts. */
int64_t gist_step_fused_slab_bytes(const gist_step_plan *plan, int32_t k);

/* Bytes of h3_workspace the plan's shapes need in the current GEMM mode (0: no layer qualifies, or
 * mode 0). Host function. */
int64_t gist_step_h3_workspace_bytes(const gist_step_plan *plan);
/* The same for GEMM mode `mode` (0, 1, 2) whatever the current one: size for every mode the steps may be
 * switched to without changing the process-wide mode under other threads' launches. */
int64_t gist_step_h3_workspace_bytes_mode(const gist_step_plan *plan, int mode);

/* Optional per-kernel timing with HIP events recorded on the launch stream by the step
 * driver around every SpMM and GEMM call (what bench.py's `roofline` is computed from).
 * kind: 0 = SpMM (m = rows, n = source rows, k = width), 1 = GEMM call (m, n, k; in mode 1 it
 * includes the split pre-pass), 2 = the split GEMM's main kernel alone (nested in a kind-1
 * record), 3 = split pre-pass work of the step outside a GEMM call (weights, gradients). */
typedef struct gist_timer gist_timer;
gist_timer *gist_timer_create(int64_t capacity);
void gist_timer_destroy(gist_timer *t);
void gist_timer_reset(gist_timer *t);
int64_t gist_timer_count(const gist_timer *t);
/* Valid after the stream has been synchronised. Returns 0 or a negative code. */
int gist_timer_read(gist_timer *t, int64_t i, float *ms, int32_t *kind, int64_t *m, int64_t *n,
                    int64_t *k);

#define GIST_STEP_EXTRACT 1   /* build the batch from ids (else: batch buffers already valid) */
#define GIST_STEP_TRAIN 2     /* dropout on, backward + Adam (else: forward + loss only)      */
#define GIST_STEP_EXTRACT_NEXT 4   /* TRAIN only: also extract plan->next_* (gist_sage_step_extracts_next says whether this call can) */
#define GIST_STEP_PREEXTRACTED 8   /* TRAIN only: the batch buffers hold THIS batch, extracted by the previous call's EXTRACT_NEXT */
/* Phases (ABI 14, TRAIN only): the reference's loop body is four statements -- `pred = model(cluster)`, `loss = loss_f(...)`,
 * `loss.backward()`, `optimizer.step()` (cluster_gcn/cluster_gcn.py:96-105, cluster_gcn_ist_distrib.py:410-417) -- and a
 * script that keeps them (gist_amd/modules.py: GCN.forward, nn.CrossEntropyLoss, optim.Adam) issues the SAME iteration as
 * three calls with the same (plan, ids, n, drop_offset, other flags):
 *   FORWARD    extraction (EXTRACT) -> forward -> logits, mean CE in plan->loss (complete on return), and what the fused
 *              class layer produces beside them for the standard loss (dlogits, the class layer's dZ and bias chunk sums);
 *   BACKWARD   the backward pass from those; the gradient arena is COMPLETE on return (deferred split-K slabs / chunk sums
 *              are summed by gist_grad_segments_finish_f32 in the optimiser's order).  With GIST_STEP_DLOGITS_GIVEN the caller
 *              has overwritten plan->dlogits with the gradient of ITS loss w.r.t. the logits: the class layer's dZ is
 *              recomputed from it;
 *   OPTIMIZER  Adam over the arena from plan->grads (+ EXTRACT_NEXT: the next batch's extraction in the same grid).
 * No phase bit (or all three) = the whole iteration in one call, as before.  Parameters after the three calls are bitwise
 * those of the one-call step. */
#define GIST_STEP_PHASE_FORWARD 16
#define GIST_STEP_PHASE_BACKWARD 32
#define GIST_STEP_PHASE_OPTIMIZER 64
#define GIST_STEP_DLOGITS_GIVEN 128

/* One iteration of the reference's training loop on the batch whose node ids (in the
 * training graph) are ids[0..n): induced subgraph + feature/label gather
 * (cluster_gcn/partition_utils.py:20-25, cluster_gcn_ist_distrib.py:409), GCN.forward
 * (cluster_gcn/modules.py:310-314), mean CE over the batch rows, backward, Adam
 * (cluster_gcn_ist_distrib.py:410-417 / cluster_gcn/cluster_gcn.py:98-105).
 * `plan` is HOST memory; drop_offset is the dropout counter base for this step
 * (layer k uses drop_offset + sum_{j<k} round_up(n*2*n_in_j, 2)); adam_step is 1-based.
 * No host synchronisation.
 * CONTRACT -- one TRAIN step per extraction: with plan->fuse and dropout, a GIST_STEP_EXTRACT | GIST_STEP_TRAIN
 * call folds layer 0's mask into the feature gather (layer[0].Z's left half then holds dropout(features), hsrc[0]
 * the features).  A later GIST_STEP_TRAIN call WITHOUT GIST_STEP_EXTRACT on the same buffers would aggregate and
 * drop those dropped values again: re-extract (pass GIST_STEP_EXTRACT) for every training step.  Forward-only calls
 * (no GIST_STEP_TRAIN) on extracted buffers are fine. */
int gist_sage_step(const gist_step_plan *plan, const int32_t *ids, int64_t n,
                   uint64_t drop_offset, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int64_t adam_step, int flags, gist_stream_t stream);
/* 1 if gist_sage_step(plan, ., n, ., flags | GIST_STEP_EXTRACT_NEXT) extracts plan->next_* beside its optimiser launch
 * (a fused TRAIN step on a union-of-parts batch whose optimiser is gist_adam_segments_f32, next_n <= n_max); 0 if the
 * flag would be ignored.  Host function. */
int gist_sage_step_extracts_next(const gist_step_plan *plan, int64_t n, int flags);

#ifdef __cplusplus
}
#endif
#endif /* GIST_HIP_H_ */
