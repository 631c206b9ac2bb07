"""ctypes binding of libgist_hip.so (C ABI declared in include/gist_hip.h).

The library is the product: there is NO fallback.  If it is missing or fails to
load, importing a compute path raises GistLibraryError -- loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GIST_LIB_PATH: dev override to A/B a variant build (gist_amd/build.py GIST_LIB_OUT=...)
LIB_PATH = os.environ.get('GIST_LIB_PATH') or os.path.join(_HERE, 'libgist_hip.so')

ABI_VERSION = 16


class GistLibraryError(RuntimeError):
    pass


class GistError(RuntimeError):
    """A gist_* entry point returned a negative code."""


_p = ctypes.c_void_p
_i64 = ctypes.c_int64
_i32 = ctypes.c_int32
_int = ctypes.c_int
_f = ctypes.c_float
_u64 = ctypes.c_uint64

# name -> (restype, argtypes); mirrors include/gist_hip.h one to one
SIGNATURES = {
    'gist_last_error': (ctypes.c_char_p, []),
    'gist_abi_version': (_int, []),
    'gist_device_count': (_int, []),
    'gist_in_degree_norm_f32': (_int, [_p, _i64, _p, _p]),
    'gist_spmm_csr_f32': (_int, [_p, _p, _p, _i64, _p, _i64, _i64, _i64, _p, _p, _int, _p]),
    'gist_partition_last_stats': (_int, [_p, _i32]),
    'gist_partition_graph': (_int, [_p, _p, _p, _p, _i64, _i32, _u64, _i32, _f, _p]),
    'gist_spmm_csr_blocked_f32': (_int, [_p, _p, _p, _i64, _p, _i64, _i64, _i64, _p, _p, _int, _p, _i64, _p]),
    'gist_spmm_blocks_bytes': (_i64, [_i64]),
    'gist_spmm_blocks_prepare': (_int, [_p, _p, _i64, _p, _i64, _p, _i64, _p]),
    'gist_spmm_csr_prepared_f32': (_int, [_p, _p, _p, _i64, _p, _i64, _i64, _i64, _p, _p, _int, _p, _i64, _p, _p]),
    'gist_gemm_workspace_bytes': (_i64, [_i64, _i64, _i64]),
    'gist_gemm_set_mode': (_int, [_int]),
    'gist_gemm_get_mode': (_int, []),
    'gist_tuning_set': (_int, [_int, ctypes.c_double]),
    'gist_tuning_get': (ctypes.c_double, [_int]),
    'gist_launch_count': (ctypes.c_uint64, []),
    'gist_empty_launches': (_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _p]),
    'gist_gemm_nt_f32': (_int, [_p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _i64, _p, _i64, _p]),
    'gist_gemm_nn_f32': (_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _p, _i64, _p]),
    'gist_gemm_tn_f32': (_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _p, _i64, _p]),
    'gist_ln_relu_fwd_f32': (_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _int, _int, _f, _p]),
    'gist_ln_relu_bwd_f32': (_int, [_p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _int, _int, _p]),
    'gist_gemm_nn_dropout_f32': (_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _f, _u64, _u64, _p, _i64, _p]),
    'gist_dropout_f32': (_int, [_p, _i64, _i64, _i64, _f, _u64, _u64, _p]),
    'gist_colsum_partials': (_i64, [_i64]),
    'gist_colsum_f32': (_int, [_p, _i64, _i64, _i64, _p, _p, _p]),
    'gist_softmax_xent_f32': (_int, [_p, _i64, _p, _p, _i64, _p, _p, _p, _i64, _i64, _i64, _p]),
    'gist_adam_f32': (_int, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _i64, _p]),
    'gist_argmax_correct_i32': (_int, [_p, _i64, _p, _p, _p, _i64, _i64, _p]),
    'gist_induced_mark': (_int, [_p, _i64, _p, _p]),
    'gist_induced_unmark': (_int, [_p, _i64, _p, _p]),
    'gist_fill_i32': (_int, [_p, _i64, _i32, _p]),
    'gist_copy_i32': (_int, [_p, _p, _i64, _p]),
    'gist_publish_i64': (_int, [_p, _i64, _p, _p]),
    'gist_induced_rowptr': (_int, [_p, _p, _p, _i64, _p, _p, _p]),
    'gist_induced_fill': (_int, [_p, _p, _p, _i64, _p, _p, _p, _i64, _p]),
    'gist_extract_batch': (_int, [_p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _i64, _p, _p, _i64, _i64, _p, _i64, _p, _p, _p]),
    'gist_gather_rows_f32': (_int, [_p, _i64, _p, _i64, _i64, _p, _i64, _p]),
    'gist_gather_i32': (_int, [_p, _p, _i64, _p, _p]),
    'gist_block_gather_f32': (_int, [_p, _i64, _p, _p, _i64, _i64, _p, _i64, _p]),
    'gist_block_scatter_f32': (_int, [_p, _i64, _p, _p, _i64, _i64, _p, _i64, _p]),
    'gist_mean_rows_f32': (_int, [_p, _i64, _i64, _i64, _p, _p]),
    'gist_standard_scaler_workspace_bytes': (_i64, [_i64, _i64]),
    'gist_standard_scaler_f32': (_int, [_p, _i64, _i64, _i64, _p, _i64, _p, _p, _p, _i64, _p]),
    'gist_timer_create': (_p, [_i64]),
    'gist_timer_destroy': (None, [_p]),
    'gist_timer_reset': (None, [_p]),
    'gist_timer_count': (_i64, [_p]),
    'gist_timer_read': (_int, [_p, _i64, _p, _p, _p, _p, _p]),
    'gist_step_h3_workspace_bytes': (_i64, [_p]),
    'gist_step_h3_workspace_bytes_mode': (_i64, [_p, _int]),
    'gist_step_fused_workspace_bytes': (_i64, [_p]),
    'gist_step_col_partials_floats': (_i64, [_p]),
    'gist_step_fused_slab_bytes': (_i64, [_p, _i32]),
    'gist_spmm_csr_drop_f32': (_int, [_p, _p, _p, _i64, _p, _i64, _i64, _i64, _p, _p, _int, _p, _i64, _int, _f,
                                      _u64, _u64, _u64, _i64, _p]),
    'gist_spmm_csr_drop_prepared_f32': (_int, [_p, _p, _p, _i64, _p, _i64, _i64, _i64, _p, _p, _int, _p, _i64, _int, _f,
                                               _u64, _u64, _u64, _i64, _p, _p]),
    'gist_spmm_csr_drop_lnbwd_f32': (_int, [_p, _p, _p, _i64, _p, _i64, _i64, _i64, _p, _p, _i64, _f, _u64, _u64, _u64, _i64,
                                            _p, _i64, _p, _p, _i64, _p, _i64, _int, _p]),
    'gist_spmm_lnb_units': (_i64, [_i64]),
    'gist_spmm_block_image_bytes': (_i64, []),
    'gist_spmm_pair_min_edges': (ctypes.c_int32, []),
    'gist_spmm_block_units_f32': (_int, [_p, _i64, _p, _p, _i64, _p, _i64, _i64, _i64, _p, _int, _p]),
    'gist_spmm_block_chains_f32': (_int, [_p, _i64, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _p, _int, _p]),
    'gist_spmm_prepared_useful': (_int, [_i64, _i64, _i64, _p, _p]),
    'gist_spmm_drop_takes': (_int, [_int, _i64, _i64, _i64, _p, _p, _int]),
    'gist_gemm_dual_takes': (_int, [_i64, _i64, _i64, _i64, _i64, _i64, _i64, _p, _p, _p, _p]),
    'gist_gemm_nn_tn_dual_f32': (_int, [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _p, _i64, _p, _p]),
    'gist_gemm_slabs_f32': (_int, [_int, _p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _i64, _p, _i64, _p, _p]),
    'gist_ln_relu_fwd_drop_f32': (_int, [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _int, _int, _f, _f, _u64,
                                         _u64, _i64, _p]),
    'gist_ln_relu_fwd_slabs_f32': (_int, [_p, _i64, _p, _i64, _int, _p, _p, _i64, _p, _i64, _p, _i64, _i64, _int, _int,
                                          _f, _f, _u64, _u64, _i64, _p]),
    'gist_gemm_splits_operands': (_int, [_i64, _i64, _i64]),
    'gist_row_chunks16': (_i64, [_i64]),
    'gist_ln_relu_bwd_colsum_f32': (_int, [_p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _int, _int, _p, _p]),
    'gist_colsum_chunks_f32': (_int, [_p, _i64, _i64, _p, _p]),
    'gist_class_layer_takes': (_int, [_i64, _i64, _i64, _i64, _i64, _p, _p]),
    'gist_class_layer_f32': (_int, [_p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _f, _u64, _u64,
                                    _p, _i64, _i64, _i64, _p]),
    'gist_class_dw_slab_bytes': (_i64, [_i64, _i64, _i64]),
    'gist_class_dw_slabs_f32': (_int, [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _p]),
    'gist_ln_relu_bwd_colsum_class_dw_f32': (_int, [_p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _int, _int, _p,
                                                    _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _p]),
    'gist_gemm_nn_dropout_colsum_f32': (_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _f, _u64, _u64,
                                               _p, _i64, _p, _p]),
    'gist_softmax_xent_slabs_f32': (_int, [_p, _i64, _p, _i64, _i64, _p, _p, _p, _i64, _p, _p, _p, _i64, _i64,
                                           _i64, _p]),
    'gist_adam_segments_f32': (_int, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _i64, _p, _i64, _p, _i64, _i64,
                                      _p, _p]),
    'gist_grad_segments_finish_f32': (_int, [_p, _i64, _p, _i64, _p]),
    'gist_extract_batch_drop': (_int, [_p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _i64, _p, _p, _i64, _i64, _p,
                                       _i64, _p, _p, _p, _i64, _f, _u64, _u64, _i64, _p]),
    'gist_extract_parts_scratch_bytes': (_i64, [_i64]),
    'gist_extract_parts_supported': (_int, [_i64]),
    'gist_extract_parts_batch': (_int, [_p, _p, _p, _p, _p, _i64, _i64, _p, _p, _i32, _p, _p, _p, _p, _i64,
                                        _p, _p, _i64, _i64, _p, _i64, _p, _p, _p, _i64, _f, _u64, _u64, _i64, _p,
                                        _p]),
    'gist_sage_step': (_int, [_p, _p, _i64, _u64, _f, _f, _f, _f, _f, _i64, _int, _p]),
    'gist_sage_step_extracts_next': (_int, [_p, _i64, _int]),
    'gist_extract_parts_desc_batch': (_int, [_p, _p]),
    'gist_adam_segments_extract_f32': (_int, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _i64, _p, _i64, _p, _i64, _i64,
                                              _p, _p, _p]),
}

GIST_MAX_LAYERS = 16
TUNE = {'h3_min_gflop': 0, 'h3_min_tiles': 1, 'h3_tm': 2, 'gemm_tile': 3, 'gemm_splits': 4,
        'spmm_chunk': 5, 'spmm_split': 6, 'spmm_kernel': 7, 'b3c': 8, 'class_fused': 9, 'gemm_dual': 10, 'host_threads': 11, 'lnb_fused': 12, 'b3c_splits': 13, 'b3_tail': 14}
GIST_STEP_EXTRACT = 1
GIST_STEP_TRAIN = 2
GIST_STEP_EXTRACT_NEXT = 4
GIST_STEP_PREEXTRACTED = 8
GIST_STEP_PHASE_FORWARD = 16
GIST_STEP_PHASE_BACKWARD = 32
GIST_STEP_PHASE_OPTIMIZER = 64
GIST_STEP_DLOGITS_GIVEN = 128


class LayerDesc(ctypes.Structure):
    """struct gist_layer_desc (include/gist_hip.h)."""
    _fields_ = [('n_in', _i64), ('n_out', _i64), ('W', _p), ('b', _p), ('dW', _p), ('db', _p),
                ('Z', _p), ('ldz', _i64), ('Y', _p), ('ldy', _i64), ('rstd', _p)]


class StepPlan(ctypes.Structure):
    """struct gist_step_plan (include/gist_hip.h)."""
    _fields_ = [('n_layers', _i32), ('use_layernorm', _i32), ('p_drop', _f), ('seed', _u64),
                ('layer', LayerDesc * GIST_MAX_LAYERS),
                ('dlogits', _p), ('ldc', _i64), ('dZ', _p), ('partials', _p),
                ('row_loss', _p), ('loss', _p), ('workspace', _p), ('workspace_bytes', _i64),
                ('workspace2', _p), ('workspace2_bytes', _i64),
                ('params', _p), ('grads', _p), ('exp_avg', _p), ('exp_avg_sq', _p),
                ('n_params', _i64),
                ('g_rowptr', _p), ('g_col', _p), ('g_t_rowptr', _p), ('g_t_col', _p),
                ('feat', _p), ('ld_feat', _i64), ('labels_all', _p), ('remap', _p),
                ('rowptr', _p), ('col', _p), ('t_rowptr', _p), ('t_col', _p),
                ('col_capacity', _i64), ('norm', _p), ('labels', _p), ('timer', _p),
                ('n_max', _i64), ('feat_absmax', _f), ('h3_workspace', _p),
                ('h3_workspace_bytes', _i64), ('row_blocks', _p), ('n_row_blocks', _i64),
                ('spmm_prepared', _p), ('spmm_prepared_bytes', _i64),
                ('fuse', _i32), ('hsrc', _p * GIST_MAX_LAYERS), ('ld_hsrc', _i64 * GIST_MAX_LAYERS),
                ('col_partials', _p), ('fused_workspace', _p), ('fused_workspace_bytes', _i64),
                ('node_part', _p), ('part_slot', _p),
                ('batch_index', _i32), ('extract_scratch', _p),
                ('next_ids', _p), ('next_n', _i64), ('next_batch_index', _i32), ('next_drop_offset', _u64),
                ('feat_intra', _p), ('ld_feat_intra', _i64), ('sibling_parts', ctypes.c_int32)]


class ExtractPartsDesc(ctypes.Structure):
    """struct gist_extract_parts_desc (include/gist_hip.h)."""
    _fields_ = [('g_rowptr', _p), ('g_col', _p), ('g_t_rowptr', _p), ('g_t_col', _p),
                ('ids', _p), ('n', _i64), ('n_max', _i64),
                ('node_part', _p), ('part_slot', _p), ('batch', _i32),
                ('rowptr', _p), ('col', _p), ('t_rowptr', _p), ('t_col', _p), ('col_capacity', _i64), ('norm', _p),
                ('feat', _p), ('ld_feat', _i64), ('n_feat', _i64), ('z0', _p), ('ldz0', _i64),
                ('labels_all', _p), ('labels', _p),
                ('x0', _p), ('ldx0', _i64), ('p', _f), ('seed', _u64), ('offset', _u64), ('mask_ld', _i64),
                ('scratch', _p), ('feat_intra', _p), ('ld_intra', _i64), ('ah', _p)]


class GradSegment(ctypes.Structure):
    """struct gist_grad_segment (include/gist_hip.h)."""
    _fields_ = [('begin', _i64), ('end', _i64), ('src', _p), ('stride', _i64), ('n_src', _i32)]


_lib = None


def load():
    """Load libgist_hip.so once; raise GistLibraryError if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GistLibraryError(
            'gist_amd: %s is missing. Build it with `python gist_amd/build.py` '
            '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
    # torch bundles its own libamdhip64.so.7; it must be the HIP runtime of this process
    # (buffers and streams come from torch), so make sure it is mapped before our library
    # resolves the same soname -- loading ours first would pull in /opt/rocm's runtime and
    # leave the process with a runtime torch's HSA layer cannot drive.
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise GistLibraryError('gist_amd: cannot load %s: %s' % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise GistLibraryError('gist_amd: %s does not export %s (stale build?)'
                                   % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if lib.gist_abi_version() != ABI_VERSION:
        raise GistLibraryError('gist_amd: ABI version mismatch: library %d, binding %d'
                               % (lib.gist_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, name):
    if rc != 0:
        msg = load().gist_last_error()
        raise GistError('%s failed (%d): %s' % (name, rc, msg.decode() if msg else ''))
