"""Tensor-level wrappers over the C ABI (include/gist_hip.h).

torch is used for device memory and streams only: every function takes tensors
that already live in HBM, checks dtype / device / layout (the TORCH_CHECK role)
and passes raw pointers to libgist_hip.so on torch's current HIP stream.
CPU tensors are rejected -- there is no fallback path.
"""
import torch

from . import _lib

LN_EPS = 1e-5


def _stream():
    return torch.cuda.current_stream().cuda_stream


# -- optional per-kernel timing with HIP events on the launch stream (bench.py) ------
_prof = None


def profile_begin():
    """Start recording (start, end) events around every SpMM / GEMM launch."""
    global _prof
    _prof = {'gemm': [], 'spmm': []}


def profile_end():
    """Stop recording; returns {'gemm': [(ms, meta)], 'spmm': [(ms, meta)]} (synchronises)."""
    global _prof
    rec, _prof = _prof, None
    torch.cuda.synchronize()
    return {k: [(a.elapsed_time(b), meta) for (a, b, meta) in v] for k, v in rec.items()}


class _Timed(object):
    def __init__(self, kind, meta):
        self.kind, self.meta = kind, meta

    def __enter__(self):
        if _prof is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if _prof is not None:
            self.b.record()
            _prof[self.kind].append((self.a, self.b, self.meta))
        return False


def _dev(t, name, dtype):
    if not torch.is_tensor(t):
        raise TypeError('%s must be a tensor' % name)
    if not t.is_cuda:
        raise RuntimeError('gist_amd: %s must live on the GPU (got %s); the HIP path has no '
                           'CPU fallback' % (name, t.device))
    if t.dtype != dtype:
        raise TypeError('gist_amd: %s must be %s (got %s)' % (name, dtype, t.dtype))
    return t


def _mat(t, name):
    """2-D fp32 with unit inner stride -> (ptr, ld)."""
    _dev(t, name, torch.float32)
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError('gist_amd: %s must be 2-D with contiguous rows' % name)
    ld = t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])
    return t.data_ptr(), ld


def _vec(t, name, dtype, n=None):
    _dev(t, name, dtype)
    if not t.is_contiguous():
        raise ValueError('gist_amd: %s must be contiguous' % name)
    if n is not None and t.numel() < n:
        raise ValueError('gist_amd: %s too small (%d < %d)' % (name, t.numel(), n))
    return t.data_ptr()


def _opt(t, name, dtype, n=None):
    return None if t is None else _vec(t, name, dtype, n)


def device_count():
    return _lib.load().gist_device_count()


# -- aggregation ---------------------------------------------------------------
def in_degree_norm(rowptr, out=None):
    L = _lib.load()
    n = rowptr.numel() - 1
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=rowptr.device)
    _lib.check(L.gist_in_degree_norm_f32(_vec(rowptr, 'rowptr', torch.int32), n,
                                         _vec(out, 'norm', torch.float32, n), _stream()),
               'gist_in_degree_norm_f32')
    return out


def spmm_prepare(rowptr, col, row_blocks=None):
    """The block structure of a row set for spmm(..., prepared=...): built once for every
    aggregation over the same graph and blocks (gist_spmm_blocks_prepare).  Returns a uint8 tensor."""
    L = _lib.load()
    n = rowptr.numel() - 1
    nb = (n + 127) // 128 if row_blocks is None else row_blocks.numel() - 1
    buf = torch.empty(max(int(L.gist_spmm_blocks_bytes(nb)), 16), dtype=torch.uint8, device=rowptr.device)
    _lib.check(L.gist_spmm_blocks_prepare(_vec(rowptr, 'rowptr', torch.int32), _vec(col, 'col', torch.int32),
                                          n, _opt(row_blocks, 'row_blocks', torch.int32),
                                          0 if row_blocks is None else nb, buf.data_ptr(), buf.numel(),
                                          _stream()), 'gist_spmm_blocks_prepare')
    return buf


def spmm(rowptr, col, x, y, out_scale=None, src_scale=None, accumulate=False, row_blocks=None,
         blocked=False, prepared=None):
    """y[v] (+)= out_scale[v] * sum_e src_scale[col[e]] * x[col[e]]; see gist_spmm_csr_f32.
    row_blocks (int32 [n_blocks+1]: locality blocks of the rows, <= 128 rows each) or blocked=True
    (uniform 128-row blocks) selects the blocked kernels (gist_spmm_csr_blocked_f32); prepared =
    spmm_prepare(rowptr, col, row_blocks) reuses the block structure (gist_spmm_csr_prepared_f32)."""
    L = _lib.load()
    n = rowptr.numel() - 1
    xp, ldx = _mat(x, 'x')
    yp, ldy = _mat(y, 'y')
    d = x.shape[1]
    if y.shape[0] != n or y.shape[1] != d:
        raise ValueError('gist_amd: spmm output shape %s != (%d, %d)' % (tuple(y.shape), n, d))
    common = (_vec(rowptr, 'rowptr', torch.int32), _vec(col, 'col', torch.int32), xp, ldx, yp, ldy, n,
              d, _opt(out_scale, 'out_scale', torch.float32, n),
              _opt(src_scale, 'src_scale', torch.float32, x.shape[0]), int(bool(accumulate)))
    with _Timed('spmm', (n, x.shape[0], d)):
        if row_blocks is not None or blocked:
            if x.shape[0] != n:
                raise ValueError('gist_amd: the blocked spmm needs a square row/source set')
            nb = 0 if row_blocks is None else row_blocks.numel() - 1
            if prepared is not None:
                rc = L.gist_spmm_csr_prepared_f32(*common, _opt(row_blocks, 'row_blocks', torch.int32),
                                                  nb, prepared.data_ptr(), _stream())
                name = 'gist_spmm_csr_prepared_f32'
            else:
                rc = L.gist_spmm_csr_blocked_f32(*common, _opt(row_blocks, 'row_blocks', torch.int32),
                                                 nb, _stream())
                name = 'gist_spmm_csr_blocked_f32'
        else:
            rc = L.gist_spmm_csr_f32(*common, _stream())
            name = 'gist_spmm_csr_f32'
    _lib.check(rc, name)
    return y


def standard_scale_(feat, fit_rows=None):
    """sklearn StandardScaler fit on rows `fit_rows` (int32 device tensor; None = all rows) and
    applied IN PLACE to every row of `feat` [N, F] (cluster_gcn_ist_distrib.py:492-499).
    Returns (mean, var) as float64 device tensors."""
    L = _lib.load()
    xp, ld = _mat(feat, 'feat')
    n, d = feat.shape
    n_fit = n if fit_rows is None else fit_rows.numel()
    mean = torch.empty(d, dtype=torch.float64, device=feat.device)
    var = torch.empty(d, dtype=torch.float64, device=feat.device)
    need = L.gist_standard_scaler_workspace_bytes(n_fit, d)
    ws = torch.empty(max(int(need), 8), dtype=torch.uint8, device=feat.device)
    _lib.check(L.gist_standard_scaler_f32(xp, ld, n, d, _opt(fit_rows, 'fit_rows', torch.int32), n_fit,
                                          mean.data_ptr(), var.data_ptr(), ws.data_ptr(), ws.numel(),
                                          _stream()), 'gist_standard_scaler_f32')
    return mean, var


# -- projection ------------------------------------------------------------------
_ws = {}


def workspace(nbytes, device):
    """Grow-only per-device scratch for split-K partial sums."""
    key = (device.type, device.index)
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = w
    return w


def _ws_for(m, n, k, device):
    L = _lib.load()
    need = L.gist_gemm_workspace_bytes(m, n, k)
    if need == 0:
        return None, 0
    w = workspace(need, device)
    return w.data_ptr(), w.numel()


def gemm_mode(mode=None):
    """Get (and with an argument, set) how large projections form their products
    (include/gist_hip.h gist_gemm_set_mode): 'bf16x3' (default) = three bf16 pieces per fp32 operand
    (all 24 bits), six cross terms on the bf16 matrix cores, fp32-MFMA-level error; 'f32' =
    v_mfma_f32_32x32x2_f32 everywhere, fp32 products like the reference; 'f16x3' (opt-in) = two f16
    halves per operand (22 bits).  Small shapes are fp32 MFMA in every mode.  Process-wide."""
    L = _lib.load()
    if mode is not None:
        code = {'f32': 0, 'f16x3': 1, 'bf16x3': 2, 0: 0, 1: 1, 2: 2}.get(mode)
        if code is None:
            raise ValueError("gist_amd: gemm mode must be 'f32', 'f16x3' or 'bf16x3'")
        _lib.check(L.gist_gemm_set_mode(code), 'gist_gemm_set_mode')
    return ('f32', 'f16x3', 'bf16x3')[L.gist_gemm_get_mode()]


def gemm_splits_own_operands(m, n, k):
    """Does a projection of this shape take a pre-split kernel in the current GEMM mode (gist_gemm_splits_operands)?"""
    return bool(_lib.load().gist_gemm_splits_operands(int(m), int(n), int(k)))


def tuning(knob, value=None):
    """Get (and with a value, set) a tuning hook (include/gist_hip.h gist_tuning_set): explicit
    overrides of the launchers' choices for sweeps and tests; 0 = the launcher decides."""
    L = _lib.load()
    code = _lib.TUNE[knob]
    if value is not None:
        _lib.check(L.gist_tuning_set(code, float(value)), 'gist_tuning_set')
    return L.gist_tuning_get(code)


def gemm_nt(a, w, bias, y):
    """y = a @ w.T + bias"""
    L = _lib.load()
    ap, lda = _mat(a, 'a')
    wp, ldw = _mat(w, 'w')
    yp, ldy = _mat(y, 'y')
    m, k = a.shape
    n = w.shape[0]
    if w.shape[1] != k or tuple(y.shape) != (m, n):
        raise ValueError('gist_amd: gemm_nt shape mismatch')
    wsp, wsb = _ws_for(m, n, k, a.device)
    with _Timed('gemm', ('nt', m, n, k)):
        rc = L.gist_gemm_nt_f32(ap, lda, wp, ldw, _opt(bias, 'bias', torch.float32, n), yp, ldy,
                                m, n, k, wsp, wsb, _stream())
    _lib.check(rc, 'gist_gemm_nt_f32')
    return y


def gemm_nn(g, w, z):
    """z = g @ w"""
    L = _lib.load()
    gp, ldg = _mat(g, 'g')
    wp, ldw = _mat(w, 'w')
    zp, ldz = _mat(z, 'z')
    m, k = g.shape
    n = w.shape[1]
    if w.shape[0] != k or tuple(z.shape) != (m, n):
        raise ValueError('gist_amd: gemm_nn shape mismatch')
    wsp, wsb = _ws_for(m, n, k, g.device)
    with _Timed('gemm', ('nn', m, n, k)):
        rc = L.gist_gemm_nn_f32(gp, ldg, wp, ldw, zp, ldz, m, n, k, wsp, wsb, _stream())
    _lib.check(rc, 'gist_gemm_nn_f32')
    return z


def gemm_tn(g, a, d):
    """d = g.T @ a"""
    L = _lib.load()
    gp, ldg = _mat(g, 'g')
    ap, lda = _mat(a, 'a')
    dp, ldd = _mat(d, 'd')
    k, m = g.shape
    n = a.shape[1]
    if a.shape[0] != k or tuple(d.shape) != (m, n):
        raise ValueError('gist_amd: gemm_tn shape mismatch')
    wsp, wsb = _ws_for(m, n, k, g.device)
    with _Timed('gemm', ('tn', m, n, k)):
        rc = L.gist_gemm_tn_f32(gp, ldg, ap, lda, dp, ldd, m, n, k, wsp, wsb, _stream())
    _lib.check(rc, 'gist_gemm_tn_f32')
    return d


# -- row epilogues -----------------------------------------------------------------
def ln_relu_fwd(y, out, rstd, use_lynorm, relu, eps=LN_EPS):
    L = _lib.load()
    yp, ldy = _mat(y, 'y')
    op, ldo = _mat(out, 'out')
    n, d = y.shape
    if tuple(out.shape) != (n, d):
        raise ValueError('gist_amd: ln_relu_fwd shape mismatch')
    _lib.check(L.gist_ln_relu_fwd_f32(yp, ldy, op, ldo, _opt(rstd, 'rstd', torch.float32, n), n, d,
                                      int(bool(use_lynorm)), int(bool(relu)), eps, _stream()),
               'gist_ln_relu_fwd_f32')
    return out


def ln_relu_bwd(d_out, yhat, rstd, dy, use_lynorm, relu):
    L = _lib.load()
    gp, ldg = _mat(d_out, 'd_out')
    yp, ldy = _mat(yhat, 'yhat')
    dp, ldd = _mat(dy, 'dy')
    n, d = yhat.shape
    if tuple(d_out.shape) != (n, d) or tuple(dy.shape) != (n, d):
        raise ValueError('gist_amd: ln_relu_bwd shape mismatch')
    _lib.check(L.gist_ln_relu_bwd_f32(gp, ldg, yp, ldy, _opt(rstd, 'rstd', torch.float32, n), dp,
                                      ldd, n, d, int(bool(use_lynorm)), int(bool(relu)), _stream()),
               'gist_ln_relu_bwd_f32')
    return dy


def dropout_(z, p, seed, offset):
    L = _lib.load()
    zp, ldz = _mat(z, 'z')
    _lib.check(L.gist_dropout_f32(zp, ldz, z.shape[0], z.shape[1], float(p), int(seed),
                                  int(offset), _stream()), 'gist_dropout_f32')
    return z


def gemm_nn_dropout_(g, w, z, p, seed, offset):
    """z = dropout(g @ w) with gist_dropout_f32's mask stream (one kernel when g is narrow)."""
    L = _lib.load()
    gp, ldg = _mat(g, 'g')
    wp, ldw = _mat(w, 'w')
    zp, ldz = _mat(z, 'z')
    m, k = g.shape
    n = w.shape[1]
    if w.shape[0] != k or tuple(z.shape) != (m, n):
        raise ValueError('gist_amd: gemm_nn_dropout_ shape mismatch')
    wsp, wsb = _ws_for(m, n, k, g.device)
    with _Timed('gemm', ('nn', m, n, k)):
        rc = L.gist_gemm_nn_dropout_f32(gp, ldg, wp, ldw, zp, ldz, m, n, k, float(p), int(seed),
                                        int(offset), wsp, wsb, _stream())
    _lib.check(rc, 'gist_gemm_nn_dropout_f32')
    return z


def colsum(g, out, partials=None):
    L = _lib.load()
    gp, ldg = _mat(g, 'g')
    n, d = g.shape
    chunks = L.gist_colsum_partials(n)
    if partials is None:
        partials = torch.empty(max(chunks * d, 1), dtype=torch.float32, device=g.device)
    _lib.check(L.gist_colsum_f32(gp, ldg, n, d, _vec(partials, 'partials', torch.float32, chunks * d),
                                 _vec(out, 'out', torch.float32, d), _stream()), 'gist_colsum_f32')
    return out


# -- loss / optimiser ------------------------------------------------------------------
def softmax_xent(logits, labels, mask, count, row_loss, loss, d_logits):
    L = _lib.load()
    lp, ldl = _mat(logits, 'logits')
    gp, ldg = _mat(d_logits, 'd_logits')
    n, c = logits.shape
    _lib.check(L.gist_softmax_xent_f32(lp, ldl, _vec(labels, 'labels', torch.int32, n),
                                       _opt(mask, 'mask', torch.uint8, n), int(count),
                                       _vec(row_loss, 'row_loss', torch.float32, n),
                                       _vec(loss, 'loss', torch.float32, 1), gp, ldg, n, c,
                                       _stream()), 'gist_softmax_xent_f32')
    return loss


def adam_(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8,
          weight_decay=0.0):
    L = _lib.load()
    n = param.numel()
    _lib.check(L.gist_adam_f32(_vec(param, 'param', torch.float32), _vec(grad, 'grad', torch.float32, n),
                               _vec(exp_avg, 'exp_avg', torch.float32, n),
                               _vec(exp_avg_sq, 'exp_avg_sq', torch.float32, n), n, lr, beta1,
                               beta2, eps, weight_decay, int(step), _stream()), 'gist_adam_f32')
    return param


def argmax_correct(logits, labels, mask, correct):
    L = _lib.load()
    lp, ldl = _mat(logits, 'logits')
    n, c = logits.shape
    _lib.check(L.gist_argmax_correct_i32(lp, ldl, _vec(labels, 'labels', torch.int32, n),
                                         _opt(mask, 'mask', torch.uint8, n),
                                         _vec(correct, 'correct', torch.int32, 1), n, c, _stream()),
               'gist_argmax_correct_i32')
    return correct


# -- cluster batch extraction -------------------------------------------------------------
def copy_i32_raw(src_ptr, dst_ptr, n):
    """dst[i] = src[i] by a kernel (gist_copy_i32); raw addresses: either side may be pinned host memory."""
    _lib.check(_lib.load().gist_copy_i32(src_ptr, dst_ptr, int(n), _stream()), 'gist_copy_i32')


def publish_i64_raw(device_word_ptr, tag, host_word_ptr):
    """host_word[0] = *device_word, host_word[1] = tag, by a kernel on the current stream (gist_publish_i64)."""
    _lib.check(_lib.load().gist_publish_i64(device_word_ptr, int(tag), host_word_ptr, _stream()), 'gist_publish_i64')


def fill_i32_(t, value):
    L = _lib.load()
    _lib.check(L.gist_fill_i32(_vec(t, 't', torch.int32), t.numel(), int(value), _stream()),
               'gist_fill_i32')
    return t


def induced_mark(ids, remap, unmark=False):
    L = _lib.load()
    f = L.gist_induced_unmark if unmark else L.gist_induced_mark
    _lib.check(f(_vec(ids, 'ids', torch.int32), ids.numel(), _vec(remap, 'remap', torch.int32),
                 _stream()), 'gist_induced_mark')


def induced_rowptr(rowptr, col, ids, remap, sub_rowptr):
    L = _lib.load()
    n = ids.numel()
    _lib.check(L.gist_induced_rowptr(_vec(rowptr, 'rowptr', torch.int32), _vec(col, 'col', torch.int32),
                                     _vec(ids, 'ids', torch.int32), n,
                                     _vec(remap, 'remap', torch.int32),
                                     _vec(sub_rowptr, 'sub_rowptr', torch.int32, n + 1), _stream()),
               'gist_induced_rowptr')
    return sub_rowptr


def induced_fill(rowptr, col, ids, remap, sub_rowptr, sub_col):
    L = _lib.load()
    n = ids.numel()
    _lib.check(L.gist_induced_fill(_vec(rowptr, 'rowptr', torch.int32), _vec(col, 'col', torch.int32),
                                   _vec(ids, 'ids', torch.int32), n, _vec(remap, 'remap', torch.int32),
                                   _vec(sub_rowptr, 'sub_rowptr', torch.int32, n + 1),
                                   _vec(sub_col, 'sub_col', torch.int32), sub_col.numel(), _stream()),
               'gist_induced_fill')
    return sub_col


def extract_batch(g, ids, remap, rowptr, col, t_rowptr, t_col, norm, feat, z0_left, labels_all,
                  labels):
    """Fused cluster-batch extraction (gist_extract_batch): induced in-/out-edge CSR, norm,
    feature + label gather, remap reset."""
    L = _lib.load()
    n = ids.numel()
    fp, ldf = _mat(feat, 'feat')
    zp, ldz = _mat(z0_left, 'z0')
    if tuple(z0_left.shape) != (n, feat.shape[1]):
        raise ValueError('gist_amd: extract_batch z0 shape mismatch')
    if col.numel() != t_col.numel():
        raise ValueError('gist_amd: extract_batch needs equal col capacities')
    _lib.check(L.gist_extract_batch(
        _vec(g.rowptr, 'g.rowptr', torch.int32), _vec(g.col, 'g.col', torch.int32),
        _vec(g.t_rowptr, 'g.t_rowptr', torch.int32), _vec(g.t_col, 'g.t_col', torch.int32),
        _vec(ids, 'ids', torch.int32), n, _vec(remap, 'remap', torch.int32),
        _vec(rowptr, 'rowptr', torch.int32, n + 1), _vec(col, 'col', torch.int32),
        _vec(t_rowptr, 't_rowptr', torch.int32, n + 1), _vec(t_col, 't_col', torch.int32),
        col.numel(), _vec(norm, 'norm', torch.float32, n), fp, ldf, feat.shape[1], zp, ldz,
        _opt(labels_all, 'labels_all', torch.int32), _opt(labels, 'labels', torch.int32, n),
        _stream()), 'gist_extract_batch')


def gather_rows(src, ids, dst):
    L = _lib.load()
    sp, lds = _mat(src, 'src')
    dp, ldd = _mat(dst, 'dst')
    n, d = ids.numel(), src.shape[1]
    if tuple(dst.shape) != (n, d):
        raise ValueError('gist_amd: gather_rows shape mismatch')
    _lib.check(L.gist_gather_rows_f32(sp, lds, _vec(ids, 'ids', torch.int32), n, d, dp, ldd,
                                      _stream()), 'gist_gather_rows_f32')
    return dst


def gather_i32(src, ids, dst):
    L = _lib.load()
    n = ids.numel()
    _lib.check(L.gist_gather_i32(_vec(src, 'src', torch.int32), _vec(ids, 'ids', torch.int32), n,
                                 _vec(dst, 'dst', torch.int32, n), _stream()), 'gist_gather_i32')
    return dst


# -- IST weight blocks ------------------------------------------------------------------------
def block_gather(src, row_idx, col_idx, dst):
    """dst[i, j] = src[row_idx[i], col_idx[j]] (None = identity)."""
    L = _lib.load()
    sp, lds = _mat(src, 'src')
    dp, ldd = _mat(dst, 'dst')
    nr, nc = dst.shape
    _lib.check(L.gist_block_gather_f32(sp, lds, _opt(row_idx, 'row_idx', torch.int32, nr),
                                       _opt(col_idx, 'col_idx', torch.int32, nc), nr, nc, dp, ldd,
                                       _stream()), 'gist_block_gather_f32')
    return dst


def block_scatter(src, row_idx, col_idx, dst):
    """dst[row_idx[i], col_idx[j]] = src[i, j] (None = identity)."""
    L = _lib.load()
    sp, lds = _mat(src, 'src')
    dp, ldd = _mat(dst, 'dst')
    nr, nc = src.shape
    _lib.check(L.gist_block_scatter_f32(sp, lds, _opt(row_idx, 'row_idx', torch.int32, nr),
                                        _opt(col_idx, 'col_idx', torch.int32, nc), nr, nc, dp, ldd,
                                        _stream()), 'gist_block_scatter_f32')
    return dst


def mean_rows(src, stride, n_src, n, out):
    L = _lib.load()
    _lib.check(L.gist_mean_rows_f32(_vec(src, 'src', torch.float32), stride, n_src, n,
                                    _vec(out, 'out', torch.float32, n), _stream()),
               'gist_mean_rows_f32')
    return out


# -- fused forms (include/gist_hip.h: dropout folded into producers / consumers, deferred reductions) --
def spmm_drop_takes(mode, d, x, y, has_row_blocks):
    L = _lib.load()
    xp, ldx = _mat(x, 'x')
    yp, ldy = _mat(y, 'y')
    return bool(L.gist_spmm_drop_takes(int(mode), d, ldx, ldy, xp, yp, int(bool(has_row_blocks))))


def spmm_block_units(units, images, x, y, out_scale=None, accumulate=False):
    """gist_spmm_block_units_f32: per unit (r0, r1, xs0, xs1) y[r0:r1] (+)= scale * C_u @ x[xs0:xs1] on the bf16x3
    matrix cores; units of one call have disjoint output rows."""
    L = _lib.load()
    xp, ldx = _mat(x, 'x')
    yp, ldy = _mat(y, 'y')
    d = x.shape[1]
    nu = units.shape[0]
    with _Timed('spmm', (y.shape[0], x.shape[0], d)):
        rc = L.gist_spmm_block_units_f32(_vec(units, 'units', torch.int32, 4 * nu), nu, images.data_ptr(), xp, ldx, yp, ldy,
                                         y.shape[0], d, _opt(out_scale, 'out_scale', torch.float32), int(bool(accumulate)),
                                         _stream())
    _lib.check(rc, 'gist_spmm_block_units_f32')
    return y


def spmm_block_chains(chain_ptr, units, images, x, y, out_scale=None, accumulate=False):
    """gist_spmm_block_chains_f32: chain c = units [chain_ptr[c], chain_ptr[c + 1]) (same output rows); one workgroup per
    (chain, column-tile group) keeps the accumulators across the chain: y written once."""
    L = _lib.load()
    xp, ldx = _mat(x, 'x')
    yp, ldy = _mat(y, 'y')
    d = x.shape[1]
    nc = chain_ptr.numel() - 1
    nu = units.shape[0]
    with _Timed('spmm', (y.shape[0], x.shape[0], d)):
        rc = L.gist_spmm_block_chains_f32(_vec(chain_ptr, 'chain_ptr', torch.int32, nc + 1), nc,
                                          _vec(units, 'units', torch.int32, 4 * nu), images.data_ptr(), xp, ldx, yp, ldy,
                                          y.shape[0], d, _opt(out_scale, 'out_scale', torch.float32), int(bool(accumulate)),
                                          _stream())
    _lib.check(rc, 'gist_spmm_block_chains_f32')
    return y


def spmm_prepared_useful(x, y):
    """Does an aggregation of x's width read a prepared block structure (gist_spmm_prepared_useful)?"""
    L = _lib.load()
    xp, ldx = _mat(x, 'x')
    yp, ldy = _mat(y, 'y')
    return bool(L.gist_spmm_prepared_useful(x.shape[1], ldx, ldy, xp, yp))


def spmm_drop(rowptr, col, x, y, mode, p, seed, y_offset, src_offset, mask_ld, out_scale=None,
              src_scale=None, accumulate=False, row_blocks=None, prepared=None):
    """spmm with gist_dropout_f32's mask folded in (gist_spmm_csr_drop_f32 / _prepared_f32)."""
    if prepared is not None:
        L = _lib.load()
        n = rowptr.numel() - 1
        xp, ldx = _mat(x, 'x')
        yp, ldy = _mat(y, 'y')
        d = x.shape[1]
        nb = 0 if row_blocks is None else row_blocks.numel() - 1
        with _Timed('spmm', (n, x.shape[0], d)):
            rc = L.gist_spmm_csr_drop_prepared_f32(
                _vec(rowptr, 'rowptr', torch.int32), _vec(col, 'col', torch.int32), xp, ldx, yp, ldy, n, d,
                _opt(out_scale, 'out_scale', torch.float32, n), _opt(src_scale, 'src_scale', torch.float32, x.shape[0]),
                int(bool(accumulate)), _opt(row_blocks, 'row_blocks', torch.int32), nb, int(mode), float(p), int(seed),
                int(y_offset), int(src_offset), int(mask_ld), prepared.data_ptr(), _stream())
        _lib.check(rc, 'gist_spmm_csr_drop_prepared_f32')
        return y
    L = _lib.load()
    n = rowptr.numel() - 1
    xp, ldx = _mat(x, 'x')
    yp, ldy = _mat(y, 'y')
    d = x.shape[1]
    nb = 0 if row_blocks is None else row_blocks.numel() - 1
    with _Timed('spmm', (n, x.shape[0], d)):
        rc = L.gist_spmm_csr_drop_f32(_vec(rowptr, 'rowptr', torch.int32), _vec(col, 'col', torch.int32), xp,
                                      ldx, yp, ldy, n, d, _opt(out_scale, 'out_scale', torch.float32, n),
                                      _opt(src_scale, 'src_scale', torch.float32, x.shape[0]),
                                      int(bool(accumulate)), _opt(row_blocks, 'row_blocks', torch.int32), nb,
                                      int(mode), float(p), int(seed), int(y_offset), int(src_offset),
                                      int(mask_ld), _stream())
    _lib.check(rc, 'gist_spmm_csr_drop_f32')
    return y


def spmm_lnb_units(n_row_blocks):
    return int(_lib.load().gist_spmm_lnb_units(int(n_row_blocks)))


def spmm_drop_lnbwd(rowptr, col, x, y, p, seed, y_offset, src_offset, mask_ld, yhat, dy, col_partials, src_scale=None,
                    row_blocks=None, rstd=None, relu=True):
    """gist_spmm_csr_drop_lnbwd_f32: d_out = mask(y) + A . (src_scale . mask(x)); dy = LayerNorm + ReLU backward of d_out
    (yhat, rstd); col_partials [spmm_lnb_units][d] = column sums of dy per workgroup."""
    L = _lib.load()
    n = rowptr.numel() - 1
    xp, ldx = _mat(x, 'x')
    yp, ldy = _mat(y, 'y')
    d = x.shape[1]
    hp, ldh = _mat(yhat, 'yhat')
    dp, lddy = _mat(dy, 'dy')
    cp, ldc = _mat(col_partials, 'col_partials')
    if ldc != d:
        raise ValueError('col_partials must be contiguous [rows][d]')
    nb = 0 if row_blocks is None else row_blocks.numel() - 1
    with _Timed('spmm', (n, x.shape[0], d)):
        rc = L.gist_spmm_csr_drop_lnbwd_f32(
            _vec(rowptr, 'rowptr', torch.int32), _vec(col, 'col', torch.int32), xp, ldx, yp, ldy, n, d,
            _opt(src_scale, 'src_scale', torch.float32, x.shape[0]), _opt(row_blocks, 'row_blocks', torch.int32), nb,
            float(p), int(seed), int(y_offset), int(src_offset), int(mask_ld), hp, ldh,
            _opt(rstd, 'rstd', torch.float32, n), dp, lddy, cp, int(col_partials.shape[0]), int(bool(relu)), _stream())
    _lib.check(rc, 'gist_spmm_csr_drop_lnbwd_f32')
    return dy


def gemm_dual_takes(dy, w, z, dz):
    """Does gist_gemm_nn_tn_dual_f32 take this hidden layer's backward (dy [m, k], w [k, n], z [m, n], dz [m, n])?"""
    L = _lib.load()
    yp, lddy = _mat(dy, 'dy')
    wp, ldw = _mat(w, 'w')
    zp, ldz = _mat(z, 'z')
    dp, lddz = _mat(dz, 'dz')
    return bool(L.gist_gemm_dual_takes(dy.shape[0], w.shape[1], dy.shape[1], lddy, ldw, ldz, lddz, yp, wp, zp, dp))


def gemm_nn_tn_dual(dy, w, dz, z, dw, slabs):
    """gist_gemm_nn_tn_dual_f32: dz = dy @ w and dW = dy.T @ z (slabs) in one launch; returns the slab count."""
    import ctypes
    L = _lib.load()
    yp, lddy = _mat(dy, 'dy')
    wp, ldw = _mat(w, 'w')
    zp, ldz = _mat(z, 'z')
    dp, lddz = _mat(dz, 'dz')
    gp, lddw = _mat(dw, 'dw')
    m, k = dy.shape
    n = w.shape[1]
    ns = ctypes.c_int32(0)
    with _Timed('gemm', ('nn+tn', m, 2 * n, k)):
        rc = L.gist_gemm_nn_tn_dual_f32(yp, lddy, wp, ldw, dp, lddz, zp, ldz, gp, lddw, m, n, k,
                                        slabs.data_ptr() if slabs is not None else None,
                                        slabs.numel() * slabs.element_size() if slabs is not None else 0,
                                        ctypes.byref(ns), _stream())
    _lib.check(rc, 'gist_gemm_nn_tn_dual_f32')
    return int(ns.value)


def gemm_slabs(layout, a, b, bias, c, slabs):
    """gist_gemm_slabs_f32: layout 'nt' (c = a @ b.T + bias), 'nn' (c = a @ b), 'tn' (c = a.T @ b).
    Returns the slab count (1: c is final)."""
    import ctypes
    L = _lib.load()
    code = {'nt': 0, 'nn': 1, 'tn': 2}[layout]
    ap, lda = _mat(a, 'a')
    bp, ldb = _mat(b, 'b')
    cp, ldc = _mat(c, 'c')
    m, n = c.shape
    k = a.shape[1] if code != 2 else a.shape[0]
    ns = ctypes.c_int32(1)
    with _Timed('gemm', (layout, m, n, k)):
        rc = L.gist_gemm_slabs_f32(code, ap, lda, bp, ldb, _opt(bias, 'bias', torch.float32, n), cp, ldc, m, n,
                                   k, slabs.data_ptr() if slabs is not None else None,
                                   slabs.numel() * slabs.element_size() if slabs is not None else 0,
                                   ctypes.byref(ns), _stream())
    _lib.check(rc, 'gist_gemm_slabs_f32')
    return ns.value


def ln_relu_fwd_drop(y, out, out2, rstd, use_lynorm, relu, p, seed, offset, mask_ld, eps=LN_EPS):
    L = _lib.load()
    yp, ldy = _mat(y, 'y')
    op, ldo = _mat(out, 'out')
    o2p, ldo2 = _mat(out2, 'out2') if out2 is not None else (None, 0)
    n, d = y.shape
    _lib.check(L.gist_ln_relu_fwd_drop_f32(yp, ldy, op, ldo, o2p, ldo2, _opt(rstd, 'rstd', torch.float32, n), n,
                                           d, int(bool(use_lynorm)), int(bool(relu)), eps, float(p), int(seed),
                                           int(offset), int(mask_ld), _stream()), 'gist_ln_relu_fwd_drop_f32')
    return out


def ln_relu_fwd_slabs(y, slabs, n_slabs, bias, out, out2, rstd, use_lynorm, relu, p=0.0, seed=0, offset=0, mask_ld=0,
                      eps=LN_EPS):
    """LayerNorm + ReLU of y = sum of the projection's n_slabs split-K slabs (dense [n, d] arrays at the start of
    `slabs`, n * d floats apart) + bias, formed inside the kernel; out / out2 / dropout as ln_relu_fwd_drop."""
    L = _lib.load()
    yp, ldy = _mat(y, 'y')
    op, ldo = _mat(out, 'out')
    o2p, ldo2 = _mat(out2, 'out2') if out2 is not None else (None, 0)
    n, d = y.shape
    if n_slabs > 0:
        if slabs is None or not slabs.is_cuda or slabs.numel() * slabs.element_size() < n_slabs * n * d * 4:
            raise ValueError('gist_amd: slabs must be a device buffer of at least n_slabs * n * d floats')
    _lib.check(L.gist_ln_relu_fwd_slabs_f32(yp, ldy, slabs.data_ptr() if n_slabs > 0 else None, n * d, int(n_slabs),
                                            _opt(bias, 'bias', torch.float32, d), op, ldo, o2p, ldo2,
                                            _opt(rstd, 'rstd', torch.float32, n), n, d, int(bool(use_lynorm)),
                                            int(bool(relu)), eps, float(p), int(seed), int(offset), int(mask_ld),
                                            _stream()), 'gist_ln_relu_fwd_slabs_f32')
    return out


def ln_relu_bwd_colsum(d_out, yhat, rstd, dy, use_lynorm, relu, col_partials):
    L = _lib.load()
    gp, ldg = _mat(d_out, 'd_out')
    yp, ldy = _mat(yhat, 'yhat')
    dp, ldd = _mat(dy, 'dy')
    n, d = yhat.shape
    _lib.check(L.gist_ln_relu_bwd_colsum_f32(gp, ldg, yp, ldy, _opt(rstd, 'rstd', torch.float32, n), dp, ldd, n,
                                             d, int(bool(use_lynorm)), int(bool(relu)),
                                             _vec(col_partials, 'col_partials', torch.float32,
                                                  L.gist_row_chunks16(n) * d), _stream()),
               'gist_ln_relu_bwd_colsum_f32')
    return dy


def colsum_chunks(partials, chunks, d, out):
    L = _lib.load()
    _lib.check(L.gist_colsum_chunks_f32(_vec(partials, 'partials', torch.float32, chunks * d), chunks, d,
                                        _vec(out, 'out', torch.float32, d), _stream()), 'gist_colsum_chunks_f32')
    return out


def gemm_nn_dropout_colsum_(g, w, z, p, seed, offset, g_col_partials):
    L = _lib.load()
    gp, ldg = _mat(g, 'g')
    wp, ldw = _mat(w, 'w')
    zp, ldz = _mat(z, 'z')
    m, k = g.shape
    n = w.shape[1]
    wsp, wsb = _ws_for(m, n, k, g.device)
    with _Timed('gemm', ('nn', m, n, k)):
        rc = L.gist_gemm_nn_dropout_colsum_f32(gp, ldg, wp, ldw, zp, ldz, m, n, k, float(p), int(seed),
                                               int(offset), wsp, wsb,
                                               _vec(g_col_partials, 'g_col_partials', torch.float32,
                                                    L.gist_row_chunks16(m) * k), _stream())
    _lib.check(rc, 'gist_gemm_nn_dropout_colsum_f32')
    return z


def class_layer_takes(z, w, n_classes):
    """Does gist_class_layer_f32 take this class layer (z [n, k] its dropped input, w [C, k])?"""
    L = _lib.load()
    zp, ldz = _mat(z, 'z')
    wp, ldw = _mat(w, 'w')
    return bool(L.gist_class_layer_takes(z.shape[0], int(n_classes), z.shape[1], ldz, ldw, zp, wp))


def class_layer(z, w, bias, labels, count, logits, d_logits, row_loss, dz, p, seed, offset, col_partials):
    """gist_class_layer_f32: logits, CE rows, d_logits, dz = mask(d_logits . w), d_logits' 16-row chunk sums."""
    L = _lib.load()
    zp, ldz = _mat(z, 'z')
    wp, ldw = _mat(w, 'w')
    lp, ldl = _mat(logits, 'logits')
    gp, ldg = _mat(d_logits, 'd_logits')
    n, k = z.shape
    c = w.shape[0]
    dzp, lddz = (None, 0) if dz is None else _mat(dz, 'dz')
    with _Timed('gemm', ('nt', n, 2 * c if dz is not None else c, k)):
        rc = L.gist_class_layer_f32(zp, ldz, wp, ldw, _opt(bias, 'bias', torch.float32, c),
                                    _vec(labels, 'labels', torch.int32, n), int(count), lp, ldl, gp, ldg,
                                    _vec(row_loss, 'row_loss', torch.float32, n), dzp, lddz, float(p), int(seed),
                                    int(offset),
                                    None if col_partials is None else
                                    _vec(col_partials, 'col_partials', torch.float32, L.gist_row_chunks16(n) * c),
                                    n, c, k, _stream())
    _lib.check(rc, 'gist_class_layer_f32')
    return logits


def class_dw_slabs(d_logits, z, slabs):
    """gist_class_dw_slabs_f32: dW = d_logits^T . z left as 128-row slabs in `slabs`; returns their number."""
    import ctypes
    L = _lib.load()
    gp, ldg = _mat(d_logits, 'd_logits')
    zp, ldz = _mat(z, 'z')
    n, c = d_logits.shape
    k = z.shape[1]
    ns = ctypes.c_int32(0)
    with _Timed('gemm', ('tn', c, k, n)):
        rc = L.gist_class_dw_slabs_f32(gp, ldg, zp, ldz, slabs.data_ptr(), slabs.numel() * slabs.element_size(),
                                       ctypes.byref(ns), n, c, k, _stream())
    _lib.check(rc, 'gist_class_dw_slabs_f32')
    return int(ns.value)


def ln_relu_bwd_colsum_class_dw(d_out, yhat, rstd, dy, use_lynorm, relu, col_partials, d_logits, z, slabs):
    """gist_ln_relu_bwd_colsum_class_dw_f32: ln_relu_bwd_colsum and class_dw_slabs in one launch; returns the slab count."""
    import ctypes
    L = _lib.load()
    gp, ldg = _mat(d_out, 'd_out')
    yp, ldy = _mat(yhat, 'yhat')
    dp, ldd = _mat(dy, 'dy')
    n, d = yhat.shape
    lp, ldl = _mat(d_logits, 'd_logits')
    zp, ldz = _mat(z, 'z')
    nc, c = d_logits.shape
    k = z.shape[1]
    ns = ctypes.c_int32(0)
    _lib.check(L.gist_ln_relu_bwd_colsum_class_dw_f32(
        gp, ldg, yp, ldy, _opt(rstd, 'rstd', torch.float32, n), dp, ldd, n, d, int(bool(use_lynorm)), int(bool(relu)),
        _vec(col_partials, 'col_partials', torch.float32, L.gist_row_chunks16(n) * d),
        lp, ldl, zp, ldz, slabs.data_ptr(), slabs.numel() * slabs.element_size(), ctypes.byref(ns), nc, c, k, _stream()),
        'gist_ln_relu_bwd_colsum_class_dw_f32')
    return int(ns.value)


def softmax_xent_slabs(logits, slabs, n_slabs, bias, labels, mask, count, row_loss, loss, d_logits):
    L = _lib.load()
    lp, ldl = _mat(logits, 'logits')
    gp, ldg = _mat(d_logits, 'd_logits')
    n, c = logits.shape
    _lib.check(L.gist_softmax_xent_slabs_f32(lp, ldl, slabs.data_ptr() if n_slabs > 1 else None, n * c,
                                             n_slabs if n_slabs > 1 else 0, _opt(bias, 'bias', torch.float32, c),
                                             _vec(labels, 'labels', torch.int32, n),
                                             _opt(mask, 'mask', torch.uint8, n), int(count),
                                             _vec(row_loss, 'row_loss', torch.float32, n),
                                             _opt(loss, 'loss', torch.float32, 1), gp, ldg, n, c, _stream()),
               'gist_softmax_xent_slabs_f32')
    return loss


def adam_segments_(param, grad, exp_avg, exp_avg_sq, step, lr, segments, row_loss=None, n_loss_rows=0,
                   loss_count=0, loss=None, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """segments = [(begin, end, src_tensor, stride, n_src)] (gist_adam_segments_f32)."""
    L = _lib.load()
    n = param.numel()
    arr = (_lib.GradSegment * max(len(segments), 1))()
    for i, (b, e, src, stride, n_src) in enumerate(segments):
        arr[i].begin, arr[i].end, arr[i].src, arr[i].stride, arr[i].n_src = b, e, src.data_ptr(), stride, n_src
    _lib.check(L.gist_adam_segments_f32(_vec(param, 'param', torch.float32), _vec(grad, 'grad', torch.float32, n),
                                        _vec(exp_avg, 'exp_avg', torch.float32, n),
                                        _vec(exp_avg_sq, 'exp_avg_sq', torch.float32, n), n, lr, beta1, beta2,
                                        eps, weight_decay, int(step), arr, len(segments),
                                        _opt(row_loss, 'row_loss', torch.float32), int(n_loss_rows),
                                        int(loss_count), _opt(loss, 'loss', torch.float32, 1), _stream()),
               'gist_adam_segments_f32')
    return param


def extract_parts(g, ids, n_max, node_part, part_slot, batch_index, rowptr, col, t_rowptr, t_col, norm, feat, z0,
                  labels_all, labels, scratch, drop=None, feat_intra=None, ah=None):
    """gist_extract_parts_desc_batch: the one-launch extraction of a batch that is a union of parts.  z0: the [n, F]
    left half of layer 0's [h | ah] buffer; drop = (x0, p, seed, offset, mask_ld) folds layer 0's dropout into the gather;
    feat_intra + ah (the right half of the same buffer): layer 0's aggregation is formed too."""
    import ctypes
    L = _lib.load()
    n = ids.numel()
    x = _lib.ExtractPartsDesc()
    x.g_rowptr, x.g_col = _vec(g.rowptr, 'g.rowptr', torch.int32), _vec(g.col, 'g.col', torch.int32)
    x.g_t_rowptr, x.g_t_col = _vec(g.t_rowptr, 'g.t_rowptr', torch.int32), _vec(g.t_col, 'g.t_col', torch.int32)
    x.ids, x.n, x.n_max = _vec(ids, 'ids', torch.int32), n, int(n_max)
    x.node_part, x.part_slot, x.batch = node_part.data_ptr(), part_slot.data_ptr(), int(batch_index)
    x.rowptr, x.col = _vec(rowptr, 'rowptr', torch.int32, n + 1), _vec(col, 'col', torch.int32)
    x.t_rowptr, x.t_col = _vec(t_rowptr, 't_rowptr', torch.int32, n + 1), _vec(t_col, 't_col', torch.int32)
    x.col_capacity, x.norm = col.numel(), _vec(norm, 'norm', torch.float32, n)
    x.feat, x.ld_feat = _mat(feat, 'feat')
    x.n_feat = feat.shape[1]
    x.z0, x.ldz0 = _mat(z0, 'z0')
    x.labels_all, x.labels = _opt(labels_all, 'labels_all', torch.int32), _opt(labels, 'labels', torch.int32, n)
    if drop is not None:
        x0, p, seed, offset, mask_ld = drop
        x.x0, x.ldx0 = _mat(x0, 'x0')
        x.p, x.seed, x.offset, x.mask_ld = float(p), int(seed), int(offset), int(mask_ld)
    x.scratch = scratch.data_ptr()
    if ah is not None:
        if feat_intra is None or feat_intra.shape != feat.shape:
            raise ValueError('gist_amd: extract_parts: ah needs feat_intra shaped like feat')
        x.feat_intra, x.ld_intra = _mat(feat_intra, 'feat_intra')
        ap, lda = _mat(ah, 'ah')
        if lda != x.ldz0 or ah.shape[0] < n or ah.shape[1] != feat.shape[1]:
            raise ValueError('gist_amd: extract_parts: ah must be the right half of z0\'s buffer')
        x.ah = ap
    _lib.check(L.gist_extract_parts_desc_batch(ctypes.byref(x), _stream()), 'gist_extract_parts_desc_batch')


def extract_batch_drop(g, ids, remap, rowptr, col, t_rowptr, t_col, norm, feat, z0_left, labels_all, labels,
                       x0, p, seed, offset, mask_ld):
    """gist_extract_batch_drop: extraction whose feature gather writes dropout(feat) to z0 and feat to x0."""
    L = _lib.load()
    n = ids.numel()
    fp, ldf = _mat(feat, 'feat')
    zp, ldz = _mat(z0_left, 'z0')
    xp, ldx = _mat(x0, 'x0')
    _lib.check(L.gist_extract_batch_drop(
        _vec(g.rowptr, 'g.rowptr', torch.int32), _vec(g.col, 'g.col', torch.int32),
        _vec(g.t_rowptr, 'g.t_rowptr', torch.int32), _vec(g.t_col, 'g.t_col', torch.int32),
        _vec(ids, 'ids', torch.int32), n, _vec(remap, 'remap', torch.int32),
        _vec(rowptr, 'rowptr', torch.int32, n + 1), _vec(col, 'col', torch.int32),
        _vec(t_rowptr, 't_rowptr', torch.int32, n + 1), _vec(t_col, 't_col', torch.int32),
        col.numel(), _vec(norm, 'norm', torch.float32, n), fp, ldf, feat.shape[1], zp, ldz,
        _opt(labels_all, 'labels_all', torch.int32), _opt(labels, 'labels', torch.int32, n),
        xp, ldx, float(p), int(seed), int(offset), int(mask_ld), _stream()), 'gist_extract_batch_drop')
