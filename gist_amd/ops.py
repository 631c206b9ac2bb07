"""torch.ops.gist.* -- the HIP kernels registered with the PyTorch dispatcher.

BASELINE.json's north star asks for the hot path to be "called from Python through
PyTorch-ROCm custom ops"; SURVEY.md section 8b(ii) lists the operators.  Each op below is a
`torch.library.custom_op` whose CUDA (= HIP on ROCm) implementation is a call through the C
ABI of libgist_hip.so (gist_amd/hip.py -> include/gist_hip.h); shapes are described to the
dispatcher by a fake (meta) implementation, gradients by `register_autograd` formulas that
are themselves gist ops.  There is NO CPU implementation: a CPU tensor fails in the dispatcher
("no kernel for CPU"), loudly -- the CPU restatement lives in oracle/ and is test-only.

  gist::spmm_sum           g.update_all(fn.copy_src, fn.sum) [* 1/deg]   modules.py:223-226
  gist::sage_layer         one whole ISTSAGELayer.forward                 modules.py:218-237
  gist::sage_layer_fwd/bwd its two halves (what the autograd formula calls)
  gist::induced_subgraph   g.subgraph(ids) structure                      partition_utils.py:23
  gist::block_gather       W[:, cols][rows, :]  (IST dispatch)            cluster_gcn_ist_distrib.py:221-226
  gist::block_scatter_     base[rows, cols] = block  (IST sync)           :126-133
  gist::adam_step_         torch.optim.Adam.step on one tensor            :405-407,417
  gist::matmul, gist::layer_norm_rows   pieces of the GraphConv path      gcn/gcn.py:30-67

The reversed CSR travels with every aggregation op (t_rowptr, t_col): the backward of an
aggregation is the same op on the reversed graph with the two scales swapped.
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import hip


def _c(t):
    return t if t.stride(-1) == 1 else t.contiguous()


# -- aggregation ----------------------------------------------------------------------------
@torch.library.custom_op('gist::spmm_sum', mutates_args=(), device_types='cuda')
def spmm_sum(rowptr: Tensor, col: Tensor, t_rowptr: Tensor, t_col: Tensor, x: Tensor,
             out_scale: Optional[Tensor] = None, src_scale: Optional[Tensor] = None) -> Tensor:
    """y[v] = out_scale[v] * sum_{u->v} src_scale[u] * x[u] over the in-edge CSR (rowptr, col)."""
    x = _c(x)
    y = torch.empty(rowptr.numel() - 1, x.shape[1], dtype=torch.float32, device=x.device)
    hip.spmm(rowptr, col, x, y, out_scale=out_scale, src_scale=src_scale)
    return y


@spmm_sum.register_fake
def _(rowptr, col, t_rowptr, t_col, x, out_scale=None, src_scale=None):
    return x.new_empty(rowptr.shape[0] - 1, x.shape[1])


def _spmm_setup(ctx, inputs, output):
    rowptr, col, t_rowptr, t_col, x, out_scale, src_scale = inputs
    ctx.save_for_backward(rowptr, col, t_rowptr, t_col, out_scale, src_scale)


def _spmm_backward(ctx, gy):
    rowptr, col, t_rowptr, t_col, out_scale, src_scale = ctx.saved_tensors
    # d/dx walks the reversed graph; the roles of the two scales swap
    gx = torch.ops.gist.spmm_sum(t_rowptr, t_col, rowptr, col, gy, src_scale, out_scale)
    return None, None, None, None, gx, None, None


spmm_sum.register_autograd(_spmm_backward, setup_context=_spmm_setup)


# -- one ISTSAGELayer -----------------------------------------------------------------------
@torch.library.custom_op('gist::sage_layer_fwd', mutates_args=(), device_types='cuda')
def sage_layer_fwd(rowptr: Tensor, col: Tensor, norm: Tensor, h: Tensor, weight: Tensor,
                   bias: Tensor, use_lynorm: bool, relu: bool, p_drop: float, seed: int,
                   drop_offset: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """aggregate -> [h | ah] -> dropout -> linear -> LayerNorm -> relu; returns
    (out, z = the dropped [h | ah], yhat = normalised pre-activation, rstd)."""
    n, n_in = h.shape
    n_out = weight.shape[0]
    dev = h.device
    z = torch.empty(n, 2 * n_in, dtype=torch.float32, device=dev)
    hip.block_gather(_c(h), None, None, z[:, :n_in])
    hip.spmm(rowptr, col, z[:, :n_in], z[:, n_in:], out_scale=norm)
    if p_drop > 0.0:
        hip.dropout_(z, p_drop, seed, drop_offset)
    y = torch.empty(n, n_out, dtype=torch.float32, device=dev)
    hip.gemm_nt(z, _c(weight), bias, y)
    rstd = torch.empty(n if use_lynorm else 0, dtype=torch.float32, device=dev)
    if use_lynorm or relu:
        out = torch.empty(n, n_out, dtype=torch.float32, device=dev)
        hip.ln_relu_fwd(y, out, rstd if use_lynorm else None, use_lynorm, relu)
    else:
        out = y.clone()
    return out, z, y, rstd


@sage_layer_fwd.register_fake
def _(rowptr, col, norm, h, weight, bias, use_lynorm, relu, p_drop, seed, drop_offset):
    n, n_in = h.shape
    n_out = weight.shape[0]
    return (h.new_empty(n, n_out), h.new_empty(n, 2 * n_in), h.new_empty(n, n_out),
            h.new_empty(n if use_lynorm else 0))


@torch.library.custom_op('gist::sage_layer_bwd', mutates_args=(), device_types='cuda')
def sage_layer_bwd(t_rowptr: Tensor, t_col: Tensor, norm: Tensor, d_out: Tensor, z: Tensor,
                   weight: Tensor, yhat: Tensor, rstd: Tensor, use_lynorm: bool, relu: bool,
                   p_drop: float, seed: int, drop_offset: int,
                   need_dh: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """SURVEY.md appendix A, backward block: (dh [n, in] or empty, dW, db)."""
    n, n_out = yhat.shape
    n_in = z.shape[1] // 2
    dev = z.device
    d_out = _c(d_out)
    if use_lynorm or relu:
        dy = torch.empty(n, n_out, dtype=torch.float32, device=dev)
        hip.ln_relu_bwd(d_out, yhat, rstd if use_lynorm else None, dy, use_lynorm, relu)
    else:
        dy = d_out
    dW = torch.empty(n_out, 2 * n_in, dtype=torch.float32, device=dev)
    hip.gemm_tn(dy, z, dW)
    db = torch.empty(n_out, dtype=torch.float32, device=dev)
    hip.colsum(dy, db)
    if not need_dh:
        return torch.empty(0, n_in, dtype=torch.float32, device=dev), dW, db
    dz = torch.empty(n, 2 * n_in, dtype=torch.float32, device=dev)
    hip.gemm_nn(dy, _c(weight), dz)
    if p_drop > 0.0:
        hip.dropout_(dz, p_drop, seed, drop_offset)
    hip.spmm(t_rowptr, t_col, dz[:, n_in:], dz[:, :n_in], src_scale=norm, accumulate=True)
    return dz[:, :n_in].contiguous(), dW, db


@sage_layer_bwd.register_fake
def _(t_rowptr, t_col, norm, d_out, z, weight, yhat, rstd, use_lynorm, relu, p_drop, seed,
      drop_offset, need_dh):
    n, n_out = yhat.shape
    n_in = z.shape[1] // 2
    return (z.new_empty(n if need_dh else 0, n_in), z.new_empty(n_out, 2 * n_in),
            z.new_empty(n_out))


@torch.library.custom_op('gist::sage_layer', mutates_args=(), device_types='cuda')
def sage_layer(rowptr: Tensor, col: Tensor, t_rowptr: Tensor, t_col: Tensor, norm: Tensor,
               h: Tensor, weight: Tensor, bias: Tensor, use_lynorm: bool, relu: bool,
               p_drop: float, seed: int, drop_offset: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """ISTSAGELayer.forward as ONE differentiable op: (out, z, yhat, rstd); only `out` carries
    a gradient, the other three are what the backward formula needs."""
    return torch.ops.gist.sage_layer_fwd(rowptr, col, norm, h, weight, bias, use_lynorm, relu,
                                         p_drop, seed, drop_offset)


@sage_layer.register_fake
def _(rowptr, col, t_rowptr, t_col, norm, h, weight, bias, use_lynorm, relu, p_drop, seed,
      drop_offset):
    n, n_in = h.shape
    n_out = weight.shape[0]
    return (h.new_empty(n, n_out), h.new_empty(n, 2 * n_in), h.new_empty(n, n_out),
            h.new_empty(n if use_lynorm else 0))


def _sage_setup(ctx, inputs, output):
    (rowptr, col, t_rowptr, t_col, norm, h, weight, bias, use_lynorm, relu, p_drop, seed,
     drop_offset) = inputs
    out, z, yhat, rstd = output
    ctx.save_for_backward(t_rowptr, t_col, norm, z, weight, yhat, rstd)
    ctx.cfg = (use_lynorm, relu, p_drop, seed, drop_offset)
    ctx.need_dh = ctx.needs_input_grad[5]


def _sage_backward(ctx, d_out, d_z, d_yhat, d_rstd):
    t_rowptr, t_col, norm, z, weight, yhat, rstd = ctx.saved_tensors
    use_lynorm, relu, p_drop, seed, drop_offset = ctx.cfg
    dh, dW, db = torch.ops.gist.sage_layer_bwd(t_rowptr, t_col, norm, d_out, z, weight, yhat, rstd,
                                               use_lynorm, relu, p_drop, seed, drop_offset,
                                               ctx.need_dh)
    return (None, None, None, None, None, dh if ctx.need_dh else None, dW, db,
            None, None, None, None, None)


sage_layer.register_autograd(_sage_backward, setup_context=_sage_setup)


# -- cluster batch structure ----------------------------------------------------------------
@torch.library.custom_op('gist::induced_subgraph', mutates_args=('remap',), device_types='cuda')
def induced_subgraph(rowptr: Tensor, col: Tensor, node_ids: Tensor,
                     remap: Tensor) -> Tuple[Tensor, Tensor]:
    """CSR of the subgraph induced by node_ids (node i of the result = node_ids[i], edge order
    kept).  `remap` is an int32 scratch of one entry per graph node holding -1 everywhere; it is
    used and restored.  One host sync sizes the column array exactly."""
    nb = node_ids.numel()
    dev = rowptr.device
    hip.induced_mark(node_ids, remap)
    srp = torch.empty(nb + 1, dtype=torch.int32, device=dev)
    hip.induced_rowptr(rowptr, col, node_ids, remap, srp)
    nnz = int(srp[-1].item())
    scl = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    hip.induced_fill(rowptr, col, node_ids, remap, srp, scl)
    hip.induced_mark(node_ids, remap, unmark=True)
    return srp, scl[:nnz].clone() if nnz < scl.numel() else scl


@induced_subgraph.register_fake
def _(rowptr, col, node_ids, remap):
    ctx = torch.library.get_ctx()
    nnz = ctx.new_dynamic_size()
    return rowptr.new_empty(node_ids.shape[0] + 1), col.new_empty(nnz)


# -- IST block movers -------------------------------------------------------------------------
@torch.library.custom_op('gist::block_gather', mutates_args=(), device_types='cuda')
def block_gather(src: Tensor, row_idx: Optional[Tensor] = None,
                 col_idx: Optional[Tensor] = None) -> Tensor:
    """out[i, j] = src[row_idx[i], col_idx[j]] (None = identity); int32 index lists."""
    nr = src.shape[0] if row_idx is None else row_idx.numel()
    nc = src.shape[1] if col_idx is None else col_idx.numel()
    out = torch.empty(nr, nc, dtype=torch.float32, device=src.device)
    hip.block_gather(_c(src), row_idx, col_idx, out)
    return out


@block_gather.register_fake
def _(src, row_idx=None, col_idx=None):
    nr = src.shape[0] if row_idx is None else row_idx.shape[0]
    nc = src.shape[1] if col_idx is None else col_idx.shape[0]
    return src.new_empty(nr, nc)


@torch.library.custom_op('gist::block_scatter_', mutates_args=('dst',), device_types='cuda')
def block_scatter_(dst: Tensor, src: Tensor, row_idx: Optional[Tensor] = None,
                   col_idx: Optional[Tensor] = None) -> None:
    """dst[row_idx[i], col_idx[j]] = src[i, j] (None = identity), in place."""
    hip.block_scatter(_c(src), row_idx, col_idx, dst)


# -- optimiser --------------------------------------------------------------------------------
@torch.library.custom_op('gist::adam_step_', mutates_args=('param', 'exp_avg', 'exp_avg_sq'),
                         device_types='cuda')
def adam_step_(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, step: int,
               lr: float, beta1: float, beta2: float, eps: float, weight_decay: float) -> None:
    """torch.optim.Adam (coupled L2), one flat tensor, in place; `step` is 1-based."""
    hip.adam_(param, grad if grad.is_contiguous() else grad.contiguous(), exp_avg, exp_avg_sq, step,
              lr, beta1, beta2, eps, weight_decay)


# -- GraphConv pieces (BASELINE config 1) --------------------------------------------------------
@torch.library.custom_op('gist::matmul', mutates_args=(), device_types='cuda')
def matmul(x: Tensor, w: Tensor) -> Tensor:
    """x [m, k] @ w [k, n] on the fp32-MFMA GEMM (GraphConv's weight is [in, out])."""
    y = torch.empty(x.shape[0], w.shape[1], dtype=torch.float32, device=x.device)
    hip.gemm_nn(_c(x), _c(w), y)
    return y


@matmul.register_fake
def _(x, w):
    return x.new_empty(x.shape[0], w.shape[1])


@torch.library.custom_op('gist::matmul_nt', mutates_args=(), device_types='cuda')
def matmul_nt(a: Tensor, b: Tensor) -> Tensor:
    """a [m, k] @ b[n, k]^T"""
    y = torch.empty(a.shape[0], b.shape[0], dtype=torch.float32, device=a.device)
    hip.gemm_nt(_c(a), _c(b), None, y)
    return y


@matmul_nt.register_fake
def _(a, b):
    return a.new_empty(a.shape[0], b.shape[0])


@torch.library.custom_op('gist::matmul_tn', mutates_args=(), device_types='cuda')
def matmul_tn(a: Tensor, b: Tensor) -> Tensor:
    """a [k, m]^T @ b [k, n]"""
    y = torch.empty(a.shape[1], b.shape[1], dtype=torch.float32, device=a.device)
    hip.gemm_tn(_c(a), _c(b), y)
    return y


@matmul_tn.register_fake
def _(a, b):
    return a.new_empty(a.shape[1], b.shape[1])


def _mm_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _mm_backward(ctx, gy):
    x, w = ctx.saved_tensors
    gx = torch.ops.gist.matmul_nt(gy, w) if ctx.needs_input_grad[0] else None     # gy @ w.T
    gw = torch.ops.gist.matmul_tn(x, gy) if ctx.needs_input_grad[1] else None     # x.T @ gy
    return gx, gw


matmul.register_autograd(_mm_backward, setup_context=_mm_setup)


@torch.library.custom_op('gist::layer_norm_rows_fwd', mutates_args=(), device_types='cuda')
def layer_norm_rows_fwd(x: Tensor, relu: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """LayerNorm without affine over the last dim (eps 1e-5), optional relu: (out, yhat, rstd)."""
    y = x.contiguous().clone()
    out = torch.empty_like(y)
    rstd = torch.empty(y.shape[0], dtype=torch.float32, device=y.device)
    hip.ln_relu_fwd(y, out, rstd, True, relu)
    return out, y, rstd


@layer_norm_rows_fwd.register_fake
def _(x, relu):
    return x.new_empty(x.shape), x.new_empty(x.shape), x.new_empty(x.shape[0])


@torch.library.custom_op('gist::layer_norm_rows_bwd', mutates_args=(), device_types='cuda')
def layer_norm_rows_bwd(g: Tensor, yhat: Tensor, rstd: Tensor, relu: bool) -> Tensor:
    dy = torch.empty_like(yhat)
    hip.ln_relu_bwd(_c(g), yhat, rstd, dy, True, relu)
    return dy


@layer_norm_rows_bwd.register_fake
def _(g, yhat, rstd, relu):
    return yhat.new_empty(yhat.shape)


def _ln_setup(ctx, inputs, output):
    out, yhat, rstd = output
    ctx.save_for_backward(yhat, rstd)
    ctx.relu = inputs[1]


def _ln_backward(ctx, g, g_yhat, g_rstd):
    yhat, rstd = ctx.saved_tensors
    return torch.ops.gist.layer_norm_rows_bwd(g, yhat, rstd, ctx.relu), None


layer_norm_rows_fwd.register_autograd(_ln_backward, setup_context=_ln_setup)

OPS = ('spmm_sum', 'sage_layer', 'sage_layer_fwd', 'sage_layer_bwd', 'induced_subgraph',
       'block_gather', 'block_scatter_', 'adam_step_', 'matmul', 'matmul_nt', 'matmul_tn',
       'layer_norm_rows_fwd', 'layer_norm_rows_bwd')
