"""Differentiable entry points of the module path: reference-style code (`pred = model(g)`,
`loss.backward()`) drives the HIP kernels through the dispatcher-registered operators of
gist_amd/ops.py (torch.ops.gist.*).  torch supplies the tape and the memory; all arithmetic is
in libgist_hip.so.

  spmm_sum(g, x)        g.update_all(copy_src, sum)                (modules.py:224-225)
  sage_layer(...)       one whole ISTSAGELayer.forward, fused      (modules.py:218-237)
  matmul, layer_norm_rows, whole_tensor_layer_norm                 (gcn/gcn.py:30-67)
"""
import torch

from . import ops  # noqa: F401  (registers torch.ops.gist.*)


def spmm_sum(g, x, out_scale=None, src_scale=None):
    """y[v] = out_scale[v] * sum_{u->v} src_scale[u] * x[u]; gradient walks the reversed CSR."""
    return torch.ops.gist.spmm_sum(g.rowptr, g.col, g.t_rowptr, g.t_col, x, out_scale, src_scale)


_drop_counter = [0]


def next_dropout_offset(n_elems):
    """Monotone counter so every dropout call of a process draws a fresh mask."""
    off = _drop_counter[0]
    _drop_counter[0] += int(n_elems) + (int(n_elems) & 1)
    return off


def sage_layer(g, h, weight, bias, use_lynorm, relu, p_drop=0.0, seed=0):
    """Fused ISTSAGELayer: aggregate -> [h | ah] -> dropout -> linear -> LN -> relu."""
    p_drop = float(p_drop)
    off = next_dropout_offset(h.shape[0] * 2 * h.shape[1]) if p_drop > 0.0 else 0
    out, _z, _yhat, _rstd = torch.ops.gist.sage_layer(
        g.rowptr, g.col, g.t_rowptr, g.t_col, g.norm(), h, weight, bias, bool(use_lynorm),
        bool(relu), p_drop, int(seed), off)
    return out


def matmul(x, w):
    """y = x @ w on the fp32-MFMA GEMM (GraphConv's weight is [in, out], gcn/gcn.py:30-56)."""
    return torch.ops.gist.matmul(x, w)


def layer_norm_rows(x, relu=False):
    """LayerNorm without affine over the last dim of a 2-D tensor (eps 1e-5), optional relu."""
    return torch.ops.gist.layer_norm_rows_fwd(x, bool(relu))[0]


def whole_tensor_layer_norm(h):
    """F.layer_norm(h, h.shape) as gcn/gcn.py:65-66 uses it: ONE mean/variance over the whole
    [N, hidden] tensor."""
    return layer_norm_rows(h.reshape(1, -1)).reshape(h.shape)
