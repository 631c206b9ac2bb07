"""torch.autograd glue: lets reference-style code (`loss.backward()`) drive the HIP
kernels.  torch supplies the tape and the memory; all arithmetic is in libgist_hip.so.

  spmm_sum(g, x)        g.update_all(copy_src, sum)                (modules.py:224-225)
  sage_layer(...)       one whole ISTSAGELayer.forward, fused      (modules.py:218-237)
"""
import torch

from . import hip


class _SpmmSum(torch.autograd.Function):
    """y[v] = out_scale[v] * sum_{u->v} src_scale[u] * x[u]; gradient walks the reversed CSR."""

    @staticmethod
    def forward(ctx, x, g, out_scale, src_scale):
        x = x.contiguous() if x.stride(-1) != 1 else x
        y = torch.empty(g.number_of_nodes(), x.shape[1], dtype=torch.float32, device=x.device)
        hip.spmm(g.rowptr, g.col, x, y, out_scale=out_scale, src_scale=src_scale)
        ctx.g, ctx.out_scale, ctx.src_scale = g, out_scale, src_scale
        return y

    @staticmethod
    def backward(ctx, gy):
        g = ctx.g
        gy = gy.contiguous() if gy.stride(-1) != 1 else gy
        gx = torch.empty_like(gy)
        # d/dx of the forward: roles of the two scales swap on the reversed graph
        hip.spmm(g.t_rowptr, g.t_col, gy, gx, out_scale=ctx.src_scale, src_scale=ctx.out_scale)
        return gx, None, None, None


def spmm_sum(g, x, out_scale=None, src_scale=None):
    return _SpmmSum.apply(x, g, out_scale, src_scale)


_drop_counter = [0]


def next_dropout_offset(n_elems):
    """Monotone counter so every dropout call of a process draws a fresh mask."""
    off = _drop_counter[0]
    _drop_counter[0] += int(n_elems) + (int(n_elems) & 1)
    return off


class _SageLayer(torch.autograd.Function):
    """Fused ISTSAGELayer: aggregate -> [h | ah] -> dropout -> linear -> LN -> relu."""

    @staticmethod
    def forward(ctx, h, weight, bias, g, use_lynorm, relu, p_drop, seed):
        n, n_in = h.shape
        n_out = weight.shape[0]
        dev = h.device
        norm = g.norm()
        z = torch.empty(n, 2 * n_in, dtype=torch.float32, device=dev)
        hip.block_gather(h if h.stride(-1) == 1 else h.contiguous(), None, None, z[:, :n_in])
        hip.spmm(g.rowptr, g.col, z[:, :n_in], z[:, n_in:], out_scale=norm)
        drop_off = None
        if p_drop > 0.0:
            drop_off = next_dropout_offset(z.numel())
            hip.dropout_(z, p_drop, seed, drop_off)
        y = torch.empty(n, n_out, dtype=torch.float32, device=dev)
        hip.gemm_nt(z, weight, bias, y)
        rstd = None
        if use_lynorm or relu:
            out = torch.empty(n, n_out, dtype=torch.float32, device=dev)
            rstd = torch.empty(n, dtype=torch.float32, device=dev) if use_lynorm else None
            hip.ln_relu_fwd(y, out, rstd, use_lynorm, relu)
        else:
            out = y
        ctx.save_for_backward(z, weight, y, rstd if rstd is not None else torch.empty(0, device=dev))
        ctx.g, ctx.norm = g, norm
        ctx.cfg = (use_lynorm, relu, p_drop, seed, drop_off, n_in)
        ctx.need_dh = ctx.needs_input_grad[0]
        return out

    @staticmethod
    def backward(ctx, d_out):
        z, weight, yhat, rstd = ctx.saved_tensors
        use_lynorm, relu, p_drop, seed, drop_off, n_in = ctx.cfg
        g = ctx.g
        n, n_out = yhat.shape
        dev = z.device
        d_out = d_out if d_out.stride(-1) == 1 else d_out.contiguous()
        if use_lynorm or relu:
            dy = torch.empty(n, n_out, dtype=torch.float32, device=dev)
            hip.ln_relu_bwd(d_out, yhat, rstd if use_lynorm else None, dy, use_lynorm, relu)
        else:
            dy = d_out
        dW = torch.empty_like(weight)
        hip.gemm_tn(dy, z, dW)
        db = torch.empty(n_out, dtype=torch.float32, device=dev)
        hip.colsum(dy, db)
        dh = None
        if ctx.need_dh:
            dz = torch.empty_like(z)
            hip.gemm_nn(dy, weight, dz)
            if p_drop > 0.0:
                hip.dropout_(dz, p_drop, seed, drop_off)
            hip.spmm(g.t_rowptr, g.t_col, dz[:, n_in:], dz[:, :n_in], src_scale=ctx.norm,
                     accumulate=True)
            dh = dz[:, :n_in]
        return dh, dW, db, None, None, None, None, None


def sage_layer(g, h, weight, bias, use_lynorm, relu, p_drop=0.0, seed=0):
    return _SageLayer.apply(h, weight, bias, g, bool(use_lynorm), bool(relu), float(p_drop),
                            int(seed))


class _MatMul(torch.autograd.Function):
    """y = x @ w on the fp32-MFMA GEMM (GraphConv's weight is [in, out], gcn/gcn.py:30-56)."""

    @staticmethod
    def forward(ctx, x, w):
        x = x if x.stride(-1) == 1 else x.contiguous()
        w = w if w.stride(-1) == 1 else w.contiguous()
        y = torch.empty(x.shape[0], w.shape[1], dtype=torch.float32, device=x.device)
        hip.gemm_nn(x, w, y)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy if gy.stride(-1) == 1 else gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            hip.gemm_nt(gy, w, None, gx)            # gy @ w.T
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(w)
            hip.gemm_tn(x, gy, gw)                  # x.T @ gy
        return gx, gw


def matmul(x, w):
    return _MatMul.apply(x, w)


class _LayerNormRows(torch.autograd.Function):
    """LayerNorm without affine over the last dim of a 2-D view (eps 1e-5), optional relu."""

    @staticmethod
    def forward(ctx, x, relu):
        y = x.contiguous().clone()
        out = torch.empty_like(y)
        rstd = torch.empty(y.shape[0], dtype=torch.float32, device=y.device)
        hip.ln_relu_fwd(y, out, rstd, True, relu)
        ctx.save_for_backward(y, rstd)
        ctx.relu = relu
        return out

    @staticmethod
    def backward(ctx, g):
        yhat, rstd = ctx.saved_tensors
        g = g if g.stride(-1) == 1 else g.contiguous()
        dy = torch.empty_like(yhat)
        hip.ln_relu_bwd(g, yhat, rstd, dy, True, ctx.relu)
        return dy, None


def layer_norm_rows(x, relu=False):
    return _LayerNormRows.apply(x, bool(relu))


def whole_tensor_layer_norm(h):
    """F.layer_norm(h, h.shape) as gcn/gcn.py:65-66 uses it: ONE mean/variance over the whole
    [N, hidden] tensor."""
    return layer_norm_rows(h.reshape(1, -1)).reshape(h.shape)
