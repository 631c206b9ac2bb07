"""torch.ops.gist.* (gist_amd/ops.py): the HIP kernels as dispatcher-registered operators
(BASELINE.json north star: "called from Python through PyTorch-ROCm custom ops"; SURVEY.md
section 8b(ii)).  CPU: the operators exist with the documented schemas and have NO CPU kernel.
GPU: torch.library.opcheck on every operator (schema, fake/meta kernel, autograd registration)
and numeric checks of the autograd formulas against the oracle."""
import numpy as np
import pytest
import torch


def test_ops_are_registered_with_schemas_and_have_no_cpu_kernel():
    from gist_amd import ops
    for name in ops.OPS:
        assert hasattr(torch.ops.gist, name), name
    s = str(torch.ops.gist.spmm_sum.default._schema)
    assert s.startswith('gist::spmm_sum(Tensor rowptr, Tensor col, Tensor t_rowptr, Tensor t_col, Tensor x')
    assert 'Tensor(a0!) param' in str(torch.ops.gist.adam_step_.default._schema)       # in place
    assert 'Tensor(a0!) dst' in str(torch.ops.gist.block_scatter_.default._schema)
    with pytest.raises(NotImplementedError):         # the product has no CPU fallback
        torch.ops.gist.matmul(torch.ones(2, 2), torch.ones(2, 2))
    rp = torch.tensor([0, 1, 2], dtype=torch.int32)
    cl = torch.tensor([1, 0], dtype=torch.int32)
    with pytest.raises(NotImplementedError):
        torch.ops.gist.spmm_sum(rp, cl, rp, cl, torch.ones(2, 4))


def _toy_graph(n=300, seed=0):
    from gist_amd.graph import Graph
    rs = np.random.RandomState(seed)
    m = n * 6
    src = rs.randint(0, n, m)
    dst = rs.randint(0, n - 1, m)                     # node n-1 has no in-edge
    return Graph.from_edges(src, dst, n).to(torch.device('cuda', 0))


@pytest.mark.gpu
def test_opcheck_all_ops():
    from torch.library import opcheck
    from gist_amd import ops  # noqa: F401
    dev = torch.device('cuda', 0)
    g = _toy_graph()
    n = g.number_of_nodes()
    utils = ('test_schema', 'test_autograd_registration', 'test_faketensor')
    gen = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(n, 24, device=dev, generator=gen, requires_grad=True)
    sc = torch.rand(n, device=dev, generator=gen)
    opcheck(torch.ops.gist.spmm_sum, (g.rowptr, g.col, g.t_rowptr, g.t_col, x, sc, None), test_utils=utils)
    opcheck(torch.ops.gist.spmm_sum, (g.rowptr, g.col, g.t_rowptr, g.t_col, x.detach()), test_utils=utils)
    W = torch.randn(16, 48, device=dev, generator=gen, requires_grad=True)
    b = torch.randn(16, device=dev, generator=gen, requires_grad=True)
    for ln, relu in ((True, True), (False, False)):
        opcheck(torch.ops.gist.sage_layer,
                (g.rowptr, g.col, g.t_rowptr, g.t_col, g.norm(), x, W, b, ln, relu, 0.0, 0, 0),
                test_utils=utils)
    out, z, yhat, rstd = torch.ops.gist.sage_layer_fwd(g.rowptr, g.col, g.norm(), x.detach(),
                                                       W.detach(), b.detach(), True, True, 0.25, 3, 0)
    opcheck(torch.ops.gist.sage_layer_fwd,
            (g.rowptr, g.col, g.norm(), x.detach(), W.detach(), b.detach(), True, True, 0.25, 3, 0),
            test_utils=utils)
    opcheck(torch.ops.gist.sage_layer_bwd,
            (g.t_rowptr, g.t_col, g.norm(), torch.randn_like(out), z, W.detach(), yhat, rstd, True,
             True, 0.25, 3, 0, True), test_utils=utils)
    ids = torch.arange(0, n, 3, dtype=torch.int32, device=dev)
    remap = torch.full((n,), -1, dtype=torch.int32, device=dev)
    opcheck(torch.ops.gist.induced_subgraph, (g.rowptr, g.col, ids, remap), test_utils=('test_schema',))
    assert bool((remap == -1).all())                                   # scratch restored
    src = torch.randn(40, 50, device=dev, generator=gen)
    ri = torch.tensor([3, 1, 7, 39], dtype=torch.int32, device=dev)
    ci = torch.tensor([0, 49, 5], dtype=torch.int32, device=dev)
    opcheck(torch.ops.gist.block_gather, (src, ri, ci), test_utils=utils)
    opcheck(torch.ops.gist.block_gather, (src, None, ci), test_utils=utils)
    opcheck(torch.ops.gist.block_scatter_, (src.clone(), torch.randn(4, 3, device=dev), ri, ci),
            test_utils=utils)
    p = torch.randn(1000, device=dev, generator=gen)
    opcheck(torch.ops.gist.adam_step_,
            (p, torch.randn_like(p), torch.zeros_like(p), torch.zeros_like(p), 1, 0.01, 0.9, 0.999,
             1e-8, 5e-4), test_utils=utils)
    a = torch.randn(33, 20, device=dev, generator=gen, requires_grad=True)
    w = torch.randn(20, 12, device=dev, generator=gen, requires_grad=True)
    opcheck(torch.ops.gist.matmul, (a, w), test_utils=utils)
    opcheck(torch.ops.gist.layer_norm_rows_fwd, (a, True), test_utils=utils)


@pytest.mark.gpu
def test_ops_match_oracle_and_autograd_formulas():
    """Values and gradients of the registered ops against the oracle (fp32, 1e-4) and against
    torch's own autograd on a dense restatement."""
    from gist_amd import ops  # noqa: F401
    from oracle import gist_oracle as O
    dev = torch.device('cuda', 0)
    g = _toy_graph(seed=1)
    n = g.number_of_nodes()
    rp, cl = g.rowptr.cpu().numpy().astype(np.int64), g.col.cpu().numpy().astype(np.int64)
    trp, tcl = O.transpose_csr(rp, cl)
    rs = np.random.RandomState(2)
    h = rs.randn(n, 20).astype(np.float32)
    W = (rs.randn(12, 40) * 0.2).astype(np.float32)
    b = (rs.randn(12) * 0.1).astype(np.float32)
    d_out = rs.randn(n, 12).astype(np.float32)
    ht = torch.from_numpy(h).to(dev).requires_grad_()
    Wt = torch.from_numpy(W).to(dev).requires_grad_()
    bt = torch.from_numpy(b).to(dev).requires_grad_()
    out = torch.ops.gist.sage_layer(g.rowptr, g.col, g.t_rowptr, g.t_col, g.norm(), ht, Wt, bt,
                                    True, True, 0.0, 0, 0)[0]
    (out * torch.from_numpy(d_out).to(dev)).sum().backward()
    ref, cache = O.sage_layer_forward(rp, cl, h, W, b, True, True)
    dh, dW, db = O.sage_layer_backward(cache, d_out, trp, tcl)
    for got, want in ((out, ref), (ht.grad, dh), (Wt.grad, dW), (bt.grad, db)):
        assert np.abs(got.detach().cpu().numpy() - want).max() < 1e-4
    # spmm_sum: value + gradient with both scales
    xs = torch.from_numpy(h).to(dev).requires_grad_()
    so = torch.rand(n, device=dev)
    ss = torch.rand(n, device=dev)
    y = torch.ops.gist.spmm_sum(g.rowptr, g.col, g.t_rowptr, g.t_col, xs, so, ss)
    y.backward(torch.from_numpy(d_out[:, :1]).to(dev).expand(n, 20).contiguous())
    want = O.spmm_sum(rp, cl, h, out_scale=so.cpu().numpy(), src_scale=ss.cpu().numpy())
    assert np.abs(y.detach().cpu().numpy() - want).max() < 1e-4
    gy = np.repeat(d_out[:, :1], 20, axis=1)
    want_g = O.spmm_sum(trp, tcl, gy, out_scale=ss.cpu().numpy(), src_scale=so.cpu().numpy())
    assert np.abs(xs.grad.cpu().numpy() - want_g).max() < 1e-4
    # induced subgraph == oracle, scratch restored
    ids = np.sort(rs.choice(n, 90, replace=False)).astype(np.int64)
    remap = torch.full((n,), -1, dtype=torch.int32, device=dev)
    srp, scl = torch.ops.gist.induced_subgraph(g.rowptr, g.col, torch.from_numpy(ids.astype(np.int32)).to(dev), remap)
    orp, ocl = O.induced_subgraph(rp, cl, ids)[:2]
    assert np.array_equal(srp.cpu().numpy(), orp) and np.array_equal(scl.cpu().numpy(), ocl)
    assert bool((remap == -1).all())
    # block movers + Adam against numpy
    src = torch.randn(30, 40, device=dev)
    ri = torch.tensor([5, 0, 29], dtype=torch.int32, device=dev)
    ci = torch.tensor([39, 2], dtype=torch.int32, device=dev)
    blk = torch.ops.gist.block_gather(src, ri, ci)
    assert torch.equal(blk, src[ri.long()][:, ci.long()])
    dst = torch.zeros(30, 40, device=dev)
    torch.ops.gist.block_scatter_(dst, blk, ri, ci)
    assert torch.equal(dst[ri.long()][:, ci.long()], blk) and int((dst != 0).sum()) <= 6
    p0 = rs.randn(500).astype(np.float32)
    gr = rs.randn(500).astype(np.float32)
    p, m, v = torch.from_numpy(p0.copy()).to(dev), torch.zeros(500, device=dev), torch.zeros(500, device=dev)
    torch.ops.gist.adam_step_(p, torch.from_numpy(gr).to(dev), m, v, 1, 0.01, 0.9, 0.999, 1e-8, 5e-4)
    pn, mn, vn = p0.copy(), np.zeros(500, np.float32), np.zeros(500, np.float32)
    O.adam_step(pn, gr, mn, vn, 1, 0.01, weight_decay=5e-4)
    assert np.abs(p.cpu().numpy() - pn).max() < 1e-6
