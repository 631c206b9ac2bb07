"""GPU parity of the MODEL level against golden vectors recorded from the reference
(tests/golden, made by oracle/gen_golden.py) -- both host paths:
  * the drop-in nn.Module path (gist_amd.modules + autograd), used like the reference
  * the SageEngine fast path (preallocated, concat-free)
Tolerance 1e-4 fp32 (north_star); index work bit exact."""
import glob
import os
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import gist_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = 'cuda:0'
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).to(DEV)


def err(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    return float(np.abs(a - b).max())


def graph_of(d):
    from gist_amd.graph import Graph
    return Graph.from_edges(d['src'], d['dst'], int(d['n'])).to(DEV)


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLD, 'G1_layer_*.npz'))),
                         ids=os.path.basename)
def test_G1_layer_module(path):
    from gist_amd.modules import ISTSAGELayer
    d = np.load(path)
    g = graph_of(d)
    # structure: our CSR builder == the reference stub's
    assert np.array_equal(g.rowptr.cpu().numpy(), d['rowptr'])
    assert np.array_equal(g.col.cpu().numpy(), d['col'])
    fin, fout = d['h'].shape[1], d['W'].shape[0]
    layer = ISTSAGELayer(fin, fout, 0.0, bool(d['use_lynorm']),
                         activation=F.relu if bool(d['relu']) else None).to(DEV)
    with torch.no_grad():
        layer.linear.weight.copy_(T(d['W']))
        layer.linear.bias.copy_(T(d['b']))
    h = T(d['h']).requires_grad_(True)
    out = layer(g, h)
    assert err(out, d['out']) < TOL
    (out * T(d['d_out'])).sum().backward()
    assert err(h.grad, d['dh']) < TOL
    assert err(layer.linear.weight.grad, d['dW']) < TOL * max(1.0, np.abs(d['dW']).max())
    assert err(layer.linear.bias.grad, d['db']) < TOL * max(1.0, np.abs(d['db']).max())


def test_update_all_surface():
    """Reference-style layer code written against the DGL surface (modules.py:218-227)."""
    import gist_amd.dgl_compat.function as fn
    d = np.load(os.path.join(GOLD, 'G1_layer_n257_ln1_act1.npz'))
    g = graph_of(d).local_var()
    h = T(d['h']).requires_grad_(True)
    norm = 1. / g.in_degrees().float().unsqueeze(1)
    norm[torch.isinf(norm)] = 0
    g.ndata['h'] = h
    g.update_all(fn.copy_src(src='h', out='m'), fn.sum(msg='m', out='h'))
    ah = g.ndata.pop('h') * norm
    ref = O.spmm_sum(d['rowptr'], d['col'], d['h'], out_scale=O.in_degree_norm(d['rowptr']))
    assert err(ah, ref) < TOL
    ah.sum().backward()
    t_rp, t_cl = O.transpose_csr(d['rowptr'], d['col'])
    gref = O.spmm_sum(t_rp, t_cl, np.ones_like(d['h']), src_scale=O.in_degree_norm(d['rowptr']))
    assert err(h.grad, gref) < TOL


def _params(d, prefix, n):
    return [(d['%sW%d' % (prefix, k)], d['%sb%d' % (prefix, k)]) for k in range(n)]


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLD, 'G2_model_*.npz'))),
                         ids=os.path.basename)
def test_G2_model_module_path(path):
    """nn.Module GCN + gist_amd.nn.CrossEntropyLoss + gist_amd.optim.Adam, reference loop shape."""
    from gist_amd.modules import GCN
    from gist_amd.nn import CrossEntropyLoss
    from gist_amd.optim import Adam
    d = np.load(path)
    L, S, H, ncls = int(d['L']), int(d['S']), int(d['H']), int(d['n_classes'])
    ln, wd = bool(d['use_layernorm']), float(d['wd'])
    torch.manual_seed(7)                                   # same seed as the generator
    if str(d['kind']) == 'full':
        model = GCN(d['feat'].shape[1], H, ncls, L, F.relu, 0.0, ln, False, False, 1, True)
    else:
        model = GCN(d['feat'].shape[1], H, ncls, L, F.relu, 0.0, ln, False, True, S, True)
    # same-seed initialisation equals the reference's (RNG call order, modules.py:201-216)
    for k, layer in enumerate(model.layers):
        assert np.array_equal(layer.linear.weight.detach().numpy(), d['init_W%d' % k])
        assert np.array_equal(layer.linear.bias.detach().numpy(), d['init_b%d' % k])
    model = model.to(DEV)
    g = graph_of(d)
    g.ndata['feat'] = T(d['feat'])
    labels = T(d['labels'])
    loss_f = CrossEntropyLoss()
    opt = Adam(model.parameters(), lr=0.01, weight_decay=wd)
    for step in range(3):
        opt.zero_grad()
        logits = model(g)
        loss = loss_f(logits, labels)
        loss.backward()
        if step == 0:
            assert err(logits, d['logits']) < TOL
            for k, layer in enumerate(model.layers):
                assert err(layer.linear.weight.grad, d['dW%d' % k]) < TOL
                assert err(layer.linear.bias.grad, d['db%d' % k]) < TOL
        assert abs(loss.item() - float(d['loss_step%d' % step])) < TOL
        opt.step()
        if step in (0, 2):
            for k, layer in enumerate(model.layers):
                assert err(layer.linear.weight, d['step%d_W%d' % (step + 1, k)]) < TOL
                assert err(layer.linear.bias, d['step%d_b%d' % (step + 1, k)]) < TOL


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLD, 'G2_model_*.npz'))),
                         ids=os.path.basename)
def test_G2_model_engine_path(path):
    from gist_amd.engine import SageEngine, Batch, dims_for
    from gist_amd import hip
    d = np.load(path)
    L, S, H, ncls = int(d['L']), int(d['S']), int(d['H']), int(d['n_classes'])
    ln, wd = bool(d['use_layernorm']), float(d['wd'])
    n, fin = d['feat'].shape
    dims = dims_for(fin, H, ncls, L, split_output=(str(d['kind']) == 'sub'), num_subnet=S)
    eng = SageEngine(dims, ln, 0.0, n_max=n + 3, device=torch.device(DEV))
    eng.arena.load(_params(d, 'init_', L + 1))
    g = graph_of(d)
    b = Batch()
    b.n, b.rowptr, b.col, b.t_rowptr, b.t_col = n, g.rowptr, g.col, g.t_rowptr, g.t_col
    b.norm, b.labels, b.ids = g.norm(), T(d['labels'], torch.int32), None
    for step in range(3):
        eng.z0_left(n).copy_(T(d['feat']))
        logits = eng.forward(b, training=True)
        if step == 0:
            assert err(logits, d['logits']) < TOL
        loss = eng.loss_and_backward(b)
        assert abs(loss.item() - float(d['loss_step%d' % step])) < TOL
        if step == 0:
            for k in range(L + 1):
                assert err(eng.arena.dW[k], d['dW%d' % k]) < TOL
                assert err(eng.arena.db[k], d['db%d' % k]) < TOL
        eng.adam_step(0.01, wd)
        if step in (0, 2):
            for k, (W, bb) in enumerate(eng.arena.export()):
                assert err(W, d['step%d_W%d' % (step + 1, k)]) < TOL
                assert err(bb, d['step%d_b%d' % (step + 1, k)]) < TOL


def _parts(d):
    return [d['part%d' % i] for i in range(int(d['psize']))]


def test_G3_sampler_device():
    """ClusterIter on the GPU yields the reference's batches (order, ids, induced CSR, ndata)."""
    from gist_amd.graph import Graph
    from gist_amd.sampler import ClusterIter
    d = np.load(os.path.join(GOLD, 'G3_sampler.npz'))
    g = Graph.from_edges(d['src'], d['dst'], int(d['n']))
    g.ndata['feat'] = torch.from_numpy(d['feat'])
    g.ndata['label'] = torch.from_numpy(d['label'])
    random.seed(int(d['seed']))
    it = ClusterIter('toy', g, int(d['psize']), int(d['batch_size']), d['train_nid'],
                     use_pp=False, par_li=_parts(d), device=torch.device(DEV))
    assert np.array_equal(it.g.rowptr.cpu().numpy(), d['train_rowptr'])
    assert np.array_equal(it.g.col.cpu().numpy(), d['train_col'])
    assert len(it) == int(d['n_batches'])
    for ep in range(2):
        for j, cluster in enumerate(it):
            assert np.array_equal(cluster.ndata['_ID'].cpu().numpy(), d['ep%d_b%d_nid' % (ep, j)])
            if ep == 0 and j == 0:
                assert np.array_equal(cluster.rowptr.cpu().numpy(), d['b0_rowptr'])
                assert np.array_equal(cluster.col.cpu().numpy(), d['b0_col'])
                assert np.array_equal(cluster.ndata['feat'].cpu().numpy(), d['b0_feat'])
                assert np.array_equal(cluster.ndata['label'].cpu().numpy(), d['b0_label'])
                # reversed CSR is the transpose of the induced CSR
                t_rp, t_cl = O.transpose_csr(d['b0_rowptr'], d['b0_col'])
                assert np.array_equal(cluster.t_rowptr.cpu().numpy(), t_rp)
                assert np.array_equal(np.sort(cluster.t_col.cpu().numpy()), np.sort(t_cl))


def test_dropout_statistics_and_backward_consistency():
    """Dropout cannot match torch's Philox stream; check that the layer with
    p > 0 equals the oracle layer given the SAME mask, forward and backward."""
    from gist_amd import autograd
    from tests.test_kernels_gpu import _dropout_mask_ref
    d = np.load(os.path.join(GOLD, 'G1_layer_n257_ln1_act1.npz'))
    g = graph_of(d)
    n, fin = d['h'].shape
    p, seed = 0.3, 99
    autograd._drop_counter[0] = 1000
    mask = _dropout_mask_ref(n, 2 * fin, p, seed, 1000).astype(np.float32)
    h = T(d['h']).requires_grad_(True)
    W = T(d['W']).requires_grad_(True)
    bb = T(d['b']).requires_grad_(True)
    out = autograd.sage_layer(g, h, W, bb, True, True, p, seed)
    ref, cache = O.sage_layer_forward(d['rowptr'], d['col'], d['h'], d['W'], d['b'], True, True,
                                      drop_mask=mask, drop_p=p)
    assert err(out, ref) < TOL
    (out * T(d['d_out'])).sum().backward()
    t_rp, t_cl = O.transpose_csr(d['rowptr'], d['col'])
    dh, dW, db = O.sage_layer_backward(cache, d['d_out'], t_rp, t_cl)
    assert err(h.grad, dh) < TOL
    assert err(W.grad, dW) < TOL * max(1.0, np.abs(dW).max())
    assert err(bb.grad, db) < TOL * max(1.0, np.abs(db).max())


def test_G5_cora_plumbing_graphconv():
    """BASELINE config 1 (gcn/gcn.py on a small graph): GraphConv stack + whole-tensor layer
    norm, forward vs the fixture and backward vs torch autograd of the same math."""
    import torch.nn.functional as Fn
    from gist_amd.gcn import GCN as SmallGCN
    d = np.load(os.path.join(GOLD, 'G5_graphconv.npz'))
    g = graph_of(d)
    model = SmallGCN(g, 7, 6, 3, 1, Fn.relu, 0.0, True).to(DEV)
    with torch.no_grad():
        for k, layer in enumerate(model.layers):
            layer.weight.copy_(T(d['W%d' % k]))
            layer.bias.copy_(T(d['b%d' % k]))
    model.eval()
    x = T(d['x']).requires_grad_(True)
    out = model(x)
    assert err(out, d['out']) < TOL
    out.sum().backward()
    # reference gradient by dense torch math on the CPU
    n = int(d['n'])
    A = torch.zeros(n, n)
    for s_, t_ in zip(d['src'], d['dst']):
        A[t_, s_] += 1
    ns = A.sum(0).clamp(min=1).pow(-0.5)
    nd = A.sum(1).clamp(min=1).pow(-0.5)
    An = nd[:, None] * A * ns[None, :]
    xc = torch.from_numpy(d['x']).requires_grad_(True)
    W0, b0 = torch.from_numpy(d['W0']), torch.from_numpy(d['b0'])
    W1, b1 = torch.from_numpy(d['W1']), torch.from_numpy(d['b1'])
    h = torch.relu(An @ (xc @ W0) + b0)          # in=7 > out=6: multiply by W first
    h = Fn.layer_norm(h, h.shape)
    o = (An @ (h @ W1) + b1)                     # in=6 > out=3
    assert err(out, o.detach().numpy()) < TOL
    o.sum().backward()
    assert err(x.grad, xc.grad.numpy()) < TOL
