"""GPU: the drop-in module path on the fused step (gist_amd/module_engine.py).

A loop shaped like the reference's (cluster_gcn/cluster_gcn.py:89-105: `pred = model(cluster)`, masked CE,
`optimizer.zero_grad()`, `loss.backward()`, `optimizer.step()`) must train to BITWISE the parameters, Adam moments and
per-step losses of the engine path (one gist_sage_step per iteration) -- it issues the same plan as three phase calls --
with dropout, with the next batch extracted inside the optimiser launch, over epoch boundaries.  Around it: what a
script may do differently (another loss on the logits, no zero_grad, a second optimiser, touching the cluster's structure)
against the op-by-op module path (GIST_MODULE_ENGINE=0's arithmetic) and the oracle.
"""
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _data(n_feats, seed=9, n=3000, blocks=30):
    from gist_amd import datasets
    return datasets.toy(seed=seed, n=n, n_blocks=blocks, n_feats=n_feats, n_classes=6, train_frac=1.0)


def _model(n_feats, hidden, n_layers, p_drop, S=1):
    from gist_amd.modules import GCN
    torch.manual_seed(3)
    if S == 1:
        m = GCN(n_feats, hidden, 6, n_layers, F.relu, p_drop, True, False, False, 1, True)
    else:
        m = GCN(n_feats, hidden, 6, n_layers, F.relu, p_drop, True, False, True, S, True)
    return m


def _module_run(p_drop, n_layers, n_feats, hidden, epochs=2, bs=5, wd=5e-4, loss_kind='gist', zero_grad=True,
                monkey=None):
    from gist_amd.nn import CrossEntropyLoss
    from gist_amd.optim import Adam
    from gist_amd.sampler import ClusterIter
    ds = _data(n_feats)
    g = ds.g
    random.seed(4)
    it = ClusterIter('toy', g, len(ds.par_li), bs, np.arange(g.number_of_nodes(), dtype=np.int64),
                     par_li=[p.copy() for p in ds.par_li], device=DEV)
    model = _model(n_feats, hidden, n_layers, p_drop)
    init = [(l.linear.weight.detach().clone(), l.linear.bias.detach().clone()) for l in model.layers]
    model.cuda()
    model.set_dropout_seed(11)
    loss_f = CrossEntropyLoss()
    opt = Adam(model.parameters(), lr=0.01, weight_decay=wd)
    losses = []
    for ep in range(epochs):
        for j, cluster in enumerate(it):
            cluster = cluster.to(torch.cuda.current_device())
            model.train()
            pred = model(cluster)
            batch_labels = cluster.ndata['label']
            batch_train_mask = cluster.ndata['train_mask']
            if loss_kind == 'gist':
                loss = loss_f(pred[batch_train_mask], batch_labels[batch_train_mask])
            else:
                loss = F.cross_entropy(pred[batch_train_mask], batch_labels[batch_train_mask].long())
            if zero_grad:
                opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
    return model, opt, it, torch.stack(losses), init


def _engine_run(p_drop, n_layers, n_feats, hidden, init, epochs=2, bs=5, wd=5e-4):
    from gist_amd.engine import SageEngine, dims_for
    from gist_amd.sampler import EngineClusterIter
    ds = _data(n_feats)
    g = ds.g
    random.seed(4)
    it = EngineClusterIter('toy', g, len(ds.par_li), bs, np.arange(g.number_of_nodes(), dtype=np.int64),
                           par_li=[p.copy() for p in ds.par_li], device=DEV)
    dims = dims_for(n_feats, hidden, 6, n_layers)
    eng = SageEngine(dims, True, p_drop, it.n_max, DEV, seed=11)
    eng.arena.load(init)
    eng.prefetch = True
    it.bind(eng)
    losses = []
    for ep in range(epochs):
        for b in it:
            losses.append(eng.train_step(b, 0.01, wd).clone())
    eng.check_extract()
    return eng, torch.stack(losses).flatten()


@pytest.mark.parametrize('p_drop,n_layers,n_feats,hidden', [
    (0.2, 2, 302, 512),
    (0.0, 2, 302, 512),
    (0.2, 4, 100, 256),
    (0.2, 1, 301, 64),
])
def test_module_loop_is_the_engine_path_bit_for_bit(p_drop, n_layers, n_feats, hidden):
    model, opt, it, losses, init = _module_run(p_drop, n_layers, n_feats, hidden)
    me = list(model._module_engines.values())[0]
    assert me, 'the model did not bind to its ClusterIter'
    eng, ref_losses = _engine_run(p_drop, n_layers, n_feats, hidden, init)
    it.engine.check_extract()
    assert torch.equal(losses, ref_losses), (losses - ref_losses).abs().max().item()
    A, B = me.engine.arena, eng.arena
    assert torch.equal(A.params, B.params), (A.params - B.params).abs().max().item()
    assert torch.equal(A.grads, B.grads)
    m, v = me.flat_state(opt)
    assert torch.equal(m, B.exp_avg) and torch.equal(v, B.exp_avg_sq)
    # the module's parameters ARE the arena
    for k, l in enumerate(model.layers):
        assert l.linear.weight.data_ptr() == A.W[k].data_ptr() and torch.equal(l.linear.weight, B.W[k])
        assert l.linear.weight.grad is not None and l.linear.weight.grad.data_ptr() == A.dW[k].data_ptr()
    # every batch but the first of an epoch was extracted inside the previous optimiser launch
    assert me.engine._prefetch_refused is not True


def test_any_loss_on_the_logits_goes_through_the_tape():
    """torch's own cross entropy on `pred` (int64 labels): an ordinary autograd graph ending in gist::gcn_backward with
    the caller's d_logits (GIST_STEP_DLOGITS_GIVEN).  Same mathematics as the fused loss: one step on one batch gives the
    same loss and the same gradients to rounding (parameters after Adam are not compared: its update is an
    ill-conditioned function of gradient elements near zero); a short run's losses stay together."""
    a = _module_run(0.0, 2, 302, 512, epochs=1, bs=30, loss_kind='torch')
    b = _module_run(0.0, 2, 302, 512, epochs=1, bs=30, loss_kind='gist')
    assert type(b[3]) is torch.Tensor
    assert abs(a[3][0].item() - b[3][0].item()) < 1e-5
    for la, lb in zip(a[0].layers, b[0].layers):
        for ga, gb in ((la.linear.weight.grad, lb.linear.weight.grad), (la.linear.bias.grad, lb.linear.bias.grad)):
            assert ((ga - gb).norm() / gb.norm()).item() < 1e-5
            assert (ga - gb).abs().max().item() < 1e-5 * gb.abs().max().item() + 1e-9
    a = _module_run(0.0, 2, 302, 512, epochs=1, loss_kind='torch')
    b = _module_run(0.0, 2, 302, 512, epochs=1, loss_kind='gist')
    assert ((a[3] - b[3]).abs() / b[3].abs()).max().item() < 1e-3


def test_first_step_gradients_match_the_op_by_op_module_path(monkeypatch):
    """One step, dropout 0: logits, loss and every p.grad of the bound model against the same model run layer by layer
    (gist::sage_layer ops on the eagerly built subgraph: the round-2 module path)."""
    from gist_amd.nn import CrossEntropyLoss
    from gist_amd.sampler import ClusterIter
    ds = _data(100)
    g = ds.g
    outs = []
    for engine_on in ('1', '0'):
        monkeypatch.setenv('GIST_MODULE_ENGINE', engine_on)
        random.seed(4)
        it = ClusterIter('toy', g, len(ds.par_li), 5, np.arange(g.number_of_nodes(), dtype=np.int64),
                         par_li=[p.copy() for p in ds.par_li], device=DEV)
        model = _model(100, 256, 3, 0.0).cuda()
        model.train()
        cluster = next(iter(it))
        pred = model(cluster)
        lab = cluster.ndata['label']
        loss = CrossEntropyLoss()(pred[cluster.ndata['train_mask']], lab[cluster.ndata['train_mask']])
        loss.backward()
        outs.append((pred.detach().clone(), loss.detach().clone(),
                     [p.grad.detach().clone() for p in model.parameters()], type(cluster).__name__,
                     bool(getattr(model, '_module_engines', None) and list(model._module_engines.values())[0])))
    on, off = outs
    assert on[4] and not off[4]
    assert on[3] == 'ClusterBatch' and off[3] == 'Graph'
    assert (on[0] - off[0]).abs().max().item() < 1e-4
    assert abs(on[1].item() - off[1].item()) < 1e-5
    for ga, gb in zip(on[2], off[2]):
        assert ((ga - gb).norm() / (gb.norm() + 1e-12)).item() < 1e-4


def test_backward_without_zero_grad_accumulates_like_torch():
    from gist_amd.nn import CrossEntropyLoss
    from gist_amd.sampler import ClusterIter
    ds = _data(100)
    g = ds.g
    random.seed(4)
    it = ClusterIter('toy', g, len(ds.par_li), 5, np.arange(g.number_of_nodes(), dtype=np.int64),
                     par_li=[p.copy() for p in ds.par_li], device=DEV)
    model = _model(100, 64, 2, 0.0).cuda()
    model.train()
    loss_f = CrossEntropyLoss()
    batches = iter(it)
    c1 = next(batches)
    loss_f(model(c1), c1.ndata['label']).backward()
    g1 = [p.grad.clone() for p in model.parameters()]
    c2 = next(batches)
    loss_f(model(c2), c2.ndata['label']).backward()          # no zero_grad: p.grad must now hold g1 + g2
    acc = [p.grad.clone() for p in model.parameters()]
    for p in model.parameters():
        p.grad = None
    loss_f(model(c2), c2.ndata['label']).backward()
    for a, x, y in zip(acc, g1, model.parameters()):
        assert torch.allclose(a, x + y.grad, rtol=1e-4, atol=1e-6)


def test_stale_forward_is_refused():
    from gist_amd.nn import CrossEntropyLoss
    from gist_amd.sampler import ClusterIter
    ds = _data(100)
    g = ds.g
    random.seed(4)
    it = ClusterIter('toy', g, len(ds.par_li), 5, np.arange(g.number_of_nodes(), dtype=np.int64),
                     par_li=[p.copy() for p in ds.par_li], device=DEV)
    model = _model(100, 64, 2, 0.0).cuda()
    model.train()
    batches = iter(it)
    c1, c2 = next(batches), next(batches)
    l1 = CrossEntropyLoss()(model(c1), c1.ndata['label'])
    model(c2)                                 # the engine's buffers now hold c2
    with pytest.raises(RuntimeError, match='no longer the model'):
        l1.backward()


def test_cluster_batch_is_the_eager_subgraph_for_any_other_consumer():
    """Structure and columns of a described batch, touched through the DGL surface, equal g.subgraph(ids)."""
    from gist_amd.sampler import ClusterIter
    ds = _data(40, n=1500, blocks=15)
    g = ds.g
    nid = np.arange(g.number_of_nodes(), dtype=np.int64)
    random.seed(2)
    it = ClusterIter('toy', g, len(ds.par_li), 3, nid, par_li=[p.copy() for p in ds.par_li], device=DEV)
    cluster = next(iter(it))
    assert type(cluster).__name__ == 'ClusterBatch'
    ids = it.batch_ids(0)
    ref = it.g.subgraph(ids)
    assert cluster.number_of_nodes() == ref.number_of_nodes()
    assert cluster.to(DEV) is cluster
    for k in ('feat', 'label', 'train_mask', 'val_mask', 'test_mask'):
        assert k in cluster.ndata
        assert torch.equal(cluster.ndata[k].to(ref.ndata[k].dtype), ref.ndata[k]), k
    assert cluster.ndata['label'].dtype == ref.ndata['label'].dtype
    pred = torch.randn(cluster.number_of_nodes(), 6, device=DEV)
    m = cluster.ndata['train_mask']
    assert pred[m] is pred and bool(m.all()) and m.dtype == torch.bool and int(m.sum()) == pred.shape[0]
    assert torch.equal(pred[:, 0][m], pred[:, 0])
    for a in ('rowptr', 'col', 't_rowptr', 't_col'):
        assert torch.equal(getattr(cluster, a), getattr(ref, a)), a
    assert cluster.number_of_edges() == ref.number_of_edges()
    assert torch.equal(cluster.in_degrees(), ref.in_degrees())
    assert torch.equal(cluster.norm(), ref.norm())
    lv = cluster.local_var()
    assert torch.equal(lv.ndata['feat'], ref.ndata['feat'])
    cpu = cluster.to('cpu')
    assert cpu.rowptr.device.type == 'cpu' and torch.equal(cpu.ndata['label'].to(DEV), ref.ndata['label'])


def test_dispatcher_ops_schema_and_fake():
    from torch.library import opcheck
    from gist_amd.sampler import ClusterIter
    ds = _data(100)
    g = ds.g
    random.seed(4)
    it = ClusterIter('toy', g, len(ds.par_li), 5, np.arange(g.number_of_nodes(), dtype=np.int64),
                     par_li=[p.copy() for p in ds.par_li], device=DEV)
    model = _model(100, 64, 2, 0.0).cuda()
    model.train()
    cluster = next(iter(it))
    with torch.no_grad():                   # (the pending forward's logits tensor must not be a tape output)
        model(cluster)
    me = list(model._module_engines.values())[0]
    assert 'gcn_forward' in str(torch.ops.gist.gcn_forward.default._schema)
    with torch.no_grad():
        args = ([p.detach() for p in me.params], me.handle, me.token, cluster.number_of_nodes(), me.ldc, True)
        opcheck(torch.ops.gist.gcn_forward.default, args, test_utils=('test_schema', 'test_faketensor'))
    # (a backward consumes its forward's activations in place: it runs once per forward, so no repeated-call checks)
    assert 'gcn_backward' in str(torch.ops.gist.gcn_backward.default._schema)
    views = torch.ops.gist.gcn_backward(me.engine.dlogits, me.handle, me.token, False)
    assert len(views) == len(me.params) and all(v.shape == p.shape for v, p in zip(views, me.params))
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.gist.gcn_forward([p.cpu() for p in me.params], me.handle, me.token, 4, me.ldc, True)


def test_eval_mode_forward_on_a_cluster_batch():
    from gist_amd.sampler import ClusterIter
    ds = _data(100)
    g = ds.g
    res = []
    for engine_on in (True, False):
        random.seed(4)
        it = ClusterIter('toy', g, len(ds.par_li), 5, np.arange(g.number_of_nodes(), dtype=np.int64),
                         par_li=[p.copy() for p in ds.par_li], device=DEV)
        model = _model(100, 64, 2, 0.5).cuda()
        model.eval()
        cluster = next(iter(it))
        with torch.no_grad():
            if engine_on:
                res.append(model(cluster).clone())
            else:
                sg = it.g.subgraph(it.batch_ids(0))
                res.append(model(sg).clone())
    assert (res[0] - res[1]).abs().max().item() < 1e-4


def _bound(n_feats=100, hidden=64, p_drop=0.0, bs=5, n=3000, blocks=30):
    from gist_amd.sampler import ClusterIter
    ds = _data(n_feats, n=n, blocks=blocks)
    g = ds.g
    random.seed(4)
    it = ClusterIter('toy', g, len(ds.par_li), bs, np.arange(g.number_of_nodes(), dtype=np.int64),
                     par_li=[p.copy() for p in ds.par_li], device=DEV)
    return _model(n_feats, hidden, 2, p_drop).cuda(), it


def test_every_call_returns_its_own_tensors():
    """torch semantics: what a call returned stays what it was.  `preds.append(model(c))` over an epoch, losses kept
    across many steps and eval-mode logits must not be overwritten by later forwards (the engine's rings skip a slot
    whose tensor -- or a view of it -- is still held); pred[mask] may be the same object, its values may not change."""
    from gist_amd.nn import CrossEntropyLoss
    from gist_amd.optim import Adam
    model, it = _bound()
    loss_f = CrossEntropyLoss()
    opt = Adam(model.parameters(), lr=0.01)
    model.train()
    kept, copies, losses, loss_copies, views = [], [], [], [], []
    for ep in range(14):                      # 84 steps: more than the loss ring's 64 slots
        for cluster in it:
            pred = model(cluster)
            loss = loss_f(pred[cluster.ndata['train_mask']], cluster.ndata['label'][cluster.ndata['train_mask']])
            opt.zero_grad()
            loss.backward()
            opt.step()
            if len(kept) < 12:
                kept.append(pred)
                copies.append(pred.detach().clone())
                views.append(pred[:7, :3])    # a view alone must hold its slot too
            losses.append(loss)
            loss_copies.append(float(loss.detach()))
    for a, b in zip(kept, copies):
        assert torch.equal(a.detach(), b)
    for v, b in zip(views, copies):
        assert torch.equal(v.detach(), b[:7, :3])
    assert [float(x.detach()) for x in losses] == loss_copies
    d = losses[0].detach()
    assert d.data_ptr() != losses[0].data_ptr()          # detach() copies out of the ring
    model.eval()
    with torch.no_grad():
        outs = [model(c) for c in it]
        again = [model(c).clone() for c in it]
    # (the epochs shuffle: compare each call with a clone taken when it was made)
    with torch.no_grad():
        held, ref = [], []
        for c in it:
            y = model(c)
            held.append(y)
            ref.append(y.clone())
    for a, b in zip(held, ref):
        assert torch.equal(a, b)
    assert len({o.data_ptr() for o in outs}) == len(outs) and len(again) == len(outs)


def test_dropped_models_give_their_memory_back():
    """Two bound models created and dropped in one process: the arena, the activation buffers, the rings and the
    batcher's copies must be freed by reference counting alone -- also after gc.freeze(), which an application may have
    called (bench.py does): a Parameter <-> engine cycle would then never be collected."""
    import gc
    from gist_amd.nn import CrossEntropyLoss
    from gist_amd.optim import Adam

    def one():
        model, it = _bound(hidden=256)
        loss_f = CrossEntropyLoss()
        opt = Adam(model.parameters(), lr=0.01)
        model.train()
        for cluster in it:
            pred = model(cluster)
            loss = loss_f(pred, cluster.ndata['label'])
            opt.zero_grad()
            loss.backward()
            opt.step()
        float(loss.detach())
        if it.engine is not None:
            it.engine.check_extract()

    one()                                # (first use: lazy imports, allocator pools)
    gc.collect()
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    gc.disable()                         # reference counting only
    try:
        one()
        gc.freeze()
        one()
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() <= base + (1 << 20), (torch.cuda.memory_allocated(), base)
    finally:
        gc.unfreeze()
        gc.enable()


def test_replacing_any_parameter_is_noticed():
    """homed() looks at every parameter: a middle layer's weight given new storage must come back into the arena (the fused
    step would otherwise train a stale copy while the visible parameter never changes)."""
    from gist_amd.nn import CrossEntropyLoss
    from gist_amd.optim import Adam
    model, it = _bound()
    loss_f = CrossEntropyLoss()
    opt = Adam(model.parameters(), lr=0.01)
    model.train()
    batches = iter(it)
    c = next(batches)
    loss_f(model(c), c.ndata['label']).backward()
    opt.step()
    me = next(iter(model.__dict__['_module_engines'].values()))
    assert me.homed()
    new_w = torch.full_like(model.layers[1].linear.weight.data, 0.01)
    model.layers[1].linear.weight.data = new_w                 # not the first, not the last parameter
    assert not me.homed()
    c = next(batches)
    opt.zero_grad()
    loss_f(model(c), c.ndata['label']).backward()
    assert me.homed() and model.layers[1].linear.weight.data_ptr() == me.engine.arena.W[1].data_ptr()
    assert torch.equal(me.engine.arena.W[1], new_w)            # the assigned values are what the step used
    opt.step()
    assert not torch.equal(model.layers[1].linear.weight.data, new_w)      # and what the optimiser updated


def test_epochs_left_early_do_not_outrun_the_gpu():
    """A loop that breaks out of every epoch never reaches the deferred check at the epoch's end; the iterator must still
    keep the host within one epoch of the GPU before it rewrites a staging buffer (ids of later batches would be wrong)."""
    from gist_amd.nn import CrossEntropyLoss
    from gist_amd.optim import Adam
    model, it = _bound()
    loss_f = CrossEntropyLoss()
    opt = Adam(model.parameters(), lr=0.01)
    model.train()
    marks = []
    for ep in range(40):
        for j, cluster in enumerate(it):
            pred = model(cluster)
            loss = loss_f(pred, cluster.ndata['label'])
            opt.zero_grad()
            loss.backward()
            opt.step()
            if j == 1:
                break
        marks.append(getattr(it.engine, '_mark_tag', 0))
    assert marks[-1] >= 38          # one progress mark per truncated epoch
    it.engine.check_extract()
    assert torch.isfinite(loss).item()
