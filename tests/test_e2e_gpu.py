"""GPU end to end: whole training runs on the HIP path against runs of the REFERENCE's
own loops recorded in tests/golden/G6_e2e_*.npz
  * cluster_gcn/cluster_gcn.py main()                -> ClusterGCNTrainer
  * cluster_gcn/cluster_gcn_ist_distrib.py train()   -> gist_amd.ist.train with all S
    sites in one process on one GPU (LocalCommGroup), HIP block kernels for sync/dispatch
Checks per-iteration losses, base model after every sync, the event schedule (epoch-0
no-redispatch quirk), accuracies.  1e-4 fp32."""
import argparse
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = torch.device('cuda', 0)
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _params(d, prefix, n):
    return [(d['%sW%d' % (prefix, k)], d['%sb%d' % (prefix, k)]) for k in range(n)]


def _graph(d):
    from gist_amd.graph import Graph
    g = Graph.from_edges(d['src'], d['dst'], int(d['n']))
    g.ndata['feat'] = torch.from_numpy(d['feat'])
    g.ndata['label'] = torch.from_numpy(d['label'])
    for m in ('train_mask', 'val_mask', 'test_mask'):
        g.ndata[m] = torch.from_numpy(d[m])
    return g


def _parts(d):
    return [d['part%d' % i] for i in range(int(d['psize']))]


def test_e2e_single_gpu_cluster_gcn():
    from gist_amd.trainer import ClusterGCNTrainer
    d = np.load(os.path.join(GOLD, 'G6_e2e_single.npz'))
    g = _graph(d)
    L = int(d['n_layers'])
    random.seed(int(d['rnd_seed']))
    tr = ClusterGCNTrainer('toy', g, _parts(d), int(d['psize']), int(d['batch_size']),
                           int(d['n_hidden']), L, int(d['n_classes']), 0.0, True, float(d['lr']),
                           0.0, DEV, init_params=_params(d, 'init_', L + 1))
    val_accs, test_accs = [], []
    for e in range(int(d['n_epochs'])):
        tr.timed_epoch()
        for k, (W, b) in enumerate(tr.engine.arena.export()):
            assert np.abs(W - d['ep%d_W%d' % (e, k)]).max() < TOL, (e, k)
            assert np.abs(b - d['ep%d_b%d' % (e, k)]).max() < TOL, (e, k)
        val_accs.append(tr.evaluate('val_mask'))
        test_accs.append(tr.evaluate('test_mask'))
    assert np.allclose(val_accs, d['val_accs'], atol=1e-6)
    assert abs(val_accs[-1] - float(d['last_val'])) < 1e-4
    assert abs(max(val_accs) - float(d['best_val'])) < 1e-4
    assert abs(test_accs[-1] - float(d['last_test'])) < 1e-4
    assert abs(max(test_accs) - float(d['best_test'])) < 1e-4
    assert tr.total_time > 0


@pytest.mark.parametrize('mode', ['bf16x3', 'f16x3'])
def test_e2e_golden_run_forced_onto_split_gemm(mode):
    """The reference's recorded cluster_gcn.py run (G6) with EVERY projection of every step forced
    onto the split GEMM kernels (thresholds lifted through the tuning hooks; the shapes are tiny:
    k = 2 * 16): parameters after every epoch and the accuracies must match the golden run like
    the fp32 kernel does -- 1e-4 for bf16x3 (all 24 operand bits) and for f16x3."""
    from gist_amd import hip, _lib
    from gist_amd.trainer import ClusterGCNTrainer
    d = np.load(os.path.join(GOLD, 'G6_e2e_single.npz'))
    g = _graph(d)
    L = int(d['n_layers'])
    prev = hip.gemm_mode()
    hip.gemm_mode(mode)
    hip.tuning('h3_min_gflop', 1e-9)
    hip.tuning('h3_min_tiles', 1)
    try:
        assert _lib.load().gist_gemm_workspace_bytes(60, int(d['n_hidden']), 2 * int(d['n_hidden'])) > 0
        random.seed(int(d['rnd_seed']))
        tr = ClusterGCNTrainer('toy', g, _parts(d), int(d['psize']), int(d['batch_size']),
                               int(d['n_hidden']), L, int(d['n_classes']), 0.0, True, float(d['lr']),
                               0.0, DEV, init_params=_params(d, 'init_', L + 1))
        tr.engine.enable_timer(4096)
        val_accs = []
        for e in range(int(d['n_epochs'])):
            tr.timed_epoch()
            for k, (W, b) in enumerate(tr.engine.arena.export()):
                assert np.abs(W - d['ep%d_W%d' % (e, k)]).max() < TOL, (mode, e, k)
                assert np.abs(b - d['ep%d_b%d' % (e, k)]).max() < TOL, (mode, e, k)
            val_accs.append(tr.evaluate('val_mask'))
        rec = tr.engine.read_timer()
        assert sum(1 for r in rec if r[1] == 2) >= 3 * (L + 1)      # kind 2: split main kernels ran
        tr.engine.disable_timer()
        assert np.allclose(val_accs, d['val_accs'], atol=1e-6)
    finally:
        hip.tuning('h3_min_gflop', 0)
        hip.tuning('h3_min_tiles', 0)
        hip.gemm_mode(prev)


@pytest.mark.parametrize('S', [2, 4])
def test_e2e_gist_in_process(S):
    from gist_amd import ist
    from gist_amd.sampler import EngineClusterIter
    from gist_amd.trainer import FullGraphEvaluator
    d = np.load(os.path.join(GOLD, 'G6_e2e_ist_S%d.npz' % S))
    g = _graph(d)
    L, H = int(d['n_layers']), int(d['n_hidden'])
    fin, ncls = d['feat'].shape[1], int(d['n_classes'])
    random.seed(int(d['rnd_seed']))
    train_nid = np.nonzero(d['train_mask'])[0].astype(np.int64)
    # same order as the reference's main(): ClusterIter first, then the wrapper
    it = EngineClusterIter('toy', g, int(d['psize']), int(d['batch_size']), train_nid,
                           par_li=_parts(d), device=DEV)
    group = ist.LocalCommGroup(S)
    models = []
    for r in range(S):
        args = argparse.Namespace(num_subnet=S, n_hidden=H, n_layers=L, rank=r, dropout=0.0,
                                  use_layernorm=True, lr=float(d['lr']), weight_decay=0.0,
                                  iter_per_site=int(d['iter_per_site']),
                                  n_epochs=int(d['n_epochs']))
        models.append(ist.DistributedGNNWrapper(
            args, None, fin, ncls, DEV,
            base_init=_params(d, 'r0_base_init_', L + 1) if r == 0 else None,
            comm=group.handle(r), n_max=it.n_max))
    part = models[0].sample_partitions()
    for m in models:
        m.ini_sync_dispatch_model(part)
    for r, m in enumerate(models):
        for k in range(L + 1):
            assert np.array_equal(m.sub.W[k].cpu().numpy(), d['r%d_sub_init_W%d' % (r, k)])
            assert np.array_equal(m.sub.b[k].cpu().numpy(), d['r%d_sub_init_b%d' % (r, k)])
    it.bind(models[0].engine)
    evaluator = FullGraphEvaluator(g, models[0].base_dims, True, models[0].base, DEV)
    snaps = []
    orig_apply = models[0].sync_apply

    def spy_apply():
        orig_apply()
        snaps.append(models[0].base.export())
    models[0].sync_apply = spy_apply
    res = ist.train(models, models[0].args, it, evaluator=evaluator, log=lambda *a: None)
    gold = [str(e) for e in d['r0_events']]
    dedup = [e for i, e in enumerate(gold) if not (e == 'eval' and gold[i - 1] == 'eval')]
    assert res['events'] == dedup
    for r in range(S):
        got = np.array([float(x.item()) for x in res['losses'][r]])
        assert np.abs(got - d['r%d_losses' % r]).max() < TOL, r
    assert len(snaps) == int(d['r0_n_syncs'])
    for i, snap in enumerate(snaps):
        for k, (W, b) in enumerate(snap):
            assert np.abs(W - d['r0_sync%d_W%d' % (i, k)]).max() < TOL, (i, k)
            assert np.abs(b - d['r0_sync%d_b%d' % (i, k)]).max() < TOL, (i, k)
    # every site's base replica is identical
    for m in models[1:]:
        assert torch.equal(m.base.params, models[0].base.params)
    tail = dict(zip([str(k) for k in d['r0_tail_keys']], d['r0_tail_vals']))
    assert abs(res['val_accs'][-1] - tail['Last Val']) < 1e-4
    assert abs(max(res['val_accs']) - tail['Best Val']) < 1e-4
    assert abs(res['test_accs'][-1] - tail['Last Test']) < 1e-4
    assert abs(max(res['test_accs']) - tail['Best Test']) < 1e-4
    lines = []
    ist.print_results(res, log=lines.append)
    assert [l.split(':')[0] for l in lines] == [str(k) for k in d['r0_tail_keys']]


def test_full_size_step_properties():
    """BASELINE.json's full size (Reddit-like batch, H=4096): size-independent properties
    instead of an oracle run -- (1) linearity of the SpMM in x, (2) gather/forward/backward
    determinism (two identical steps from identical state give bitwise identical arenas),
    (3) dropout off: loss equals the CE of the logits recomputed by torch on the device."""
    from gist_amd import datasets, hip
    from gist_amd.engine import SageEngine, dims_for
    from gist_amd.sampler import EngineClusterIter
    random.seed(0)
    ds = datasets.reddit_synth(seed=0)
    g = ds.g
    it = EngineClusterIter('reddit-synth', g, len(ds.par_li), 20,
                           np.arange(g.number_of_nodes(), dtype=np.int64),
                           par_li=[p.copy() for p in ds.par_li], device=DEV)
    dims = dims_for(602, 4096, 41, 2)
    eng = SageEngine(dims, True, 0.0, it.n_max, DEV)
    gen = torch.Generator(device='cpu').manual_seed(0)
    for k, (i, o) in enumerate(dims):
        eng.arena.W[k].copy_((torch.rand(o, 2 * i, generator=gen) - 0.5) * (2.0 / np.sqrt(2 * i)))
        eng.arena.b[k].copy_((torch.rand(o, generator=gen) - 0.5) * (2.0 / np.sqrt(2 * i)))
    it.bind(eng, native=False)          # explicit extraction: the batch buffers are filled now
    start = eng.arena.params.clone()
    batch = next(iter(it))
    n = batch.n
    assert batch.ready and 1900 < n <= it.n_max
    assert int(batch.rowptr[n].item()) > 100000          # ~1.3e5 in-batch edges
    # (1) SpMM linearity: A(2x + y) == 2 A x + A y (exactly representable scaling)
    x = torch.randn(n, 256, device=DEV)
    y = torch.randn(n, 256, device=DEV)
    o1, o2, o3 = (torch.empty(n, 256, device=DEV) for _ in range(3))
    hip.spmm(batch.rowptr, batch.col, 2 * x + y, o1)
    hip.spmm(batch.rowptr, batch.col, x, o2)
    hip.spmm(batch.rowptr, batch.col, y, o3)
    assert (o1 - (2 * o2 + o3)).abs().max().item() < 1e-3
    # row sums of A = in-degree
    ones = torch.ones(n, 4, device=DEV)
    deg = torch.empty(n, 4, device=DEV)
    hip.spmm(batch.rowptr, batch.col, ones, deg)
    assert torch.equal(deg[:, 0].to(torch.int32), batch.rowptr[1:n + 1] - batch.rowptr[:n])
    # (3) loss == CE of the logits
    loss = eng.train_step(batch, 0.01).clone()
    logits = eng.logits(n).clone()
    ref = torch.nn.functional.cross_entropy(logits, batch.labels.long())
    assert abs(loss.item() - ref.item()) < 1e-4
    after1 = eng.arena.params.clone()
    # (2) determinism: rewind and repeat the identical step
    eng.arena.params.copy_(start)
    eng.arena.reset_optimizer()
    hip.gather_rows(it.batcher.feat, batch.ids, eng.z0_left(n))
    eng.train_step(batch, 0.01)
    assert torch.equal(eng.arena.params, after1)
    assert torch.isfinite(eng.arena.params).all()


@pytest.mark.parametrize('n_layers,use_ln,p_drop,n_feats,n_hidden', [
    (3, True, 0.2, 50, 96), (1, False, 0.2, 50, 96), (2, True, 0.0, 50, 96),
    (2, True, 0.2, 302, 512),       # wide rows: workgroup-per-row SpMM, VEC2 features
    (3, False, 0.5, 301, 1024),     # odd feature width (scalar loads), no LayerNorm
    (2, True, 0.2, 64, 2048),       # split-K dZ GEMM: mask applied in the slab reduction
])
def test_native_step_equals_op_by_op(n_layers, use_ln, p_drop, n_feats, n_hidden):
    """gist_sage_step (one C-ABI call per iteration) issues the same kernels in the same
    order as the Python op-by-op path: parameters after 4 steps with dropout must be
    BITWISE identical, and the native HIP-event timer must see 5*(L+1)-3 launches/step (layer 0 aggregates inside the extraction).
    (GEMM mode f32: in mode f16x3 the step keeps its own split operands under different
    scales than a per-call split -- test_step_kept_split_operands_equal_per_call_splits.)"""
    from gist_amd import datasets, hip
    from gist_amd.engine import SageEngine, dims_for
    from gist_amd.sampler import EngineClusterIter
    prev_mode = hip.gemm_mode()
    hip.gemm_mode('f32')
    try:
        _native_vs_op_by_op(n_layers, use_ln, p_drop, n_feats, n_hidden)
    finally:
        hip.gemm_mode(prev_mode)


def _native_vs_op_by_op(n_layers, use_ln, p_drop, n_feats, n_hidden):
    from gist_amd import datasets
    from gist_amd.engine import SageEngine, dims_for
    from gist_amd.sampler import EngineClusterIter
    ds = datasets.toy(seed=9, n=3000, n_blocks=30, n_feats=n_feats, n_classes=6, train_frac=1.0)
    g = ds.g
    nid = np.arange(g.number_of_nodes(), dtype=np.int64)
    dims = dims_for(n_feats, n_hidden, 6, n_layers)
    results = []
    for native in (True, False):
        random.seed(4)
        it = EngineClusterIter('toy', g, len(ds.par_li), 5, nid, par_li=[p.copy() for p in ds.par_li],
                               device=DEV)
        eng = SageEngine(dims, use_ln, p_drop, it.n_max, DEV, seed=11)
        gen = torch.Generator().manual_seed(1)
        for k, (i, o) in enumerate(dims):
            eng.arena.W[k].copy_((torch.rand(o, 2 * i, generator=gen) - 0.5) * 0.3)
            eng.arena.b[k].copy_((torch.rand(o, generator=gen) - 0.5) * 0.3)
        it.bind(eng, native=native)
        assert (eng.plan is not None) == native
        if native:
            eng.enable_timer(1000)
        losses = []
        for j, b in enumerate(it):
            losses.append(eng.train_step(b, 0.01, 5e-4).clone())
            if j == 3:
                break
        if native:
            rec = eng.read_timer()
            # (round 4: the class layer's projection and dZ are ONE launch -- gist_class_layer_f32 -- where it is taken)
            k_cls, c_cls = 2 * dims[-1][0], dims[-1][1]
            cls_fused = k_cls % 64 == 0 and k_cls <= 1024 and c_cls <= 48
            # (and dZ + dW of a narrow hidden layer are ONE launch -- gist_gemm_nn_tn_dual_f32 -- where the shapes are
            # taken: up to len(dims) - 2 records fewer per step)
            # (and layer 0's aggregation comes with the extraction -- gist_extract_parts_desc.feat_intra: one record fewer)
            assert it.batcher.feat_intra is not None and eng.plan.feat_intra
            # (and the fused class layer's dW slabs ride in the grid of the LayerNorm backward below it --
            # gist_ln_relu_bwd_colsum_class_dw_f32, one-kernel backward up to 1024 columns: their record is gone too)
            dw_in_ln = cls_fused and len(dims) >= 2 and dims[-2][1] <= 1024 and dims[-2][1] % 4 == 0
            full = 4 * (5 * len(dims) - 3 - (1 if cls_fused else 0) - (1 if dw_in_ln else 0))
            assert full - 4 * max(len(dims) - 2, 0) <= len(rec) <= full, (len(rec), full)
            assert all(ms > 0 for ms, *_ in rec)
            eng.disable_timer()
        results.append((eng.arena.params.clone(), torch.stack(losses)))
    assert torch.equal(results[0][0], results[1][0])
    assert torch.equal(results[0][1], results[1][1])


@pytest.mark.parametrize('split', ['f16x3', 'bf16x3'])
@pytest.mark.parametrize('n_layers,n_feats,n_hidden', [(2, 64, 2048), (3, 302, 2048)])
def test_native_step_kept_splits_small_batch(n_layers, n_feats, n_hidden, split):
    """A ~500-row batch at width 2048 (projections of 4-8 GFLOP: above the kept-split threshold,
    below the per-call one; batch rows not a multiple of 32, so the transposed splits are
    zero-padded along k): the native step in GEMM mode f16x3 / bf16x3 (the latter with its tile
    threshold lowered through the tuning hooks: 256 x 128 tiles, this batch has 32 of them) against
    the same step in mode f32, same dropout stream -- losses of 4 steps within 1e-4 (of their magnitude: this toy run
    diverges to losses ~20), parameters close."""
    from gist_amd import datasets, hip
    from gist_amd.engine import SageEngine, dims_for
    from gist_amd.sampler import EngineClusterIter
    ds = datasets.toy(seed=9, n=3000, n_blocks=30, n_feats=n_feats, n_classes=6, train_frac=1.0)
    g = ds.g
    nid = np.arange(g.number_of_nodes(), dtype=np.int64)
    dims = dims_for(n_feats, n_hidden, 6, n_layers)
    prev = hip.gemm_mode()
    out = {}
    try:
        for mode in ('f32', split):
            hip.gemm_mode(mode)
            if mode == 'bf16x3':
                hip.tuning('h3_min_tiles', 1)
                hip.tuning('h3_min_gflop', 0.5)
            random.seed(4)
            it = EngineClusterIter('toy', g, len(ds.par_li), 5, nid,
                                   par_li=[p.copy() for p in ds.par_li], device=DEV)
            eng = SageEngine(dims, True, 0.2, it.n_max, DEV, seed=11)
            gen = torch.Generator().manual_seed(1)
            for k, (i, o) in enumerate(dims):
                s_ = 1.0 / np.sqrt(2 * i)
                eng.arena.W[k].copy_((torch.rand(o, 2 * i, generator=gen) - 0.5) * 2 * s_)
                eng.arena.b[k].copy_((torch.rand(o, generator=gen) - 0.5) * 2 * s_)
            it.bind(eng)
            assert (eng.plan.h3_workspace is not None) == (mode != 'f32')
            losses = []
            for j, b in enumerate(it):
                assert b.n % 32 != 0 or j > 0
                losses.append(float(eng.train_step(b, 0.01, 5e-4).item()))
                if j == 3:
                    break
            out[mode] = (losses, eng.arena.params.clone())
    finally:
        hip.tuning('h3_min_tiles', 0)
        hip.tuning('h3_min_gflop', 0)
        hip.gemm_mode(prev)
    la, lb = out['f32'][0], out[split][0]
    assert all(abs(a - b) <= TOL * max(1.0, abs(a)) for a, b in zip(la, lb)), (la, lb)
    d = (out['f32'][1] - out[split][1]).abs()
    assert d.mean().item() < 2e-5 and (d > TOL).float().mean().item() < 3e-2, \
        (d.mean().item(), (d > TOL).float().mean().item(), d.max().item())


def test_ist_wide_block_algebra_properties():
    """Ultra-wide-style IST (cluster_gcn_ist_ultra_wide.py path, base model resident in HBM)
    at H=8192, S=8: (1) every rank's sub-model blocks are disjoint and tile the 'diagonal',
    (2) dispatch -> sync with no training leaves the base model BIT-identical, (3) after a
    perturbation only the S diagonal blocks (next_s x full_prev_s) change -- the S^2-S
    off-diagonal blocks keep their old values, (4) the shared
    last bias becomes the mean of the S copies."""
    from gist_amd import ist
    S, H, L, F, C = 8, 8192, 2, 602, 41
    random.seed(5)
    group = ist.LocalCommGroup(S)
    models = []
    gen = torch.Generator().manual_seed(0)
    base_init = None
    for r in range(S):
        args = argparse.Namespace(num_subnet=S, n_hidden=H, n_layers=L, rank=r, dropout=0.0,
                                  use_layernorm=True)
        if r == 0:
            from gist_amd.engine import dims_for
            base_init = [(torch.rand(o, 2 * i, generator=gen), torch.rand(o, generator=gen))
                         for (i, o) in dims_for(F, H, C, L)]
        models.append(ist.DistributedGNNWrapper(args, None, F, C, DEV,
                                                base_init=base_init if r == 0 else None,
                                                comm=group.handle(r)))
    part = models[0].sample_partitions()
    for m in models:
        m.ini_sync_dispatch_model(part)
    base0 = models[0].base.params.clone()
    for m in models[1:]:
        assert torch.equal(m.base.params, base0)           # replicas
    # (1) index sets are a partition of [0, H)
    for l in range(L):
        allidx = torch.cat([part[l][s][0] for s in range(S)])
        assert torch.equal(torch.sort(allidx).values, torch.arange(H))
    # (2) identity
    for m in models:
        m.sync_gather()
    for m in models:
        m.sync_apply()
    assert torch.equal(models[0].base.params, base0)
    # (3) perturb every sub-model, sync, compare block structure on W1 [H, 2H]
    for r, m in enumerate(models):
        m.sub.params.add_(1.0 + r)
    for m in models:
        m.sync_gather()
    for m in models:
        m.sync_apply()
    W1_old = base0[models[0].base.offsets[1][0]:models[0].base.offsets[1][1]].view(H, 2 * H)
    W1_new = models[0].base.W[1]
    changed = (W1_new != W1_old)
    expect = torch.zeros(H, 2 * H, dtype=torch.bool, device=DEV)
    for s in range(S):
        rows = part[1][s][0].to(DEV)
        cols = part[0][s][1].to(DEV)
        expect[rows[:, None], cols[None, :]] = True
        blk = W1_new[rows[:, None], cols[None, :]] - W1_old[rows[:, None], cols[None, :]]
        assert torch.allclose(blk, torch.full_like(blk, 1.0 + s))
    assert torch.equal(changed, expect)
    assert int(expect.sum().item()) == S * (H // S) * (2 * H // S)
    # (4) shared bias = mean over sites of (old + 1 + s)
    bL_old = base0[models[0].base.offsets[L][1]:models[0].base.offsets[L][1] + C]
    assert torch.allclose(models[0].base.b[L], bL_old + 1.0 + (S - 1) / 2.0, atol=1e-5)
    for m in models:
        assert torch.equal(m.sub.b[L], models[0].base.b[L])


def test_config5_ultra_wide_H32768_S8():
    """BASELINE config 5 at its full size on one GPU: n_hidden 32768, 8 sub-GCNs of width 4096,
    n_layers 2 -- an 8.8 GB base model per site, resident in HBM (cluster_gcn_ist_ultra_wide.py
    keeps it on the host, :84).  (1) dispatch -> sync without training is the bit-exact identity
    on every replica; (2) one training step of a 4096-wide sub-GCN on its dispatched weights
    equals the oracle's; (3) after a perturbation of every sub-model only the 8 diagonal blocks
    of W1 [32768, 65536] change, by exactly the perturbation, and the shared class bias becomes
    the mean; (4) a re-dispatch hands every site the block of the synced base it indexes."""
    from gist_amd import datasets, ist
    from gist_amd.engine import dims_for
    from gist_amd.sampler import EngineClusterIter
    from oracle import gist_oracle as O
    from oracle import train_oracle as TO
    S, H, L, F, C = 8, 32768, 2, 602, 41
    random.seed(6)
    ds = datasets.make_block_dataset('uw-toy', 6000, 60, F, C, intra_deg=6, inter_deg=2, seed=13)
    g = ds.g
    it = EngineClusterIter(ds.name, g, len(ds.par_li), 5, np.arange(g.number_of_nodes(), dtype=np.int64),
                           par_li=[p.copy() for p in ds.par_li], device=DEV)
    group = ist.LocalCommGroup(S)
    gen = torch.Generator(device=DEV).manual_seed(0)
    base_dims = dims_for(F, H, C, L)
    base_init = [((torch.rand(o, 2 * i, device=DEV, generator=gen) * 2 - 1) / np.sqrt(2 * i),
                  (torch.rand(o, device=DEV, generator=gen) * 2 - 1) / np.sqrt(2 * i))
                 for (i, o) in base_dims]
    models = []
    for r in range(S):
        args = argparse.Namespace(num_subnet=S, n_hidden=H, n_layers=L, rank=r, dropout=0.0,
                                  use_layernorm=True)
        models.append(ist.DistributedGNNWrapper(args, None, F, C, DEV,
                                                base_init=base_init if r == 0 else None,
                                                comm=group.handle(r),
                                                n_max=it.n_max if r == 0 else None))
    del base_init
    assert models[0].base.numel * 4 > 8.7e9 and models[0].sub_dims == dims_for(F, 4096, C, L)
    part = models[0].sample_partitions()
    for m in models:
        m.ini_sync_dispatch_model(part)
    base0 = models[0].base.params.clone()
    # (1) identity
    for m in models:
        m.sync_gather()
    for m in models:
        m.sync_apply()
    for m in models:
        assert torch.equal(m.base.params, base0)
    # (2) one sub-model step (h = 4096) against the oracle, then put the weights back
    m0 = models[0]
    saved = m0.sub.params.clone()
    params = m0.sub.export()
    it.bind(m0.engine)
    batch = next(iter(it))
    loss = m0.engine.train_step(batch, 0.01, 0.0)
    logits = m0.engine.logits(batch.n).cpu().numpy()
    tg = TO.TrainGraph(g.rowptr.numpy().astype(np.int64), g.col.numpy().astype(np.int64),
                       g.ndata['feat'].numpy(), g.ndata['label'].numpy().astype(np.int64))
    b = tg.batch(it.batch_ids(0))
    opt = O.new_opt_state(params)
    ref_loss, ref_logits, _ = O.train_step(b[0], b[1], b[2], b[3], b[4], b[5], params, opt, True, 0.01)
    assert abs(float(loss.item()) - float(ref_loss)) < TOL
    assert np.abs(logits - ref_logits).max() <= TOL * max(1.0, np.abs(ref_logits).max())
    errs = np.concatenate([np.abs(W - Wr).ravel() for (W, _), (Wr, _) in zip(m0.sub.export(), params)])
    stats = (float(errs.mean()), float((errs > TOL).mean()), float(errs.max()))
    _record_parity('config5_sub_step_vs_oracle', 'bf16x3', 4096, stats)
    assert stats[2] <= 2 * 0.01 + 1e-6
    m0.sub.params.copy_(saved)
    m0.sub.reset_optimizer()
    del saved, params, errs
    # (3) perturb, sync, block structure of W1 [H, 2H]
    for r, m in enumerate(models):
        m.sub.params.add_(1.0 + r)
    for m in models:
        m.sync_gather()
    for m in models:
        m.sync_apply()
    for m in models[1:]:
        assert torch.equal(m.base.params, models[0].base.params)
    w0, w1 = models[0].base.offsets[1]
    W1_old = base0[w0:w1].view(H, 2 * H)
    W1_new = models[0].base.W[1]
    expect = torch.zeros(H, 2 * H, dtype=torch.bool, device=DEV)
    for s in range(S):
        rows = part[1][s][0].to(DEV)
        cols = part[0][s][1].to(DEV)
        expect[rows[:, None], cols[None, :]] = True
        blk = W1_new[rows[:, None], cols[None, :]] - W1_old[rows[:, None], cols[None, :]]
        assert torch.allclose(blk, torch.full_like(blk, 1.0 + s), atol=1e-6)
        del blk
    assert torch.equal(W1_new != W1_old, expect)
    assert int(expect.sum().item()) == S * (H // S) * (2 * H // S)
    del expect
    bL_old = base0[models[0].base.offsets[L][1]:models[0].base.offsets[L][1] + C]
    assert torch.allclose(models[0].base.b[L], bL_old + 1.0 + (S - 1) / 2.0, atol=1e-5)
    # (4) re-dispatch under a new partition: every site gets the block it indexes
    part2 = models[0].sample_partitions()
    for m in models:
        m.dispatch_model(part2)
    for s in (0, 5):
        rows = part2[1][s][0].to(DEV)
        cols = part2[0][s][1].to(DEV)
        assert torch.equal(models[s].sub.W[1], models[s].base.W[1][rows[:, None], cols[None, :]])
        assert torch.equal(models[s].sub.W[0], models[s].base.W[0][part2[0][s][0].to(DEV)])
        assert torch.equal(models[s].sub.W[2], models[s].base.W[2][:, part2[1][s][1].to(DEV)])


def test_drop_in_evaluate_matches_engine_evaluator():
    """gist_amd.utils.evaluate(model, g, labels, mask) (cluster_gcn/utils.py:70-80 signature)
    on the nn.Module path == FullGraphEvaluator on the engine path, same parameters."""
    import torch.nn.functional as F
    from gist_amd import datasets
    from gist_amd.modules import GCN
    from gist_amd.trainer import FullGraphEvaluator
    from gist_amd.utils import evaluate
    from gist_amd.engine import ParamArena, dims_for
    ds = datasets.toy(seed=3)
    g = ds.g
    torch.manual_seed(1)
    model = GCN(g.ndata['feat'].shape[1], 48, ds.num_classes, 2, F.relu, 0.3, True, False, False,
                1, True).to(DEV)
    gd = g.to(DEV)
    dims = dims_for(g.ndata['feat'].shape[1], 48, ds.num_classes, 2)
    arena = ParamArena(dims, DEV, with_grads=False)
    arena.adopt_module(model)
    ev = FullGraphEvaluator(g, dims, True, arena, DEV)
    for mask in ('val_mask', 'test_mask', 'train_mask'):
        a = evaluate(model, gd, gd.ndata['label'], gd.ndata[mask], 'f1')
        b = ev.accuracy(mask)
        assert abs(a - b) < 1e-9, (mask, a, b)
    assert not model.training                           # evaluate() put the model in eval mode


@pytest.mark.parametrize('blocks', [None, False])
def test_evaluator_keeps_input_aggregation(blocks):
    """Layer 0 of the full-graph evaluation aggregates the input features -- the same product at every evaluation of a
    run.  The evaluator keeps it after its first forward: logits of later forwards (parameters changed in between) are
    bit for bit those of an evaluator that aggregates every time, and invalidate_input_aggregation() recomputes."""
    from gist_amd import datasets
    from gist_amd.trainer import FullGraphEvaluator
    from gist_amd.engine import ParamArena, dims_for
    ds = datasets.make_block_dataset('ev', 3000, 30, 132, 5, intra_deg=10, inter_deg=3, seed=5)
    g = ds.g.to(DEV)                  # (one device copy: both evaluators see the same feature tensor)
    dims = dims_for(132, 160, 5, 2)
    arena = ParamArena(dims, DEV, with_grads=False)
    gen = torch.Generator().manual_seed(2)
    kept = FullGraphEvaluator(g, dims, True, arena, DEV, row_block=1100, node_blocks=blocks)
    plain = FullGraphEvaluator(g, dims, True, arena, DEV, row_block=1100, node_blocks=blocks, cache_input_aggregation=False)
    assert kept.cache_input_aggregation and not plain.cache_input_aggregation
    assert (kept.split is not None) == (blocks is None) and kept.feat is plain.feat
    for rnd in range(3):
        for k, (i, o) in enumerate(dims):
            arena.W[k].copy_((torch.rand(o, 2 * i, generator=gen) - 0.5) * (2.0 / np.sqrt(2 * i)))
            arena.b[k].copy_((torch.rand(o, generator=gen) - 0.5) * 0.1)
        a = kept.forward().clone()
        assert kept._ah0_ready and plain._ah0 is None
        assert torch.equal(a, plain.forward())
        assert abs(kept.accuracy('val_mask') - plain.accuracy('val_mask')) == 0.0
    # features changed in place: the kept product is stale until invalidated
    kept.feat.mul_(1.5)
    stale = kept.forward().clone()
    fresh = plain.forward().clone()
    assert not torch.equal(stale, fresh)
    kept.invalidate_input_aggregation()
    assert torch.equal(kept.forward(), fresh)
    kept.feat.div_(1.5)


def _steps_vs_oracle(ds, batch_parts, n_hidden, n_layers, n_steps, p_seed):
    """Run n_steps real training iterations (native step driver, dropout off) on the GPU and
    the same iterations with the CPU oracle on the same cluster batches; compare losses,
    batch structure and final parameters."""
    from gist_amd.engine import SageEngine, dims_for
    from gist_amd.sampler import EngineClusterIter
    from oracle import gist_oracle as O
    from oracle import train_oracle as TO
    g = ds.g
    nid = np.arange(g.number_of_nodes(), dtype=np.int64)
    random.seed(p_seed)
    it = EngineClusterIter(ds.name, g, len(ds.par_li), batch_parts, nid,
                           par_li=[p.copy() for p in ds.par_li], device=DEV)
    F_, C_ = g.ndata['feat'].shape[1], ds.num_classes
    dims = dims_for(F_, n_hidden, C_, n_layers)
    eng = SageEngine(dims, True, 0.0, it.n_max, DEV)
    rs = np.random.RandomState(p_seed)
    params = []
    for (i, o) in dims:
        s = 1.0 / np.sqrt(2 * i)
        params.append((rs.uniform(-s, s, (o, 2 * i)).astype(np.float32),
                       rs.uniform(-s, s, o).astype(np.float32)))
    eng.arena.load(params)
    it.bind(eng)
    rp = g.rowptr.numpy().astype(np.int64)
    cl = g.col.numpy().astype(np.int64)
    tg = TO.TrainGraph(rp, cl, g.ndata['feat'].numpy(), g.ndata['label'].numpy().astype(np.int64))
    opt = O.new_opt_state(params)
    worst = 0.0
    for j, batch in enumerate(it):
        loss = eng.train_step(batch, 0.01, 5e-4)
        b = tg.batch(it.batch_ids(j))
        ref, _, _ = O.train_step(b[0], b[1], b[2], b[3], b[4], b[5], params, opt, True, 0.01,
                                 weight_decay=5e-4)
        assert np.array_equal(batch.rowptr.cpu().numpy(), b[0])
        assert np.array_equal(batch.col[:int(b[0][-1])].cpu().numpy(), b[1])
        worst = max(worst, abs(float(loss.item()) - float(ref)))
        if j == n_steps - 1:
            break
    errs = np.concatenate([np.abs(W - Wr).ravel() for (W, _), (Wr, _) in
                           zip(eng.arena.export(), params)])
    return worst, dict(mean=float(errs.mean()), frac_over_tol=float((errs > TOL).mean()),
                       max=float(errs.max()))


# Parameter-level comparisons of two GPU variants of the same step: bars = 2x the values measured on MI355X
# (mean |diff|, fraction of weights above 1e-4; GIST_PARITY_STATS=<file> appends the statistics of a run).
# The FREE-RUNNING post-Adam comparison with the oracle that lived here (bars 5e-5 / 9e-2 after two steps) is gone:
# it was chaotic by construction (Adam's first update is lr * sign(g); a ReLU input within rounding of zero flips in
# any two fp32 implementations) and a bar at 2x the observation could not catch a regression.  Its place is taken by
# tests/test_timed_step_parity_gpu.py: five TEACHER-FORCED steps of the timed configuration (dropout 0.2, fused
# step, default GEMM mode) with the pre-Adam gradients at 2e-5 relative / 1e-4 max-norm and the post-step
# parameters at 1e-4 max-norm; the statistics of the free-running runs are still recorded for the curious.
PARITY_BARS = {
    # two GPU variants of the same step (same masks unless a projection's rounding differs): 2x measured
    ('kept_vs_per_call_splits', 'f16x3', 4096): (1.8e-7, 4.6e-5),     # (round 5, DPP row sums in the LayerNorms: 8.9e-8 / 2.3e-5 measured)
    ('kept_vs_per_call_splits', 'bf16x3', 4096): (6e-9, 6e-6),      # (dW_0: 3 k slices summed in Adam vs in the call's own order)
    ('wide_class_layer_split_vs_f32', 'f16x3', 2048): (6.5e-7, 4.3e-4),
    ('wide_class_layer_split_vs_f32', 'bf16x3', 2048): (9.4e-7, 4.3e-4),
}


def _parity_ok(test, mode, hidden, stats):
    _record_parity(test, mode, hidden, stats)
    mean_bar, frac_bar = PARITY_BARS[(test, mode, hidden)]
    assert stats[0] < mean_bar and stats[1] < frac_bar, (test, mode, hidden, stats, (mean_bar, frac_bar))


def _record_parity(test, mode, hidden, stats):
    path = os.environ.get('GIST_PARITY_STATS')
    if path:
        with open(path, 'a') as f:
            f.write('%s mode=%s hidden=%d mean=%.3e frac_over_1e-4=%.3e max=%.3e\n' % ((test, mode, hidden) + stats))


def _check(loss_err, pstats):
    """Outputs (losses) are held to 1e-4.  Parameters AFTER Adam steps are checked in
    distribution: Adam divides the first moment by sqrt(second moment), so where a gradient is
    at rounding level (dead ReLU units, |g| ~ eps) two correct fp32 implementations with
    different summation orders can differ by a sizeable fraction of lr on those few weights;
    a max-norm is therefore not a parity metric at this scale (the small golden cases G2 and
    G6 do hold 1e-4 in max-norm)."""
    assert loss_err < TOL, loss_err
    assert pstats['mean'] < 2e-6, pstats
    assert pstats['frac_over_tol'] < 2e-3, pstats
    assert pstats['max'] < 3 * 0.01, pstats            # never more than lr per step


def test_config2_reddit_like_hidden256_L4():
    """BASELINE config 2: Reddit-like ClusterGCN GraphSAGE hidden=256, n_layers=4, batch of
    20 of 1500 parts (~2046 rows, ~1.3e5 edges): 3 full steps, GPU vs oracle."""
    from gist_amd import datasets
    ds = datasets.reddit_synth(seed=0)
    _check(*_steps_vs_oracle(ds, 20, 256, 4, 3, p_seed=1))


def test_config4_amazon_like_F100_C47_L4():
    """BASELINE config 4's per-rank shape family: Amazon-like graph (F=100, C=47, small parts,
    batch of 10 parts, mean in-batch degree ~16), sub-GCN width 512, n_layers=4 -- exercises
    the narrow-D lane-group SpMM (D=100) and a 5-layer model."""
    from gist_amd import datasets
    ds = datasets.amazon_synth(seed=1, n=171000, n_blocks=1500)
    _check(*_steps_vs_oracle(ds, 10, 512, 4, 3, p_seed=2))


def _metric_config_engine(mode, hidden=4096):
    """Engine + first batches of the configuration the metric is quoted on (Reddit-like batch,
    n_hidden=4096, L=2, full width on one GPU; hidden = 4096/N: the per-rank sub-GCN of the
    N-GPU points), projections in the given GEMM mode."""
    from gist_amd import datasets, hip, _lib
    from gist_amd.engine import SageEngine, dims_for
    from gist_amd.sampler import EngineClusterIter
    hip.gemm_mode(mode)
    ds = datasets.reddit_synth(seed=0)
    g = ds.g
    random.seed(3)
    it = EngineClusterIter(ds.name, g, len(ds.par_li), 20,
                           np.arange(g.number_of_nodes(), dtype=np.int64),
                           par_li=[p.copy() for p in ds.par_li], device=DEV)
    dims = dims_for(602, hidden, 41, 2)
    eng = SageEngine(dims, True, 0.0, it.n_max, DEV)
    if hidden == 4096:
        big = _lib.load().gist_gemm_workspace_bytes(it.n_max, 4096, 8192)
        assert mode == 'f32' or big > (1 << 27)            # the step really runs on the split path
    rs = np.random.RandomState(3)
    params = []
    for (i, o) in dims:
        sc = 1.0 / np.sqrt(2 * i)
        params.append((rs.uniform(-sc, sc, (o, 2 * i)).astype(np.float32),
                       rs.uniform(-sc, sc, o).astype(np.float32)))
    eng.arena.load(params)
    return ds, it, eng, dims, params


@pytest.mark.parametrize('mode,hidden', [('f32', 4096), ('bf16x3', 4096), ('bf16x3', 2048),
                                         ('bf16x3', 1024), ('bf16x3', 512),     # (both: every GEMM on the fp32 MFMA)
                                         ('f16x3', 4096), ('f16x3', 2048), ('f16x3', 1024)])
def test_metric_config_hidden4096_vs_oracle(mode, hidden):
    """Metric configuration (and the per-rank widths of the 2- and 4-GPU points), 2 full training
    steps against the oracle on the same batches, with the projections on the fp32 matrix-core
    path and on the f16x3 split path (every GEMM of the 4096-wide step but the 41-wide class
    layer takes it): the same 1e-4 bar on the outputs for both
    (gradients at this width are compared against float64 in the next test: a ReLU input at
    rounding level flips its mask in any fp32 implementation, the oracle's included)."""
    from gist_amd import hip
    from oracle import gist_oracle as O
    from oracle import train_oracle as TO
    prev = hip.gemm_mode()
    try:
        ds, it, eng, dims, params = _metric_config_engine(mode, hidden)
        it.bind(eng)
        if mode != 'f32' and hidden >= 1024:         # layer 1 (and 0) keep their split operands
            assert eng.plan.h3_workspace is not None    # (width 512: no projection reaches a split path)
        g = ds.g
        tg = TO.TrainGraph(g.rowptr.numpy().astype(np.int64), g.col.numpy().astype(np.int64),
                           g.ndata['feat'].numpy(), g.ndata['label'].numpy().astype(np.int64))
        opt = O.new_opt_state(params)
        for j, batch in enumerate(it):
            loss = eng.train_step(batch, 0.01, 0.0)
            logits = eng.logits(batch.n).cpu().numpy()
            b = tg.batch(it.batch_ids(j))
            ref_loss, ref_logits, _ = O.train_step(b[0], b[1], b[2], b[3], b[4], b[5], params, opt,
                                                   True, 0.01)
            # (1e-4 of the loss: at width 1024 the random-init loss is ~13, and the second step's loss sits behind
            # one Adam update, which turns rounding-level differences of near-zero gradients into +-lr)
            assert abs(float(loss.item()) - float(ref_loss)) < TOL * max(1.0, abs(float(ref_loss))), \
                (j, float(loss.item()), ref_loss)
            if j == 0:
                assert abs(float(loss.item()) - float(ref_loss)) < 2e-5 * max(1.0, abs(float(ref_loss)))
                assert np.abs(logits - ref_logits).max() <= TOL * max(1.0, np.abs(ref_logits).max())
            if j == 1:
                break
        errs = np.concatenate([np.abs(W - Wr).ravel() for (W, _), (Wr, _) in
                               zip(eng.arena.export(), params)])
        stats = (float(errs.mean()), float((errs > TOL).mean()), float(errs.max()))
        _record_parity('metric_config_vs_oracle', mode, hidden, stats)
        assert stats[2] <= 4 * 0.01 + 1e-6   # two Adam steps: at most 2 lr apart per step
    finally:
        hip.gemm_mode(prev)


def _float64_step(rowptr, col, feat, labels, params, masks):
    """float64 autograd evaluation of one forward/backward of the model on
    the device, dense normalised adjacency; masks[k] (bool [n, H]) replaces relu's own mask so
    that the comparison does not depend on the sign of rounding-level activations."""
    import torch.nn.functional as F
    n = len(rowptr) - 1
    A = torch.zeros(n, n, dtype=torch.float64, device=DEV)
    rows = torch.repeat_interleave(torch.arange(n, device=DEV),
                                   torch.from_numpy(np.diff(rowptr)).to(DEV))
    A.index_put_((rows, torch.from_numpy(col).to(DEV)),
                 torch.ones(len(col), dtype=torch.float64, device=DEV), accumulate=True)
    deg = A.sum(1, keepdim=True)
    A = torch.where(deg > 0, A / deg.clamp(min=1), torch.zeros_like(A))
    ps = [(torch.from_numpy(W).to(DEV).double().requires_grad_(),
           torch.from_numpy(b).to(DEV).double().requires_grad_()) for W, b in params]
    h = torch.from_numpy(feat).to(DEV).double()
    yhats = []
    for k, (W, b) in enumerate(ps):
        y = torch.cat([h, A @ h], 1) @ W.t() + b
        if k + 1 < len(ps):
            y = F.layer_norm(y, (y.shape[1],), eps=1e-5)
            yhats.append(y.detach())
            h = y * masks[k]
    loss = F.cross_entropy(y, torch.from_numpy(labels).to(DEV).long())
    loss.backward()
    return loss.item(), y.detach(), [(W.grad, b.grad) for W, b in ps], yhats


@pytest.mark.parametrize('mode,hidden', [('f32', 4096), ('bf16x3', 4096), ('f16x3', 4096),
                                         ('bf16x3', 1024), ('bf16x3', 512), ('f32', 1024)])
def test_metric_config_hidden4096_vs_float64(mode, hidden):
    """Metric configuration (and the per-rank widths 1024 / 512 of the N = 4 / 8 points, whose NT / NN
    projections run on the convert-on-load bf16x3 kernel in the default mode), one forward/backward against
    float64 autograd on the same batch.  Every GEMM mode must reproduce the float64 activations, logits, loss
    and every gradient to fp32 rounding level; the handful of ReLU inputs within rounding of zero (their
    count is bounded here) take the GPU's own mask in the reference -- THE parity statement for the
    gradients: the post-Adam comparison with the oracle (test_metric_config_hidden4096_vs_oracle) is chaotic
    in exactly those few masks (profiles/r03_parity_stats.txt)."""
    from gist_amd import hip
    from oracle import train_oracle as TO
    prev = hip.gemm_mode()
    try:
        ds, it, eng, dims, params = _metric_config_engine(mode, hidden)
        it.bind(eng, native=False)
        g = ds.g
        tg = TO.TrainGraph(g.rowptr.numpy().astype(np.int64), g.col.numpy().astype(np.int64),
                           g.ndata['feat'].numpy(), g.ndata['label'].numpy().astype(np.int64))
        batch = next(iter(it))
        n = batch.n
        b = tg.batch(it.batch_ids(0))
        eng.forward(batch, True)
        yh = [eng.Y[k][:n].clone() for k in range(2)]
        masks = [(t > 0).double() for t in yh]
        l64, y64, g64, yh64 = _float64_step(b[0], b[1], b[4], b[5], params, masks)
        for k in range(2):
            d = (yh[k].double() - yh64[k])
            assert d.pow(2).mean().sqrt().item() < 5e-6, (k, d.pow(2).mean().sqrt().item())
            assert d.abs().max().item() < 1e-4
            flips = int(((yh[k] > 0) != (yh64[k] > 0)).sum().item())
            assert flips <= 40, (k, flips)              # of 8.4 M activations
        loss = eng.loss_and_backward(batch)
        assert abs(loss.item() - l64) < 1e-5 * max(1.0, abs(l64))
        assert (eng.logits(n).double() - y64).abs().max().item() < TOL * max(1.0, y64.abs().max().item())
        for k in range(len(dims)):
            for got, ref in ((eng.arena.dW[k], g64[k][0]), (eng.arena.db[k], g64[k][1])):
                rel = (torch.linalg.norm(got.double() - ref) / torch.linalg.norm(ref)).item()
                assert rel < 2e-5, (k, tuple(got.shape), rel)
                assert (got.double() - ref).abs().max().item() <= TOL * ref.abs().max().item()
    finally:
        hip.gemm_mode(prev)


@pytest.mark.parametrize('mode', ['f16x3', 'bf16x3'])
def test_step_kept_split_operands_equal_per_call_splits(monkeypatch, mode):
    """Metric configuration, dropout 0.2, GEMM mode f16x3 / bf16x3: the native step that keeps its own
    split operands (dropout applied inside the activation split, one split of W per step,
    gradient maxima from the LayerNorm-backward / bias-gradient kernels) against the native
    step that splits inside every gist_gemm_* call (GIST_STEP_H3=0).  Same dropout stream, same
    arithmetic up to the operand scales: losses of 3 steps within 2e-5, parameters close."""
    from gist_amd import hip
    prev = hip.gemm_mode()
    try:
        runs = {}
        monkeypatch.setenv('GIST_STEP_PREAGG', '0')      # (the bars below were measured with layer 0's aggregation launch)
        for tag, env in (('per_call', '0'), ('kept', '1')):
            monkeypatch.setenv('GIST_STEP_H3', env)
            ds, it, eng, dims, params = _metric_config_engine(mode)
            eng.p_drop = 0.2
            it.bind(eng)
            assert (eng.plan.h3_workspace is not None) == (tag == 'kept')
            eng.plan.p_drop = 0.2
            losses = []
            for j, batch in enumerate(it):
                losses.append(float(eng.train_step(batch, 0.01, 0.0).item()))
                if j == 2:
                    break
            runs[tag] = (losses, eng.arena.params.clone())
        la, lb = runs['per_call'][0], runs['kept'][0]
        assert max(abs(a - b) / max(1.0, abs(a)) for a, b in zip(la, lb)) < 2e-5, (la, lb)      # (losses of ~40)
        assert la[0] > 3.0 and abs(la[0] - la[2]) > 1e-3          # it did train, with dropout
        d = (runs['per_call'][1] - runs['kept'][1]).abs()
        stats = (d.mean().item(), (d > TOL).float().mean().item(), d.max().item())
        _parity_ok('kept_vs_per_call_splits', mode, 4096, stats)
    finally:
        hip.gemm_mode(prev)


@pytest.mark.parametrize('hidden', [4096, 2048])
def test_metric_config_fused_step_equals_unfused_bf16x3(monkeypatch, hidden):
    """The metric's step (dropout 0.2, GEMM mode bf16x3, kept split operands) with the fused sequence against
    the un-fused one after ONE iteration: same masks, same projections; k slices are summed by their consumers
    (LayerNorm, loss kernel, optimiser) instead of reduce passes, bias gradients as chunk sums -- every gradient
    agrees to rounding."""
    from gist_amd import hip
    prev = hip.gemm_mode()
    try:
        runs = {}
        # (layer 0's aggregation stays its own launch in both runs: formed by the extraction -- the fused default, pinned in
        # tests/test_preagg_gpu.py -- it sums in another order; the losses stay bitwise equal, but among 8 M hidden units
        # three ReLU inputs within that rounding of zero flip their masks and every element of dW_0 moves by ~1e-3 of its
        # size: scripts/r4_preagg_flip_probe.py, profiles/r04_preagg.txt)
        monkeypatch.setenv('GIST_STEP_PREAGG', '0')
        for fuse in ('0', '1'):
            monkeypatch.setenv('GIST_STEP_FUSE', fuse)
            ds, it, eng, dims, params = _metric_config_engine('bf16x3', hidden)
            eng.p_drop = 0.2
            it.bind(eng)
            assert bool(eng.plan.fuse) == (fuse == '1') and eng.plan.h3_workspace is not None
            eng.plan.p_drop = 0.2
            batch = next(iter(it))
            loss = float(eng.train_step(batch, 0.01, 0.0).item())
            runs[fuse] = (loss, [w.clone() for w in eng.arena.dW], [b.clone() for b in eng.arena.db],
                          eng.arena.params.clone())
        assert abs(runs['0'][0] - runs['1'][0]) < 1e-6 * max(1.0, abs(runs['0'][0]))
        for k in range(len(dims)):
            gw = runs['0'][1][k]
            # (k slices left to a consumer may be cut differently from slices the call sums itself -- the class
            # layer's logits and weight gradient -- so everything downstream agrees to rounding, not bit for bit;
            # the slab sums themselves are held bitwise in tests/test_fused_ops_gpu.py)
            dgw = (gw - runs['1'][1][k]).abs()
            assert dgw.max().item() < 1e-4 * gw.abs().max().item() and dgw.mean().item() < 1e-6 * gw.abs().max().item(), k
            gb = runs['0'][2][k]
            assert (gb - runs['1'][2][k]).abs().max().item() < 1e-5 * max(1e-3, gb.abs().max().item()), k
        d = (runs['0'][3] - runs['1'][3]).abs()
        assert float((d > 1e-5).float().mean().item()) < 1e-4 and d.max().item() < 0.021      # (lr per step)
    finally:
        hip.gemm_mode(prev)


def test_wide_class_layer_stays_off_the_kept_split_path():
    """A class layer wide enough for the split projections (C=256, 8400 batch rows, hidden
    2048): its dY = dlogits has no row maxima from a LayerNorm backward, so the step must not
    split it with the kept-operand scales (it takes the per-call path, which computes its own).
    The f16x3 step must equal the f32 step: finite, losses within 2e-5, parameters close."""
    from gist_amd import datasets, hip
    from gist_amd.engine import SageEngine, dims_for
    from gist_amd.sampler import EngineClusterIter
    prev = hip.gemm_mode()
    try:
        runs = {}
        ds = datasets.make_block_dataset('wide-c', 8400, 4, 64, 256, intra_deg=6, inter_deg=2, seed=11)
        g = ds.g
        for mode in ('f32', 'f16x3', 'bf16x3'):
            hip.gemm_mode(mode)
            random.seed(5)
            it = EngineClusterIter(ds.name, g, len(ds.par_li), 4,
                                   np.arange(g.number_of_nodes(), dtype=np.int64),
                                   par_li=[p.copy() for p in ds.par_li], device=DEV)
            dims = dims_for(64, 2048, 256, 2)
            eng = SageEngine(dims, True, 0.0, it.n_max, DEV)
            rs = np.random.RandomState(7)
            eng.arena.load([(rs.uniform(-1, 1, (o, 2 * i)).astype(np.float32) / np.sqrt(2 * i),
                             rs.uniform(-1, 1, o).astype(np.float32) / np.sqrt(2 * i))
                            for (i, o) in dims])
            # stale "row maxima" of another tensor would sit in the split workspace: poison it
            it.bind(eng)
            if mode != 'f32':
                assert eng.plan.h3_workspace is not None      # layer 1 keeps its split operands
                eng._h3_ws.view(torch.float32).fill_(float('nan'))
            losses = []
            for j, batch in enumerate(it):
                assert batch.n >= 8192
                losses.append(float(eng.train_step(batch, 0.01, 0.0).item()))
                if j == 1:
                    break
            runs[mode] = (losses, eng.arena.params.clone())
        for split in ('f16x3', 'bf16x3'):
            la, lb = runs['f32'][0], runs[split][0]
            assert all(np.isfinite(lb)) and torch.isfinite(runs[split][1]).all()
            assert max(abs(a - b) for a, b in zip(la, lb)) < 2e-5, (split, la, lb)
            d = (runs['f32'][1] - runs[split][1]).abs()
            stats = (d.mean().item(), (d > TOL).float().mean().item(), d.max().item())
            _parity_ok('wide_class_layer_split_vs_f32', split, 2048, stats)
    finally:
        hip.gemm_mode(prev)


def test_cluster_iter_partitions_on_cache_miss(tmp_path, monkeypatch):
    """sampler.py:44-53: without a cache file ClusterIter partitions the train graph itself
    (here: gist_partition_graph instead of METIS), writes the reference's .npy format and
    reads the same list back on the next construction."""
    from gist_amd import datasets
    from gist_amd.sampler import ClusterIter, load_partition_cache
    ds = datasets.toy(seed=4, train_frac=0.8)
    train_nid = np.nonzero(ds.g.ndata['train_mask'].numpy())[0].astype(np.int64)
    work = tmp_path / 'run'
    work.mkdir()
    monkeypatch.chdir(work)                      # the cache path is relative: ../data/
    random.seed(3)
    it = ClusterIter('toy-miss', ds.g, 24, 4, train_nid, device=DEV)
    cache = tmp_path / 'data' / 'toy-miss_24.npy'
    assert cache.exists()
    on_disk = load_partition_cache(str(cache))
    assert len(on_disk) == 24
    ids = np.sort(np.concatenate(on_disk))
    assert np.array_equal(ids, np.arange(train_nid.shape[0]))        # ids of the TRAIN graph
    random.seed(3)
    it2 = ClusterIter('toy-miss', ds.g, 24, 4, train_nid, device=DEV)
    assert all(np.array_equal(a, b) for a, b in zip(it.par_li, it2.par_li))
    sub = next(iter(it2))
    assert sub.number_of_nodes() == sum(len(p) for p in it2.par_li[:4])
