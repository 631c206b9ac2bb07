"""GPU: oracle parity of THE STEP THAT IS TIMED -- the fused native step (gist_sage_step, plan.fuse = 1), dropout 0.2,
default GEMM mode (bf16x3), on the configurations bench.py reports -- and a chaos-free multi-step statement.

Reference: cluster_gcn/cluster_gcn_ist_distrib.py:405-417 (zero_grad, forward with nn.Dropout on [h | ah]
modules.py:227-231, CE, backward, Adam step).

Teacher forcing: before EVERY step the oracle's parameters and Adam state (exp_avg, exp_avg_sq, step count) are
copied into the engine's arena, so each of the 5 steps starts from identical state and a rounding-level difference
of one step cannot grow into a different trajectory (Adam's first updates are +-lr whatever the gradient's size).
The dropout masks are not stored anywhere on the GPU: the test rebuilds them on the host from the generator's
definition (seed, layer offset, element index -> splitmix64) and hands them to the oracle as `drop_masks`.

Per step, against O.train_step(..., drop_masks=..., drop_p=0.2):
  * loss at 1e-4 (relative to max(1, |ref|)); every ROW of the logits within 1e-4 of its own largest |logit| (or of 1),
    the absolute maximum printed;
  * the PRE-Adam gradients (Adam writes the summed split-K slabs / bias chunk sums back to the gradient arena):
    every tensor within 2e-5 of the oracle's in relative L2 norm and within 1e-4 x max|g| in max-norm;
  * the POST-step parameters in max-norm 1e-4; an element may exceed it only where Adam's update is an
    ill-conditioned function of the gradient (|g| < 1e-7, ten times Adam's eps: with exp_avg_sq ~ g^2 the update
    g / (|g| + 1e-8) turns a 1e-9 difference into a different step of up to lr -- in ANY two fp32 implementations);
    those elements are counted (< 1 %) and bounded by 2 lr;
  * ReLU: the GPU's and the oracle's masks agree on every activation except LayerNorm outputs within 4e-6 of zero
    (at most 8 per layer and step); on those the oracle's backward follows the GPU's decision -- one such flip
    moves a 512-wide layer's weight gradient by 1e-3 of its norm, which is a property of ReLU, not of either
    implementation -- AND, with nothing overwritten, the GPU's gradients are held to the oracle's backward under its
    OWN decisions, the allowance being exactly what the disagreeing activations contribute (the difference of the
    oracle's two backward passes);
  * the Adam kernel itself on ALL elements: float64 Adam applied on the host to the GPU's own gradient reproduces
    the GPU's parameters to 2e-7.
"""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)
TOL = 1e-4
P_DROP = 0.2
LR = 0.01


def _mask(n, d, p, seed, offset):
    """The step's dropout mask of one layer (gist_dropout_f32's generator: one splitmix64 per element pair)."""
    idx = np.arange(n * d, dtype=np.uint64) + np.uint64(offset)
    with np.errstate(over='ignore'):
        z = (idx >> np.uint64(1)) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    w = np.where(idx & np.uint64(1), z >> np.uint64(32), z & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    u = (w >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u >= np.float32(p)).reshape(n, d).astype(np.float32)


def _flat(pairs):
    return np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for (W, b) in pairs]).astype(np.float32)


def _teacher_force(eng, params, opt):
    A = eng.arena
    A.params.copy_(torch.from_numpy(_flat(params)))
    A.exp_avg.copy_(torch.from_numpy(_flat(opt['m'])))
    A.exp_avg_sq.copy_(torch.from_numpy(_flat(opt['v'])))
    A.step = int(opt['step'])


def _adam64(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    p, g, m, v = (a.astype(np.float64) for a in (p, g, m, v))
    m = m + (1 - beta1) * (g - m)
    v = beta2 * v + (1 - beta2) * g * g
    bc1, bc2 = 1 - beta1 ** t, 1 - beta2 ** t
    return p - (lr / bc1) * m / (np.sqrt(v) / np.sqrt(bc2) + eps)


def _run(ds, batch_parts, hidden, n_layers, n_steps, seed, mode='bf16x3', prefetch=False, module=False, first_parts=None,
         expect=None):
    """module=True: the step is issued by the reference's loop body on the drop-in classes (GCN.forward,
    nn.CrossEntropyLoss, loss.backward(), optim.Adam.step(): gist_amd/module_engine.py) instead of
    SageEngine.train_step -- same plan, three phase calls."""
    from gist_amd import hip
    from gist_amd.engine import SageEngine, dims_for
    from gist_amd.sampler import ClusterIter, EngineClusterIter
    from oracle import gist_oracle as O
    from oracle import train_oracle as TO
    prev = hip.gemm_mode()
    hip.gemm_mode(mode)
    try:
        g = ds.g
        random.seed(seed)
        it = (ClusterIter if module else EngineClusterIter)(
            ds.name, g, len(ds.par_li), batch_parts, np.arange(g.number_of_nodes(), dtype=np.int64),
            par_li=[p.copy() for p in ds.par_li], device=DEV)
        if first_parts is not None:      # these parts of ds.par_li lead the first epoch's (shuffled) order
            lead = {int(ds.par_li[i][0]) for i in first_parts}
            it.par_li = ([p for i in first_parts for p in it.par_li if int(p[0]) == int(ds.par_li[i][0])] +
                         [p for p in it.par_li if int(p[0]) not in lead])
        F_, C_ = g.ndata['feat'].shape[1], ds.num_classes
        dims = dims_for(F_, hidden, C_, n_layers)
        me = model = optimizer = loss_f = None
        if module:
            import torch.nn.functional as F
            from gist_amd import module_engine
            from gist_amd.modules import GCN
            from gist_amd.nn import CrossEntropyLoss
            from gist_amd.optim import Adam
            model = GCN(F_, hidden, C_, n_layers, F.relu, P_DROP, True, False, False, 1, True).cuda()
            model.set_dropout_seed(seed + 11)
            loss_f, optimizer = CrossEntropyLoss(), Adam(model.parameters(), lr=LR, weight_decay=0.0)
            assert it.feed()
            me = module_engine.ModuleEngine(model, it)
            model.__dict__['_module_engines'] = {id(it): me}
            me.engine.prefetch = prefetch
            eng = me.engine
        else:
            eng = SageEngine(dims, True, P_DROP, it.n_max, DEV, seed=seed + 11)
        assert eng.fuse                                   # the fused sequence: what bench.py times
        rs = np.random.RandomState(seed)
        params = []
        for (i, o) in dims:
            s = 1.0 / np.sqrt(2 * i)
            params.append((rs.uniform(-s, s, (o, 2 * i)).astype(np.float32),
                           rs.uniform(-s, s, o).astype(np.float32)))
        if not module:
            it.bind(eng)
            eng.prefetch = prefetch                       # (what bench.py and the trainers run at the narrow widths)
        assert eng.plan is not None                       # native step driver (gist_sage_step)
        tg = TO.TrainGraph(g.rowptr.numpy().astype(np.int64), g.col.numpy().astype(np.int64),
                           g.ndata['feat'].numpy(), g.ndata['label'].numpy().astype(np.int64))
        opt = O.new_opt_state(params)
        report, flips, logit_stats = [], [], []
        for j, batch in enumerate(it):
            _teacher_force(eng, params, opt)
            before = [(W.copy(), b.copy()) for (W, b) in params]
            m_before = [(a.copy(), b.copy()) for (a, b) in opt['m']]
            v_before = [(a.copy(), b.copy()) for (a, b) in opt['v']]
            off = eng.drop_calls
            if module:
                m_, v_ = me.flat_state(optimizer)
                m_.copy_(eng.arena.exp_avg)
                v_.copy_(eng.arena.exp_avg_sq)
                optimizer.step_count = int(opt['step'])
                cluster = batch.to(torch.cuda.current_device())      # cluster_gcn/cluster_gcn.py:96-105, verbatim
                model.train()
                pred = model(cluster)
                batch_labels = cluster.ndata['label']
                batch_train_mask = cluster.ndata['train_mask']
                loss = loss_f(pred[batch_train_mask], batch_labels[batch_train_mask])
                optimizer.zero_grad()
                loss.backward()
                optimizer.step()
                assert type(loss).__name__ == 'StepLoss' and me.state == 0      # (the fused phases ran, all three)
                n = cluster.number_of_nodes()
                logits = pred.detach().cpu().numpy()
                rowptr_now = it.batcher.rowptr[:n + 1]
            else:
                loss = eng.train_step(batch, LR, 0.0)
                n = batch.n
                logits = eng.logits(n).cpu().numpy()
                rowptr_now = batch.rowptr
            if expect is not None:
                expect(j, n, bool(getattr(batch, 'siblings', False)))
            # the masks of this step, rebuilt on the host
            masks, o_ = [], off
            for (i, o) in dims:
                masks.append(_mask(n, 2 * i, P_DROP, eng.seed, o_))
                numel = n * 2 * i
                o_ += numel + (numel & 1)
            assert o_ == eng.drop_calls
            b = tg.batch(it.batch_ids(j))
            if prefetch and j + 1 < len(it):
                # the batch buffers already hold the NEXT batch (extracted beside this step's optimiser launch):
                # its induced CSR against the oracle's
                assert it.batcher.prefetched is not None
                nb = tg.batch(it.batch_ids(j + 1))
                assert np.array_equal(it.batcher.rowptr[:len(nb[0])].cpu().numpy(), nb[0])
                assert np.array_equal(it.batcher.col[:len(nb[1])].cpu().numpy(), nb[1])
            else:
                assert np.array_equal(rowptr_now.cpu().numpy(), b[0])
            # the oracle's step, composed from its own parts (O.train_step's body) so that the ReLU decision of the
            # handful of LayerNorm outputs WITHIN ROUNDING OF ZERO can follow the GPU's: everywhere else the two
            # ReLU masks must agree, and the ambiguous elements are counted
            ref_logits, caches = O.gcn_forward(b[0], b[1], b[4], params, True, drop_masks=masks, drop_p=P_DROP)
            own_out = [c['out'].copy() if 'out' in c else None for c in caches]      # the oracle's OWN ReLU decisions
            step_flips = 0
            for k in range(len(dims) - 1):
                i_next = dims[k + 1][0]
                pos_gpu = (eng.Z[k + 1][:n, :i_next] > 0)
                if eng.H[k + 1] is not None:           # the undropped copy the fused producers leave
                    pos_gpu = pos_gpu | (eng.H[k + 1][:n, :i_next] > 0)
                pos_gpu = pos_gpu.cpu().numpy()
                yhat = caches[k]['yhat']
                pos_ref = caches[k]['out'] > 0
                differ = pos_gpu != pos_ref
                # a dropped element reads 0 in Z's left half whatever its sign: only trust "GPU says positive" there
                if eng.H[k + 1] is None:
                    differ &= pos_gpu
                assert np.abs(yhat[differ]).max(initial=0.0) < 4e-6, (j, k, float(np.abs(yhat[differ]).max()))
                assert int(differ.sum()) <= 8, (j, k, int(differ.sum()))
                flips.append(int(differ.sum()))
                step_flips += int(differ.sum())
                out = caches[k]['out']
                out[differ & pos_gpu] = np.float32(1e-30)          # backward reads only (out > 0)
                out[differ & ~pos_gpu] = 0.0
            ref_loss, dlog = O.cross_entropy(ref_logits, b[5])
            ref_grads = O.gcn_backward(caches, dlog, b[2], b[3])
            if step_flips:
                # Second statement, with NOTHING overwritten: the oracle's backward under its OWN ReLU decisions.  The
                # GPU's gradients may differ from it by what the disagreeing activations contribute -- measured on the
                # oracle itself as the difference between its two backward passes -- plus the usual rounding bar.
                for c, o in zip(caches, own_out):
                    if o is not None:
                        c['out'] = o
                own_grads = O.gcn_backward(caches, dlog, b[2], b[3])
                for k in range(len(dims)):
                    for gg, g_own, g_forced in ((eng.arena.dW[k].cpu().numpy(), own_grads[k][0], ref_grads[k][0]),
                                                (eng.arena.db[k].cpu().numpy(), own_grads[k][1], ref_grads[k][1])):
                        nrm = lambda a: float(np.linalg.norm(a.astype(np.float64)))
                        assert nrm(gg - g_own) <= 2e-5 * nrm(g_own) + 1.01 * nrm(g_forced - g_own), (j, k, step_flips)
            opt['step'] += 1
            for k, ((W, bb), (dW, db)) in enumerate(zip(params, ref_grads)):
                O.adam_step(W, dW, opt['m'][k][0], opt['v'][k][0], opt['step'], LR)
                O.adam_step(bb, db, opt['m'][k][1], opt['v'][k][1], opt['step'], LR)
            t = opt['step']
            assert abs(float(loss.item()) - float(ref_loss)) < TOL * max(1.0, abs(float(ref_loss))), (j, float(loss.item()), ref_loss)
            # every row of the logits against ITS OWN size (not the batch's largest logit: in the H = 4096 transient
            # they reach ~100), and the absolute maximum on record
            row_ref = np.maximum(1.0, np.abs(ref_logits).max(axis=1))
            row_err = np.abs(logits - ref_logits).max(axis=1)
            assert (row_err <= TOL * row_ref).all(), (j, float((row_err / row_ref).max()))
            logit_stats.append((float(row_err.max()), float(np.abs(ref_logits).max()), float((row_err / row_ref).max())))
            got_p = eng.arena.export()
            worst_rel, ill, n_all = 0.0, 0, 0
            for k in range(len(dims)):
                for name, gg, gr, pg, pr, pb, mb, vb in (
                        ('W', eng.arena.dW[k].cpu().numpy(), ref_grads[k][0], got_p[k][0], params[k][0],
                         before[k][0], m_before[k][0], v_before[k][0]),
                        ('b', eng.arena.db[k].cpu().numpy(), ref_grads[k][1], got_p[k][1], params[k][1],
                         before[k][1], m_before[k][1], v_before[k][1])):
                    rel = float(np.linalg.norm((gg - gr).astype(np.float64)) / max(np.linalg.norm(gr.astype(np.float64)), 1e-30))
                    assert rel < 2e-5, (j, k, name, rel)
                    assert np.abs(gg - gr).max() <= TOL * np.abs(gr).max(), (j, k, name)
                    worst_rel = max(worst_rel, rel)
                    # post-step parameters in max-norm 1e-4.  The only elements allowed above it are those where Adam's
                    # update is an ill-conditioned function of the gradient -- |g| within 10x of Adam's eps = 1e-8, where
                    # g / (|g| + eps) turns a 1e-9 difference into a different step in ANY two fp32 implementations --
                    # they are counted (< 1 % of the tensor) and can differ by at most 2 lr
                    d = np.abs(pg - pr)
                    above = d >= TOL
                    assert np.abs(gr[above]).max(initial=0.0) < 1e-7, (j, k, name, float(np.abs(gr[above]).max(initial=0.0)))
                    assert d.max(initial=0.0) <= 2 * LR + 1e-6, (j, k, name)
                    ill += int(above.sum())
                    n_all += d.size
                    # the Adam kernel on ALL elements, from the GPU's own gradient
                    p64 = _adam64(pb, gg, mb, vb, t, LR)
                    assert np.abs(pg.astype(np.float64) - p64).max() < 2e-7, (j, k, name)
            assert ill < 0.01 * n_all, (ill, n_all)
            report.append((j, float(ref_loss), worst_rel, ill / float(n_all)))
            if j == n_steps - 1:
                break
        print('timed-step parity h=%d L=%d: max |logit error| %.3g absolute (largest |logit| %.3g), %.3g of the row\'s own '
              'size; ReLU decisions that differ from the oracle\'s: %d in %d steps; worst gradient rel. L2 %.2g'
              % (hidden, n_layers, max(a for a, _, _ in logit_stats), max(b_ for _, b_, _ in logit_stats),
                 max(c for _, _, c in logit_stats), sum(flips), len(report), max(r[2] for r in report)))
        return report
    finally:
        hip.gemm_mode(prev)


@pytest.mark.parametrize('hidden', [4096, 2048, 1024, 512])
def test_config3_timed_step_dropout_teacher_forced(hidden):
    """BASELINE config 3: the default bench workload (Reddit-like, L=2, H=4096: kept bf16x3 splits, block-dense
    aggregation) and the per-rank widths of its 2-, 4- and 8-GPU points (2048: kept splits; 1024 -- config 3's own S = 4 --
    and 512: convert-on-load bf16x3 + fp32 kernels, LDS gather)."""
    from gist_amd import datasets
    rep = _run(datasets.reddit_synth(seed=0), 20, hidden, 2, 5, seed=3)
    assert len(rep) == 5


@pytest.mark.parametrize('prefetch', [False, True])
def test_config2_timed_step_dropout_teacher_forced(prefetch):
    """BASELINE config 2: Reddit-like, hidden 256, 4 hidden layers (--config 2); with the next batch extracted inside
    the optimiser's launch (what bench.py times at this width) and without."""
    from gist_amd import datasets
    rep = _run(datasets.reddit_synth(seed=0), 20, 256, 4, 5, seed=4, prefetch=prefetch)
    assert len(rep) == 5


@pytest.mark.parametrize('hidden,n_layers,prefetch', [(512, 2, True), (4096, 2, False), (256, 4, True)])
def test_module_path_timed_step_dropout_teacher_forced(hidden, n_layers, prefetch):
    """The SAME statements through the drop-in module path (north_star: "training scripts and GraphSAGE module API are
    drop-in unchanged"): the reference's loop body on gist_amd.modules.GCN / nn.CrossEntropyLoss / optim.Adam /
    sampler.ClusterIter, dropout 0.2, default GEMM mode, teacher-forced against the oracle -- config 3's 8-GPU per-rank
    width, the metric's H = 4096, config 2."""
    from gist_amd import datasets
    rep = _run(datasets.reddit_synth(seed=0), 20, hidden, n_layers, 3, seed=3, prefetch=prefetch, module=True)
    assert len(rep) == 3


def test_unplanted_partition_timed_step_dropout_teacher_forced():
    """A batch whose parts are NOT planted: a power-law community graph (sizes 30-400, mixing 0.3, random node ids) cut into
    parts of ~100 by gist_partition_graph -- communities larger than a part are split, small ones share a part, a third of
    a row's neighbours lie outside its part -- through the timed step (width 512, dropout 0.2, the next batch extracted
    inside the optimiser launch), teacher-forced against the oracle like the planted configurations."""
    from gist_amd import datasets
    from gist_amd.dgl_compat.transform import partition_assignment
    ds = datasets.community_dataset('communities', 30000, 602, 41, seed=3)
    k = 300
    assign = partition_assignment(ds.g, k, seed=0)
    order = np.argsort(assign, kind='stable')
    bounds = np.searchsorted(assign[order], np.arange(k + 1))
    ds = ds._replace(par_li=[order[bounds[i]:bounds[i + 1]].astype(np.int64) for i in range(k)])
    rep = _run(ds, 20, 512, 2, 3, seed=5, prefetch=True)
    assert len(rep) == 3


@pytest.mark.parametrize('hidden', [2048, 4096])
def test_unplanted_batches_over_2048_rows_with_sibling_parts_teacher_forced(hidden):
    """What the planted model never shows, at the widths that take the kept-split bf16x3 projections and the block-dense
    matrix-core aggregation: batches of 2049-2150 rows (a NINTH 256-row tile: the projections' tail units, gemm_b3.hip) that
    hold two parts of one community (their off-diagonal blocks as dense pairs, spmm_mfma.hip) -- the power-law community
    graph cut by gist_partition_graph into parts of ~104, the first batches led by sibling parts -- through the timed
    step, dropout 0.2, teacher-forced against the oracle."""
    from gist_amd import datasets
    from gist_amd.dgl_compat.transform import partition_assignment
    ds = datasets.community_dataset('communities', 30000, 602, 41, seed=3)
    k = 288
    assign = partition_assignment(ds.g, k, seed=0)
    order = np.argsort(assign, kind='stable')
    bounds = np.searchsorted(assign[order], np.arange(k + 1))
    ds = ds._replace(par_li=[order[bounds[i]:bounds[i + 1]].astype(np.int64) for i in range(k)])
    # sibling parts: pairs joined by at least gist_spmm_pair_min_edges() edges in one direction (the prepare kernel's threshold)
    rp, col = ds.g.rowptr.numpy().astype(np.int64), ds.g.col.numpy().astype(np.int64)
    rows = np.repeat(np.arange(ds.g.number_of_nodes()), np.diff(rp))
    pr, pc = assign[rows].astype(np.int64), assign[col].astype(np.int64)
    keys, cnt = np.unique(pr[pr != pc] * k + pc[pr != pc], return_counts=True)
    from gist_amd import _lib
    sib = [(int(q // k), int(q % k)) for q in keys[cnt >= int(_lib.load().gist_spmm_pair_min_edges())]]
    assert len(sib) >= 3, 'test data: the partition splits no community across two parts'
    lead, used = [], set()
    for (a, b) in sib:                       # three disjoint pairs, one for each of the first three batches
        if a not in used and b not in used and len(lead) < 3:
            lead.append((a, b))
            used.update((a, b))
    rest = [i for i in range(k) if i not in used]
    first = []
    for j, (a, b) in enumerate(lead):
        first += [a, b] + rest[18 * j:18 * (j + 1)]
    seen = []
    rep = _run(ds, 20, hidden, 2, 3, seed=5, first_parts=first, expect=lambda j, n, s_: seen.append((n, s_)))
    assert len(rep) == 3
    assert all(s_ for _, s_ in seen), seen                       # the host knew: every one of these batches has sibling parts
    assert any(2049 <= n <= 2150 for n, _ in seen), seen         # a ninth row tile


def test_config3_per_rank_width_with_prefetched_batches():
    """Config 3's 8-GPU per-rank width (512) as bench.py times it: every batch but the first extracted beside the
    previous step's optimiser launch."""
    from gist_amd import datasets
    rep = _run(datasets.reddit_synth(seed=0), 20, 512, 2, 5, seed=3, prefetch=True)
    assert len(rep) == 5


def test_config4_full_size_timed_step_dropout_teacher_forced():
    """BASELINE config 4 at its FULL size: the Amazon-like graph with 1 709 997 training nodes in 15 000 parts
    (F=100, C=47), batch of 10 parts, the per-rank sub-GCN of the 8-GPU run (width 512, 4 hidden layers).  Exercises
    the resident-graph scale (int32 offsets of a ~40 M-edge CSR, part tables of 15 000 parts, 1 500 batches per
    epoch) that the 171 k-node case of test_e2e_gpu.py does not; plus structural properties of the epoch."""
    from gist_amd import datasets
    from gist_amd.sampler import EngineClusterIter
    ds = datasets.amazon_synth(seed=1)
    g = ds.g
    assert g.number_of_nodes() == 1709997 and len(ds.par_li) == 15000
    rep = _run(ds, 10, 512, 4, 3, seed=5)
    assert len(rep) == 3
    # properties of one epoch at this size: every node appears in exactly one batch; the batches' induced edge
    # counts never exceed the extraction buffers
    random.seed(6)
    it = EngineClusterIter(ds.name, g, len(ds.par_li), 10, np.arange(g.number_of_nodes(), dtype=np.int64),
                           par_li=[p.copy() for p in ds.par_li], device=DEV)
    assert len(it) == 1500
    seen = np.zeros(g.number_of_nodes(), np.int32)
    for j in range(len(it)):
        np.add.at(seen, it.batch_ids(j), 1)
    assert seen.min() == 1 and seen.max() == 1
