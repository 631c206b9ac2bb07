"""Pins the CPU oracle (oracle/) against golden vectors recorded from the reference's
own code (tests/golden/*.npz, made by oracle/gen_golden.py).  CPU only.

Tolerance: 1e-4 absolute on fp32 values (BASELINE.json north_star); indices and
orders bit-exact.
"""
import glob
import os
import random

import numpy as np
import pytest

from oracle import gist_oracle as O
from oracle import train_oracle as T

TOL = 1e-4


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _params(d, prefix, n):
    return [(d['%sW%d' % (prefix, k)].copy(), d['%sb%d' % (prefix, k)].copy()) for k in range(n)]


def _g1_files(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, 'G1_layer_*.npz')))


def test_golden_present(golden_dir):
    assert len(_g1_files(golden_dir)) == 12
    assert len(glob.glob(os.path.join(golden_dir, 'G2_model_*.npz'))) == 15


def test_c_spmm_matches_python_loop():
    rs = np.random.RandomState(0)
    n, d = 50, 7
    src, dst = rs.randint(0, n, 300), rs.randint(0, n - 1, 300)
    rowptr, col = O.csr_from_edges(src, dst, n)
    x = rs.randn(n, d).astype(np.float32)
    assert np.array_equal(O.spmm_sum(rowptr, col, x), O.spmm_sum_py(rowptr, col, x))
    # strided input / accumulate / scales
    big = rs.randn(n, 2 * d).astype(np.float32)
    r = O.in_degree_norm(rowptr)
    y = np.ones((n, d), np.float32)
    O.spmm_sum(rowptr, col, big[:, d:], src_scale=r, out=y, accumulate=True)
    ref = 1 + O.spmm_sum_py(rowptr, col, np.ascontiguousarray(big[:, d:]) * r[:, None])
    assert np.allclose(y, ref, atol=1e-5)


@pytest.mark.parametrize('idx', range(12))
def test_G1_layer(golden_dir, idx):
    d = np.load(_g1_files(golden_dir)[idx])
    rowptr, col = d['rowptr'], d['col']
    # the stub's CSR and the oracle's CSR builder agree
    rp2, cl2 = O.csr_from_edges(d['src'], d['dst'], int(d['n']))
    assert np.array_equal(rp2, rowptr) and np.array_equal(cl2, col)
    out, cache = O.sage_layer_forward(rowptr, col, d['h'], d['W'], d['b'],
                                      bool(d['use_lynorm']), bool(d['relu']))
    assert np.abs(out - d['out']).max() < TOL
    t_rp, t_cl = O.transpose_csr(rowptr, col)
    dh, dW, db = O.sage_layer_backward(cache, d['d_out'], t_rp, t_cl)
    assert np.abs(dh - d['dh']).max() < TOL
    assert np.abs(dW - d['dW']).max() < TOL * max(1.0, np.abs(d['dW']).max())
    assert np.abs(db - d['db']).max() < TOL * max(1.0, np.abs(d['db']).max())
    # zero in-degree node exists and its aggregate is exactly zero
    deg = np.diff(rowptr)
    assert (deg == 0).any()
    n_in = d['h'].shape[1]
    assert np.all(cache['z'][deg == 0, n_in:] == 0)


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(
    os.path.dirname(__file__), 'golden', 'G2_model_*.npz'))), ids=os.path.basename)
def test_G2_model(path):
    d = np.load(path)
    L, S, H = int(d['L']), int(d['S']), int(d['H'])
    kind = str(d['kind'])
    nl = L + 1
    dims = O.gcn_layer_dims(d['feat'].shape[1], H, int(d['n_classes']), L,
                            split_output=(kind == 'sub'), num_subnet=S)
    params = _params(d, 'init_', nl)
    for (i, o, _, _), (W, b) in zip(dims, params):
        assert W.shape == (o, 2 * i) and b.shape == (o,)
    ln = bool(d['use_layernorm'])
    rowptr, col = d['rowptr'], d['col']
    t_rp, t_cl = O.transpose_csr(rowptr, col)
    logits, caches = O.gcn_forward(rowptr, col, d['feat'], params, ln)
    assert np.abs(logits - d['logits']).max() < TOL
    loss, dlog = O.cross_entropy(logits, d['labels'])
    assert abs(loss - d['loss']) < TOL
    grads = O.gcn_backward(caches, dlog, t_rp, t_cl)
    for k, (dW, db) in enumerate(grads):
        assert np.abs(dW - d['dW%d' % k]).max() < TOL
        assert np.abs(db - d['db%d' % k]).max() < TOL
    opt = O.new_opt_state(params)
    for step in range(3):
        loss, _, _ = O.train_step(rowptr, col, t_rp, t_cl, d['feat'], d['labels'], params,
                                  opt, ln, lr=0.01, weight_decay=float(d['wd']))
        assert abs(loss - d['loss_step%d' % step]) < TOL
        if step in (0, 2):
            for k, (W, b) in enumerate(params):
                assert np.abs(W - d['step%d_W%d' % (step + 1, k)]).max() < TOL
                assert np.abs(b - d['step%d_b%d' % (step + 1, k)]).max() < TOL


def _parts(d):
    return [d['part%d' % i] for i in range(int(d['psize']))]


def test_G3_sampler(golden_dir):
    d = _load(golden_dir, 'G3_sampler.npz')
    n = int(d['n'])
    f_rowptr, f_col = O.csr_from_edges(d['src'], d['dst'], n)
    tr_rowptr, tr_col = O.induced_subgraph(f_rowptr, f_col, d['train_nid'])
    assert np.array_equal(tr_rowptr, d['train_rowptr'])
    assert np.array_equal(tr_col, d['train_col'])
    random.seed(int(d['seed']))
    it = O.ClusterIterOracle(_parts(d), int(d['psize']), int(d['batch_size']))
    assert len(it) == int(d['n_batches'])
    for ep in range(2):
        for j, ids in enumerate(it):
            assert np.array_equal(ids, d['ep%d_b%d_nid' % (ep, j)])
            if ep == 0 and j == 0:
                rp, cl = O.induced_subgraph(tr_rowptr, tr_col, ids)
                assert np.array_equal(rp, d['b0_rowptr']) and np.array_equal(cl, d['b0_col'])
                assert np.array_equal(d['feat'][d['train_nid']][ids], d['b0_feat'])
                assert np.array_equal(d['label'][d['train_nid']][ids], d['b0_label'])


def test_G4_create_partition(golden_dir):
    d = _load(golden_dir, 'G4_create_partition.npz')
    for seed in (0, 3):
        for S, H in ((2, 16), (4, 16), (8, 64)):
            random.seed(seed)
            part = O.create_partition(S, H)
            for s, (idx, full) in enumerate(part):
                assert np.array_equal(idx, d['cp_seed%d_S%d_H%d_s%d' % (seed, S, H, s)])
                assert np.array_equal(full, np.concatenate([idx, idx + H]))


@pytest.mark.parametrize('name', ['G4_ist_S2_H16_L2.npz', 'G4_ist_S4_H16_L2.npz',
                                  'G4_ist_S2_H8_L1.npz', 'G4_ist_S4_H16_L3.npz'])
def test_G4_dispatch_sync(golden_dir, name):
    d = _load(golden_dir, name)
    S, H, L = int(d['S']), int(d['H']), int(d['L'])
    random.seed(int(d['seed']))
    base = _params(d, 'base0_', L + 1)
    part = O.sample_partitions(L, S, H)
    for l in range(L):
        for s in range(S):
            assert np.array_equal(part[l][s][0], d['part0_l%d_s%d' % (l, s)])
    subs = [O.dispatch_site(base, part, s) for s in range(S)]
    for s in range(S):
        for k, (W, b) in enumerate(subs[s]):
            assert np.array_equal(W, d['r%d_sub_ini_W%d' % (s, k)])
            assert np.array_equal(b, d['r%d_sub_ini_b%d' % (s, k)])
    subs = [_params(d, 'r%d_sub_pert_' % s, L + 1) for s in range(S)]
    O.sync_sites(base, subs, part)
    for k, (W, b) in enumerate(base):
        assert np.array_equal(W, d['base1_W%d' % k])
        if k < L:
            assert np.array_equal(b, d['base1_b%d' % k])
        else:   # all-reduce SUM order may differ from gloo's in the last ulp
            assert np.abs(b - d['base1_b%d' % k]).max() < 1e-6
    base = _params(d, 'base1_', L + 1)
    part = O.sample_partitions(L, S, H)
    for l in range(L):
        for s in range(S):
            assert np.array_equal(part[l][s][0], d['part1_l%d_s%d' % (l, s)])
    subs = [O.dispatch_site(base, part, s) for s in range(S)]
    for s in range(S):
        for k, (W, b) in enumerate(subs[s]):
            assert np.array_equal(W, d['r%d_sub_disp_W%d' % (s, k)])
            if k < L:
                assert np.array_equal(b, d['r%d_sub_disp_b%d' % (s, k)])
    # dispatch -> sync with no training is the identity on the base model
    O.sync_sites(base, subs, part)
    for k, (W, b) in enumerate(base):
        assert np.array_equal(W, d['base2_W%d' % k])
        assert np.abs(b - d['base2_b%d' % k]).max() < 1e-6


def test_G5_graphconv(golden_dir):
    d = _load(golden_dir, 'G5_graphconv.npz')
    h = d['x']
    for k in range(2):
        h = O.graphconv_forward(d['rowptr'], d['col'], d['out_deg'], h, d['W%d' % k],
                                d['b%d' % k], relu=(k == 0))
        if k == 0:
            h = O.whole_tensor_layer_norm(h)
    assert np.abs(h - d['out']).max() < TOL


def _full(d):
    n = int(d['n'])
    rowptr, col = O.csr_from_edges(d['src'], d['dst'], n)
    return (rowptr, col, d['feat'], d['label'])


def test_G6_e2e_single(golden_dir):
    d = _load(golden_dir, 'G6_e2e_single.npz')
    full = _full(d)
    masks = (d['train_mask'], d['val_mask'], d['test_mask'])
    params = _params(d, 'init_', int(d['n_layers']) + 1)
    random.seed(int(d['rnd_seed']))
    snaps = {}

    def on_epoch(e, p):
        snaps[e] = [(W.copy(), b.copy()) for W, b in p]

    def eval_fn(p, mask):
        return T.evaluate(full[0], full[1], full[2], full[3], mask, p, True)
    res = T.run_cluster_gcn(full, masks, _parts(d), int(d['psize']), int(d['batch_size']),
                            params, True, float(d['lr']), int(d['n_epochs']),
                            eval_fn=eval_fn, on_epoch=on_epoch)
    for e in range(int(d['n_epochs'])):
        for k, (W, b) in enumerate(snaps[e]):
            assert np.abs(W - d['ep%d_W%d' % (e, k)]).max() < TOL
            assert np.abs(b - d['ep%d_b%d' % (e, k)]).max() < TOL
    assert np.allclose(res['val_accs'], d['val_accs'], atol=1e-6)
    assert abs(res['val_accs'][-1] - float(d['last_val'])) < 1e-4
    assert abs(max(res['test_accs']) - float(d['best_test'])) < 1e-4


@pytest.mark.parametrize('S', [2, 4])
def test_G6_e2e_ist(golden_dir, S):
    d = _load(golden_dir, 'G6_e2e_ist_S%d.npz' % S)
    full = _full(d)
    masks = (d['train_mask'], d['val_mask'], d['test_mask'])
    L = int(d['n_layers'])
    base = _params(d, 'r0_base_init_', L + 1)
    random.seed(int(d['rnd_seed']))
    syncs = []

    def eval_fn(p, mask):
        return T.evaluate(full[0], full[1], full[2], full[3], mask, p, True)
    res = T.run_gist(full, masks, _parts(d), int(d['psize']), int(d['batch_size']), base, S,
                     int(d['n_hidden']), L, True, float(d['lr']), int(d['n_epochs']),
                     int(d['iter_per_site']), eval_fn=eval_fn,
                     on_sync=lambda b: syncs.append([(W.copy(), x.copy()) for W, x in b]))
    # event schedule (golden logs 'eval' twice per evaluation: val and test)
    gold = [str(e) for e in d['r0_events']]
    dedup = [e for i, e in enumerate(gold) if not (e == 'eval' and gold[i - 1] == 'eval')]
    assert res['events'] == dedup
    assert res['events'] == [a for _, a in O.ist_schedule(
        int(d['n_epochs']), S, int(d['psize']) // int(d['batch_size']),
        int(d['iter_per_site'])) if a != 'new_adam']
    for s in range(S):
        assert np.abs(np.array(res['losses'][s]) - d['r%d_losses' % s]).max() < TOL
    assert len(syncs) == int(d['r0_n_syncs'])
    for i, snap in enumerate(syncs):
        for k, (W, b) in enumerate(snap):
            assert np.abs(W - d['r0_sync%d_W%d' % (i, k)]).max() < TOL
            assert np.abs(b - d['r0_sync%d_b%d' % (i, k)]).max() < TOL
    tail = dict(zip([str(k) for k in d['r0_tail_keys']], d['r0_tail_vals']))
    assert abs(res['val_accs'][-1] - tail['Last Val']) < 1e-4
    assert abs(max(res['val_accs']) - tail['Best Val']) < 1e-4
    assert abs(res['test_accs'][-1] - tail['Last Test']) < 1e-4


@pytest.mark.parametrize('tag', ['ln1_L1', 'ln0_L2'])
def test_G5_train_cora_loop(golden_dir, tag):
    """BASELINE config 1: the oracle's restatement of gcn/train.py's loop (oracle/gcn_oracle.py)
    against a run of the reference's main() on a small citation graph -- per-epoch training
    loss, val/test accuracy per epoch, final parameters, the three printed accuracies."""
    from oracle import gcn_oracle as G
    d = _load(golden_dir, 'G5_train_%s.npz' % tag)
    n, L = int(d['n']), int(d['n_layers'])
    src, dst = G.with_self_loops(d['src'], d['dst'], n)
    assert len(src) == int(d['n_edges_with_loops'])
    g = G.CitationGraph(src, dst, n)
    init = [(d['init_W%d' % k], d['init_b%d' % k]) for k in range(L + 1)]
    losses, record, params = G.train(g, d['feat'], d['label'], d['train_mask'], d['val_mask'],
                                     d['test_mask'], init, bool(d['use_layernorm']), float(d['lr']),
                                     float(d['weight_decay']), int(d['n_epochs']),
                                     lr_scheduler=bool(d['lr_scheduler']))
    assert np.abs(losses - d['losses']).max() < 1e-4
    assert np.allclose([r[0] for r in record], d['val_accs'], atol=1e-9)
    assert np.allclose([r[1] for r in record], d['test_accs'], atol=1e-9)
    for k, (W, b) in enumerate(params):
        assert np.abs(W - d['final_W%d' % k]).max() < 1e-4, k
        assert np.abs(b - d['final_b%d' % k]).max() < 1e-4, k
    tail = dict(zip([str(k) for k in d['tail_keys']], d['tail_vals']))
    assert abs(record[-1][1] - tail['Final Test Accuracy']) < 1e-4
    assert abs(max(r[0] for r in record) - tail['Best Val Accuracy']) < 1e-4
    assert abs(max(r[1] for r in record) - tail['Best Test Accuracy']) < 1e-4
    assert losses[-1] < losses[0]                       # it trains


def test_standard_scaler_matches_sklearn():
    """The oracle's StandardScaler restatement against the installed sklearn (the reference's own
    dependency, cluster_gcn_ist_distrib.py:492-499): zero-variance column, large offsets."""
    sk = pytest.importorskip('sklearn.preprocessing')
    rs = np.random.RandomState(3)
    x = (rs.randn(500, 37) * rs.uniform(0.01, 50, 37) + rs.uniform(-100, 100, 37)).astype(np.float32)
    x[:, 5] = 7.0                                      # zero variance -> scale 1
    mask = rs.rand(500) < 0.6
    scaler = sk.StandardScaler()
    scaler.fit(x[mask])
    want = scaler.transform(x)
    got, mean, var = O.standard_scaler(x, mask)
    assert want.dtype == np.float32
    assert np.allclose(mean, scaler.mean_, rtol=1e-12, atol=1e-12)
    assert np.allclose(var, scaler.var_, rtol=1e-9, atol=1e-12)
    assert np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max())
    assert np.array_equal(got[:, 5], want[:, 5])


def test_preaggregate_is_the_layer0_aggregation(golden_dir):
    """sampler.py:58-69: [X | A^X] equals what ISTSAGELayer computes for its first layer
    (G1 fixture's z = cat(h, ah))."""
    d = _load(golden_dir, 'G1_layer_n64_ln1_act1.npz')
    z = O.preaggregate(d['rowptr'], d['col'], d['h'])
    _, cache = O.sage_layer_forward(d['rowptr'], d['col'], d['h'], d['W'], d['b'], True, True)
    assert np.array_equal(z, cache['z'])
