"""GPU: layer 0's aggregation formed by the one-launch extraction (gist_extract_parts_desc.feat_intra / ah, include/gist_hip.h).
A batch is a union of whole parts, so a row's neighbours inside its own part are summed ONCE per run (feat_intra); the
extraction adds the kept neighbours in the batch's other parts and the in-degree norm.  Checked against the aggregation it
replaces (gist_spmm_csr_f32 on the extracted CSR: modules.py:223-226 of the reference, g.update_all(copy_src, sum) * norm)
-- another summation order, so to 1e-5 of the row's magnitude, with everything else of the extraction bit for bit."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


@pytest.fixture(scope='module')
def hip():
    from gist_amd import hip as h
    assert h.device_count() >= 1
    return h


def _iter(kind, n_feats, batch):
    from gist_amd import datasets
    from gist_amd.sampler import EngineClusterIter
    if kind == 'toy':
        ds = datasets.toy(seed=9, n=3000, n_blocks=30, n_feats=n_feats, n_classes=6, train_frac=1.0)
    else:       # hub rows: more kept neighbours outside the row's part than the LDS list holds
        ds = datasets.make_block_dataset('hubs', 6000, 12, n_feats, 6, intra_deg=8, inter_deg=2, seed=3, hub_frac=0.002,
                                         hub_mult=60)
    random.seed(4)
    g = ds.g
    it = EngineClusterIter(kind, g, len(ds.par_li), batch, np.arange(g.number_of_nodes(), dtype=np.int64),
                           par_li=[p.copy() for p in ds.par_li], device=torch.device(DEV))
    return ds, it


def test_intra_part_sums_against_float64(hip):
    """feat_intra[v] = sum over v's in-neighbours inside v's part, against a float64 scatter-add on the host."""
    ds, it = _iter('toy', 50, 5)
    fi = it._intra_part_sums()
    g = it.batcher.g
    rp, col = g.rowptr.cpu().numpy().astype(np.int64), g.col.cpu().numpy().astype(np.int64)
    part = it._part_of_host
    feat = it.batcher.feat.cpu().numpy().astype(np.float64)
    rows = np.repeat(np.arange(len(rp) - 1), rp[1:] - rp[:-1])
    keep = part[rows] == part[col]
    assert 0 < keep.sum() < len(col)
    want = np.zeros_like(feat)
    np.add.at(want, rows[keep], feat[col[keep]])
    assert np.abs(fi.cpu().numpy() - want).max() < 1e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize('kind,n_feats,batch,with_drop', [('toy', 50, 5, False), ('toy', 50, 5, True), ('toy', 301, 5, True),
                                                          ('toy', 64, 1, False), ('hubs', 16, 6, False),
                                                          ('hubs', 260, 6, True),
                                                          ('toy', 640, 5, True),       # (the widest register form: 10 x 64)
                                                          ('toy', 1000, 5, False),     # (16 x 64)
                                                          ('toy', 1100, 5, True)])     # (rows beyond 1024 floats: the loop form)
def test_extraction_forms_layer0_aggregation(hip, kind, n_feats, batch, with_drop):
    """ah of the extraction == norm . A . feat on the extracted CSR (1e-5), the rest of the extraction bit for bit; with
    layer 0's dropout folded in, [h | ah] is bit for bit gist_dropout_f32 of the undropped pair."""
    from gist_amd import _lib
    L = _lib.load()
    ds, it = _iter(kind, n_feats, batch)
    bt = it.batcher
    fi = it._intra_part_sums()
    n_max = it.n_max
    scratch = torch.zeros(int(L.gist_extract_parts_scratch_bytes(n_max)) // 8 + 1, dtype=torch.int64, device=DEV)
    i32 = dict(dtype=torch.int32, device=DEV)
    rp2, trp2 = torch.zeros(n_max + 1, **i32), torch.zeros(n_max + 1, **i32)
    cl2, tcl2 = torch.zeros(bt.col.numel(), **i32), torch.zeros(bt.col.numel(), **i32)
    norm2, lab2 = torch.zeros(n_max, device=DEV), torch.zeros(n_max, **i32)
    ld = n_feats + (2 if n_feats % 4 else 0)
    hubs_seen = 0
    for epoch in range(2):
        it.__iter__()
        for j in range(len(it)):
            a, b = int(it._offsets[j]), int(it._offsets[j + 1])
            ids, n = it._epoch_ids[a:b], b - a
            z_ref = torch.zeros(n, 2 * n_feats, device=DEV)
            ref = bt.extract(ids, z_ref[:, :n_feats])
            hip.spmm(ref.rowptr, ref.col, z_ref[:, :n_feats], z_ref[:, n_feats:], out_scale=ref.norm)
            nnz = int(ref.rowptr[n].item())
            ref_rp, ref_col = ref.rowptr[:n + 1].clone(), ref.col[:nnz].clone()
            z_new = torch.full((n, 2 * n_feats), float('nan'), device=DEV)
            x_new = torch.zeros(n, ld, device=DEV)
            drop = (x_new[:, :n_feats], 0.3, 11, 1000 * j + 2, 2 * n_feats) if with_drop else None
            hip.extract_parts(bt.g, ids, n_max, it._node_part, it._part_tables, j, rp2[:n + 1], cl2, trp2[:n + 1], tcl2, norm2,
                              bt.feat, z_new[:, :n_feats], bt.labels, lab2, scratch, drop=drop, feat_intra=fi,
                              ah=z_new[:, n_feats:])
            assert torch.equal(rp2[:n + 1], ref_rp) and torch.equal(cl2[:nnz], ref_col)
            assert torch.equal(norm2[:n], ref.norm[:n]) and torch.equal(lab2[:n], ref.labels[:n])
            scale = z_ref[:, n_feats:].abs().max().item()
            if with_drop:
                assert torch.equal(x_new[:, :n_feats], z_ref[:, :n_feats])
                # the same launch without the mask gives the undropped pair; the mask is gist_dropout_f32's
                z_plain = torch.full((n, 2 * n_feats), float('nan'), device=DEV)
                hip.extract_parts(bt.g, ids, n_max, it._node_part, it._part_tables, j, rp2[:n + 1], cl2, trp2[:n + 1], tcl2,
                                  norm2, bt.feat, z_plain[:, :n_feats], bt.labels, lab2, scratch, feat_intra=fi,
                                  ah=z_plain[:, n_feats:])
                assert (z_plain - z_ref).abs().max().item() <= 1e-5 * max(1.0, scale)
                hip.dropout_(z_plain, 0.3, 11, 1000 * j + 2)
                assert torch.equal(z_new, z_plain)
                assert 0.2 < float((z_new == 0).float().mean()) < 0.45
            else:
                assert torch.equal(z_new[:, :n_feats], z_ref[:, :n_feats])
                assert (z_new - z_ref).abs().max().item() <= 1e-5 * max(1.0, scale)
            if kind == 'hubs':
                # rows whose kept neighbours OUTSIDE their part exceed the LDS list (64): the list is walked again
                po = it._node_part[:, 0].long()
                rows = torch.repeat_interleave(torch.arange(n, device=DEV), (ref_rp[1:] - ref_rp[:-1]).long())
                outside = po[ids.long()][rows] != po[ids.long()][ref_col.long()]
                per_row = torch.zeros(n, dtype=torch.int64, device=DEV).index_add_(0, rows[outside],
                                                                                   torch.ones_like(rows[outside]))
                hubs_seen += int((per_row > 64).sum().item())
        random.shuffle(it.par_li)
    assert int(scratch[1].item()) == 0
    if kind == 'hubs':
        assert hubs_seen > 0


@pytest.mark.parametrize('p_drop,n_layers', [(0.0, 2), (0.2, 3)])
def test_training_with_and_without_preaggregation(hip, monkeypatch, p_drop, n_layers):
    """4 steps of the native step with layer 0's aggregation from the extraction against the same steps with the launch it
    replaces (GIST_STEP_PREAGG=0): losses to 1e-5, one aggregation launch fewer per step on the native timer."""
    from gist_amd import datasets
    from gist_amd.engine import SageEngine, dims_for
    from gist_amd.sampler import EngineClusterIter
    ds = datasets.toy(seed=9, n=3000, n_blocks=30, n_feats=50, n_classes=6, train_frac=1.0)
    g = ds.g
    dims = dims_for(50, 96, 6, n_layers)
    out = {}
    for pre in ('1', '0'):
        monkeypatch.setenv('GIST_STEP_PREAGG', pre)
        random.seed(4)
        it = EngineClusterIter('toy', g, len(ds.par_li), 5, np.arange(g.number_of_nodes(), dtype=np.int64),
                               par_li=[p.copy() for p in ds.par_li], device=DEV)
        eng = SageEngine(dims, True, p_drop, it.n_max, DEV, seed=11)
        gen = torch.Generator().manual_seed(1)
        for k, (i, o) in enumerate(dims):
            s_ = 1.0 / np.sqrt(2 * i)
            eng.arena.W[k].copy_((torch.rand(o, 2 * i, generator=gen) - 0.5) * 2 * s_)
            eng.arena.b[k].copy_((torch.rand(o, generator=gen) - 0.5) * 2 * s_)
        it.bind(eng)
        assert (it.batcher.feat_intra is not None) == (pre == '1') and bool(eng.plan.feat_intra) == (pre == '1')
        eng.enable_timer(1000)
        losses = []
        for j, b in enumerate(it):
            losses.append(float(eng.train_step(b, 0.01, 5e-4).item()))
            if j == 3:
                break
        out[pre] = (losses, sum(1 for ms, kind, *_ in eng.read_timer() if kind == 0), eng.arena.params.clone())
        eng.disable_timer()
        eng.check_extract()
    (l1, a1, p1), (l0, a0, p0) = out['1'], out['0']
    assert a0 - a1 == 4                                   # layer 0's aggregation launch is gone from every step
    assert max(abs(x - y) for x, y in zip(l1, l0)) <= 1e-5 * max(1.0, max(abs(y) for y in l0))
    assert (p1 - p0).abs().max().item() < 1e-4


def test_refresh_after_features_changed_in_place(hip):
    """The part-internal sums are formed once, at bind(); refresh_input_aggregation() re-forms them (same buffer) after an
    in-place change of the features: the next extraction's ah follows the new features."""
    from gist_amd.engine import SageEngine, dims_for
    ds, it = _iter('toy', 64, 5)
    eng = SageEngine(dims_for(64, 96, 6, 2), True, 0.0, it.n_max, DEV, seed=1)
    it.bind(eng)
    fi = it.batcher.feat_intra
    assert fi is not None
    before, ptr = fi.clone(), fi.data_ptr()
    it.batcher.feat.mul_(2.0)
    it.refresh_input_aggregation()
    assert it.batcher.feat_intra.data_ptr() == ptr
    assert torch.equal(it.batcher.feat_intra, before * 2.0)          # (a sum of doubled terms: exact)
    b = next(iter(it))
    eng.forward(b, training=False)
    n, f = b.n, 64
    want = torch.zeros(n, f, device=DEV)
    hip.spmm(b.rowptr, b.col, eng.Z[0][:n, :f], want, out_scale=b.norm)
    assert (eng.Z[0][:n, f:2 * f] - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    it.batcher.feat.mul_(0.5)
