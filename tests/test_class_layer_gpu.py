"""GPU: the class layer of the fused step as two launches (round 4: gist_class_layer_f32 = projection + CE + dZ with
its dropout mask + the bias gradient's chunk sums; gist_class_dw_slabs_f32 = dW as 128-row slabs) against float64
on the shapes the step meets, and the step with the fused class layer against the four-launch sequence it replaces.

Reference: the last ISTSAGELayer's nn.Linear (cluster_gcn/modules.py:233,299-308), nn.CrossEntropyLoss and their
backward (cluster_gcn_ist_distrib.py:411-415)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def hip():
    from gist_amd import hip as h
    assert h.device_count() >= 1
    return h


def _mask(n, d, p, seed, offset):
    idx = np.arange(n * d, dtype=np.uint64) + np.uint64(offset)
    with np.errstate(over='ignore'):
        z = (idx >> np.uint64(1)) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    w = np.where(idx & np.uint64(1), z >> np.uint64(32), z & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    u = (w >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u >= np.float32(p)).reshape(n, d)


@pytest.mark.parametrize('n,c,k,p,with_dz', [
    (2046, 41, 1024, 0.2, True), (1140, 47, 1024, 0.2, True), (2046, 41, 512, 0.0, True), (2046, 41, 768, 0.2, True),
    (50, 7, 64, 0.5, True), (17, 3, 128, 0.0, True), (300, 48, 192, 0.2, True), (2046, 41, 256, 0.2, True),
    (333, 41, 1024, 0.0, False), (16, 1, 64, 0.2, True)])
def test_class_layer_against_float64(hip, n, c, k, p, with_dz):
    rs = np.random.RandomState(n + c + k)
    ldz, ldc = k + 8, (c + 3) // 4 * 4
    z = (rs.randn(n, ldz) * 0.7).astype(np.float32)
    w = (rs.randn(c, k) / np.sqrt(k) * 3).astype(np.float32)
    b = rs.randn(c).astype(np.float32)
    lab = rs.randint(0, c, n).astype(np.int32)
    zt = torch.from_numpy(z).to(DEV)
    wt, bt, lt = torch.from_numpy(w).to(DEV), torch.from_numpy(b).to(DEV), torch.from_numpy(lab).to(DEV)
    assert hip.class_layer_takes(zt[:, :k], wt, c)
    logits = torch.full((n, ldc), 7.0, device=DEV)
    dlog = torch.full((n, ldc), 7.0, device=DEV)
    row_loss = torch.zeros(n, device=DEV)
    dz = torch.full((n, k + 4), 9.0, device=DEV) if with_dz else None
    chunks = (n + 15) // 16
    part = torch.full((chunks * c,), 5.0, device=DEV)
    seed, off = 1234, 2 * 77
    hip.class_layer(zt[:, :k], wt, bt, lt, n, logits[:, :c], dlog, row_loss, dz[:, :k] if with_dz else None, p, seed,
                    off, part)
    z64, w64 = z[:, :k].astype(np.float64), w.astype(np.float64)
    lg = z64 @ w64.T + b
    mx = lg.max(1, keepdims=True)
    sm = np.exp(lg - mx)
    sm /= sm.sum(1, keepdims=True)
    nll = -(lg[np.arange(n), lab] - mx[:, 0] - np.log(np.exp(lg - mx).sum(1)))
    g = sm.copy()
    g[np.arange(n), lab] -= 1
    g /= n
    assert np.abs(logits[:, :c].cpu().numpy() - lg).max() < 2e-5 * max(1.0, np.abs(lg).max())
    assert np.all(logits[:, c:].cpu().numpy() == 7.0)                    # pad columns of the logits untouched
    assert np.abs(row_loss.cpu().numpy() - nll).max() < 2e-5 * max(1.0, np.abs(nll).max())
    got_g = dlog.cpu().numpy()
    assert np.abs(got_g[:, :c] - g).max() < 1e-6 * max(1.0, np.abs(g).max() * n) / n + 1e-9
    assert np.all(got_g[:, c:] == 0.0)                                   # pad columns of d_logits zeroed
    # bias gradient in 16-row chunks == the chunk sums of the d_logits the kernel wrote, rows in order (bit for bit)
    ref_part = np.zeros((chunks, c), np.float32)
    for ch in range(chunks):
        acc = np.zeros(c, np.float32)
        for r in range(ch * 16, min(n, ch * 16 + 16)):
            acc = (acc + got_g[r, :c]).astype(np.float32)
        ref_part[ch] = acc
    assert np.array_equal(part.cpu().numpy().reshape(chunks, c), ref_part)
    if with_dz:
        ref_dz = g @ w64
        if p > 0:
            ref_dz = ref_dz * _mask(n, k, p, seed, off) / (1.0 - p)
        got = dz.cpu().numpy()
        assert np.abs(got[:, :k] - ref_dz).max() < 2e-5 * max(np.abs(ref_dz).max(), 1e-12)
        if p > 0:
            assert np.array_equal(got[:, :k] == 0, ~_mask(n, k, p, seed, off) | (ref_dz == 0))
        assert np.all(got[:, k:] == 9.0)


@pytest.mark.parametrize('n,c,k', [(2046, 41, 1024), (1140, 47, 1024), (2046, 41, 512), (129, 5, 64), (128, 48, 2048),
                                   (2060, 41, 4096)])      # (the slab kernel itself takes any k % 64 == 0)
def test_class_dw_slabs_against_float64(hip, n, c, k):
    from gist_amd import _lib
    L = _lib.load()
    rs = np.random.RandomState(n + k)
    ldc = (c + 3) // 4 * 4
    g = np.zeros((n, ldc), np.float32)
    g[:, :c] = (rs.randn(n, c) / n).astype(np.float32)
    z = rs.randn(n, k + 4).astype(np.float32)
    nb = int(L.gist_class_dw_slab_bytes(n, c, k))
    assert nb == ((n + 127) // 128) * c * k * 4
    slabs = torch.full((nb // 4 + 16,), 3.0, device=DEV)
    gt, zt = torch.from_numpy(g).to(DEV), torch.from_numpy(z).to(DEV)
    ns = hip.class_dw_slabs(gt[:, :c], zt[:, :k], slabs)
    assert ns == (n + 127) // 128
    s = slabs[:ns * c * k].view(ns, c, k).cpu().numpy().astype(np.float64)
    ref = g[:, :c].astype(np.float64).T @ z[:, :k].astype(np.float64)
    assert np.abs(s.sum(0) - ref).max() < 2e-6 * np.abs(ref).max() + 1e-12
    for j in range(ns):                                   # every slab is its own 128 rows' product
        rj = g[j * 128:(j + 1) * 128, :c].astype(np.float64).T @ z[j * 128:(j + 1) * 128, :k].astype(np.float64)
        assert np.abs(s[j] - rj).max() < 2e-6 * max(np.abs(rj).max(), 1e-12) + 1e-12
    assert np.all(slabs[ns * c * k:].cpu().numpy() == 3.0)


@pytest.mark.parametrize('n,d,c,k,use_ln', [(2046, 512, 41, 1024, True), (1140, 256, 47, 512, True), (333, 1024, 5, 64, False),
                                             (2046, 1028, 41, 1024, True)])      # (d > 1024: no shared grid, two launches inside)
def test_layernorm_backward_and_class_dw_in_one_launch(hip, n, d, c, k, use_ln):
    """gist_ln_relu_bwd_colsum_class_dw_f32 (the LayerNorm backward of the layer below the class layer and the class layer's
    weight-gradient slabs in one grid) == gist_ln_relu_bwd_colsum_f32 + gist_class_dw_slabs_f32, bit for bit: dy, the bias
    gradient's chunk sums, every slab."""
    from gist_amd import _lib
    L = _lib.load()
    gen = torch.Generator(device=DEV).manual_seed(n + d)
    d_out = torch.randn(n, 2 * d, device=DEV, generator=gen)
    yhat = torch.randn(n, d, device=DEV, generator=gen)
    rstd = torch.rand(n, device=DEV, generator=gen) + 0.5
    dlog = torch.zeros(n, 48, device=DEV)
    dlog[:, :c] = torch.randn(n, c, device=DEV, generator=gen) * 1e-3
    z = torch.randn(n, k, device=DEV, generator=gen)
    chunks = int(L.gist_row_chunks16(n))
    n_slabs = (n + 127) // 128
    outs = []
    for fused in (False, True):
        dy = torch.full((n, d), float('nan'), device=DEV)
        part = torch.full((chunks * d,), float('nan'), device=DEV)
        slabs = torch.full((n_slabs * c * k + 5,), 3.0, device=DEV)
        if fused:
            ns = hip.ln_relu_bwd_colsum_class_dw(d_out[:, :d], yhat, rstd if use_ln else None, dy, use_ln, True, part, dlog[:, :c],
                                                 z, slabs)
        else:
            hip.ln_relu_bwd_colsum(d_out[:, :d], yhat, rstd if use_ln else None, dy, use_ln, True, part)
            ns = hip.class_dw_slabs(dlog[:, :c], z, slabs)
        assert ns == n_slabs
        outs.append((dy, part, slabs))
    for a, b in zip(*outs):
        assert not torch.isnan(a).any() and torch.equal(a, b)
    assert torch.all(outs[1][2][n_slabs * c * k:] == 3.0)


@pytest.mark.parametrize('p_drop,n_layers,hidden', [(0.2, 2, 512), (0.0, 2, 512), (0.2, 4, 256), (0.2, 1, 512)])
def test_step_with_fused_class_layer_equals_four_launch_sequence(hip, p_drop, n_layers, hidden):
    """gist_sage_step with the class layer as gist_class_layer_f32 + gist_class_dw_slabs_f32 (the default) against the
    same step with the tuning hook `class_fused` = 1 (projection GEMM, CE kernel, narrow dZ kernel, transposed GEMM):
    GEMM mode f32, 3 steps; same arithmetic up to the order of fp32 sums (k split over four waves / 128-row slabs)."""
    import random
    from gist_amd import datasets
    from gist_amd.engine import SageEngine, dims_for
    from gist_amd.sampler import EngineClusterIter
    prev = hip.gemm_mode()
    hip.gemm_mode('f32')
    try:
        res = []
        for knob in (1, 0):
            hip.tuning('class_fused', knob)
            random.seed(9)
            ds = datasets.toy(seed=9, n=3000, n_blocks=30, n_feats=302, n_classes=5, train_frac=1.0)
            g = ds.g
            it = EngineClusterIter('toy', g, len(ds.par_li), 5, np.arange(g.number_of_nodes(), dtype=np.int64),
                                   par_li=[p.copy() for p in ds.par_li], device=torch.device(DEV))
            dims = dims_for(302, hidden, 5, n_layers)
            eng = SageEngine(dims, True, p_drop, it.n_max, torch.device(DEV), seed=11)
            gen = torch.Generator().manual_seed(1)
            for k, (i, o) in enumerate(dims):
                eng.arena.W[k].copy_((torch.rand(o, 2 * i, generator=gen) - 0.5) * 0.3)
                eng.arena.b[k].copy_((torch.rand(o, generator=gen) - 0.5) * 0.3)
            it.bind(eng)
            losses, logits = [], None
            for j, b in enumerate(it):
                losses.append(eng.train_step(b, 0.01, 5e-4).clone())
                if j == 0:
                    logits = eng.logits(b.n).clone()
                if j == 2:
                    break
            res.append((eng.arena.params.clone(), torch.stack(losses), eng.arena.grads.clone(), logits))
        assert (res[0][3] - res[1][3]).abs().max().item() < 2e-5 * max(1.0, res[0][3].abs().max().item())
        dl = (res[0][1] - res[1][1]).abs()
        assert dl[0].item() < 2e-6 * max(1.0, res[0][1][0].abs().item()) and dl.max().item() < 1e-4
        assert (res[0][2] - res[1][2]).abs().max().item() < 1e-4 * max(1.0, res[0][2].abs().max().item())
        d = (res[0][0] - res[1][0]).abs()
        # (Adam normalises by sqrt(v): a parameter whose gradient is at rounding level moves by a fraction of lr either
        # way; how many there are follows the inputs' last bits -- up to 1.7e-3 of them over the cases here)
        assert float((d > 1e-5).float().mean().item()) < 3e-3 and d.max().item() < 0.02
    finally:
        hip.tuning('class_fused', 0)
        hip.gemm_mode(prev)
