"""GPU: the block-dense aggregation on the fp32 matrix cores (spmm_dense32.hip; what gist_spmm_csr_prepared_f32 and
gist_spmm_csr_drop_prepared_f32 run below 1536 columns and for rows that are not 16-byte aligned) against the
oracle, against the row-split kernel bit for bit on integer-valued features, and its folded dropout masks against
gist_dropout_f32 around the plain call.  Graphs: multigraphs with most edges inside the row blocks, remote edges,
hub rows (more than 8 outside neighbours: gathered in full), a 300-fold edge (a count above 256), an oversized
block and blocks of 1 and 128 rows.

Reference: g.update_all(fn.copy_src, fn.sum) * norm, cluster_gcn/modules.py:223-226, and its autograd."""
import numpy as np
import pytest
import torch

from oracle import gist_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def hip():
    """The library routes a prepared aggregation to the fp32 block-dense kernel only for widths the other blocked
    kernels do not take (rows that are not 16-byte aligned: the F = 602 input layer); the tuning hook spmm_kernel = 3
    sends EVERY prepared call there, so that the kernel is tested at all widths."""
    from gist_amd import hip as h
    assert h.device_count() >= 1
    h.tuning('spmm_kernel', 3)
    yield h
    h.tuning('spmm_kernel', 0)


def _graph(n, bounds, deg_in, deg_out, seed, hub=0, fold=0):
    rs = np.random.RandomState(seed)
    bounds = np.asarray(bounds, np.int64)
    blk = np.searchsorted(bounds, np.arange(n), side='right') - 1
    src, dst = [], []
    for v in range(n):
        lo, hi = bounds[blk[v]], bounds[blk[v] + 1]
        src.append(rs.randint(lo, hi, deg_in))
        dst.append(np.full(deg_in, v))
        src.append(rs.randint(0, n, deg_out))
        dst.append(np.full(deg_out, v))
    if hub:                                  # row 3 with many neighbours everywhere: leaves the dense product
        src.append(rs.randint(0, n, hub))
        dst.append(np.full(hub, 3))
    if fold:                                 # one edge `fold` times: a count the bf16 image cannot hold exactly
        src.append(np.full(fold, 5))
        dst.append(np.full(fold, 7))
    src, dst = np.concatenate(src), np.concatenate(dst)
    rp, cl = O.csr_from_edges(src, dst, n)
    trp, tcl = O.csr_from_edges(dst, src, n)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(torch.int32).to(DEV)
    return (rp, cl, trp, tcl), (t(rp), t(cl), t(trp), t(tcl), t(bounds))


CASES = [
    # n, block bounds, D, ld pad
    (700, np.linspace(0, 700, 8).astype(int), 512, 0),
    (700, np.linspace(0, 700, 8).astype(int), 602, 2),          # the Reddit input layer: rows 8-byte aligned only
    (333, [0, 1, 129, 130, 258, 333], 256, 0),                  # blocks of 1 and 128 rows
    (500, [0, 100, 400, 500], 100, 0),                          # an oversized block (300 rows), Amazon's F = 100
    (257, np.linspace(0, 257, 4).astype(int), 16, 0),
    (900, np.linspace(0, 900, 10).astype(int), 1024, 0),
    (130, [0, 64, 130], 41, 3),
    (2046, np.linspace(0, 2046, 21).astype(int), 256, 0),       # the config-2 batch shape
]


@pytest.mark.parametrize('n,bounds,d,pad', CASES)
def test_dense32_against_oracle_forward_and_backward(hip, n, bounds, d, pad):
    host, (rp, cl, trp, tcl, rb) = _graph(n, bounds, 12, 2, seed=n + d, hub=40, fold=300)
    rs = np.random.RandomState(d)
    x = rs.randn(n, d + pad).astype(np.float32)
    xt = torch.from_numpy(x).to(DEV)[:, :d]
    norm = hip.in_degree_norm(rp)
    nrm = norm.cpu().numpy()
    prep = hip.spmm_prepare(rp, cl, rb)
    prep_t = hip.spmm_prepare(trp, tcl, rb)
    # forward: y = norm * sum of in-neighbours
    y = torch.full((n, d + 4), 3.0, device=DEV)
    hip.spmm(rp, cl, xt, y[:, :d], out_scale=norm, row_blocks=rb, prepared=prep)
    ref = O.spmm_sum(host[0], host[1], x[:, :d].astype(np.float64).astype(np.float32), out_scale=nrm)
    ref64 = np.zeros((n, d))
    rows = np.repeat(np.arange(n), np.diff(host[0]))
    np.add.at(ref64, rows, x[host[1], :d].astype(np.float64))
    ref64 *= nrm[:, None]
    got = y.cpu().numpy()
    assert np.abs(got[:, :d] - ref64).max() < 2e-5 * max(1.0, np.abs(ref64).max())
    assert np.abs(got[:, :d] - ref).max() < 1e-4
    assert np.all(got[:, d:] == 3.0)
    # backward: y += sum over out-neighbours of norm[u] * x[u]
    acc0 = rs.randn(n, d).astype(np.float32)
    yb = torch.from_numpy(acc0).to(DEV)
    hip.spmm(trp, tcl, xt, yb, src_scale=norm, accumulate=True, row_blocks=rb, prepared=prep_t)
    refb = acc0.astype(np.float64)
    rows_t = np.repeat(np.arange(n), np.diff(host[2]))
    np.add.at(refb, rows_t, (x[:, :d].astype(np.float64) * nrm[:, None])[host[3]])
    assert np.abs(yb.cpu().numpy() - refb).max() < 2e-5 * max(1.0, np.abs(refb).max())


@pytest.mark.parametrize('n,bounds,d,pad', CASES[:4])
def test_dense32_exact_on_integer_features(hip, n, bounds, d, pad):
    """Counts x values are exact fp32 products and sums of small integers are exact in any order: the dense kernel
    must equal the row-split kernel BIT FOR BIT (including the 300-fold edge, the hub row and the oversized block)."""
    _, (rp, cl, trp, tcl, rb) = _graph(n, bounds, 12, 2, seed=3 * n + d, hub=40, fold=300)
    rs = np.random.RandomState(1)
    x = torch.from_numpy(rs.randint(-64, 64, (n, d + pad)).astype(np.float32)).to(DEV)[:, :d]
    prep = hip.spmm_prepare(rp, cl, rb)
    a = torch.zeros(n, d, device=DEV)
    b = torch.zeros(n, d, device=DEV)
    hip.spmm(rp, cl, x, a, row_blocks=rb, prepared=prep)
    hip.spmm(rp, cl, x, b)                                   # gist_spmm_csr_f32: the row-split / lane-group kernels
    assert torch.equal(a, b)


@pytest.mark.parametrize('n,d', [(700, 512), (333, 128), (900, 1024), (700, 602), (300, 100)])
def test_dense32_masks_equal_dropout_around_the_plain_call(hip, n, d):
    """Mode 1: what is stored = dropout(aggregate(x)) bit for bit; mode 2: x read through its mask and the old y
    through y's = gist_dropout_f32 on [dZ_left | dZ_right] in front of the plain prepared call, bit for bit."""
    _, (rp, cl, trp, tcl, rb) = _graph(n, np.linspace(0, n, 7).astype(int), 10, 3, seed=n, hub=30)
    gen = torch.Generator(device=DEV).manual_seed(2)
    norm = hip.in_degree_norm(rp)
    prep, prep_t = hip.spmm_prepare(rp, cl, rb), hip.spmm_prepare(trp, tcl, rb)
    ld = 2 * d if d % 2 == 0 else 2 * d + 1
    # mode 1
    x = torch.randn(n, d, device=DEV, generator=gen)
    z_ref = torch.zeros(n, 2 * d, device=DEV)
    z_new = torch.zeros(n, 2 * d, device=DEV)
    hip.spmm(rp, cl, x, z_ref[:, d:], out_scale=norm, row_blocks=rb, prepared=prep)
    hip.dropout_(z_ref, 0.3, 17, 1000)
    hip.spmm_drop(rp, cl, x, z_new[:, d:], 1, 0.3, 17, 1000 + d, 0, 2 * d, out_scale=norm, row_blocks=rb,
                  prepared=prep)
    assert torch.equal(z_new[:, d:], z_ref[:, d:])
    assert 0.2 < float((z_new[:, d:] == 0).float().mean().item()) < 0.45
    # mode 2
    if d % 4 == 0 and d >= 128:               # (the shapes the step folds: gist_spmm_drop_takes)
        dz = torch.randn(n, 2 * d, device=DEV, generator=gen)
        ref, new = dz.clone(), dz.clone()
        hip.dropout_(ref, 0.25, 5, 64)
        hip.spmm(trp, tcl, ref[:, d:], ref[:, :d], src_scale=norm, accumulate=True, row_blocks=rb, prepared=prep_t)
        hip.spmm_drop(trp, tcl, new[:, d:], new[:, :d], 2, 0.25, 5, 64, 64 + d, 2 * d, src_scale=norm,
                      accumulate=True, row_blocks=rb, prepared=prep_t)
        # (to rounding: this kernel is never the step's backward form -- the widths it takes by default are not folded,
        # gist_spmm_drop_takes -- and its mask-then-scale products are not contracted like the plain call's)
        assert (new[:, :d] - ref[:, :d]).abs().max().item() <= 2e-6 * max(1.0, ref[:, :d].abs().max().item())
        assert torch.equal(new[:, :d] == 0, ref[:, :d] == 0)


def test_dense32_row_tile_groups_agree(hip):
    """The launcher's split of a block's row tiles over 2 or 4 waves changes no value (tuning hook spmm_split)."""
    n, d = 700, 256
    _, (rp, cl, _, _, rb) = _graph(n, np.linspace(0, n, 8).astype(int), 12, 2, seed=5, hub=40)
    x = torch.randn(n, d, device=DEV)
    prep = hip.spmm_prepare(rp, cl, rb)
    outs = []
    try:
        for g in (2, 4):
            hip.tuning('spmm_split', g)
            y = torch.zeros(n, d, device=DEV)
            hip.spmm(rp, cl, x, y, row_blocks=rb, prepared=prep)
            outs.append(y)
    finally:
        hip.tuning('spmm_split', 0)
    assert torch.equal(outs[0], outs[1])


def test_default_routing_of_prepared_calls(hip):
    """Without the hook: the unaligned input width runs on the fp32 block-dense kernel when the batch's blocks are
    prepared (bit-equal to the forced call), an aligned narrow width keeps the LDS-gather kernel (bit-equal to the
    unprepared blocked call)."""
    n = 700
    _, (rp, cl, _, _, rb) = _graph(n, np.linspace(0, n, 8).astype(int), 12, 2, seed=9, hub=40)
    prep = hip.spmm_prepare(rp, cl, rb)
    x602 = torch.randn(n, 604, device=DEV)[:, :602]
    x512 = torch.randn(n, 512, device=DEV)
    forced = torch.zeros(n, 602, device=DEV)
    hip.spmm(rp, cl, x602, forced, row_blocks=rb, prepared=prep)          # (hook = 3 from the fixture)
    try:
        hip.tuning('spmm_kernel', 0)
        a = torch.zeros(n, 602, device=DEV)
        hip.spmm(rp, cl, x602, a, row_blocks=rb, prepared=prep)
        assert torch.equal(a, forced)
        b, c = torch.zeros(n, 512, device=DEV), torch.zeros(n, 512, device=DEV)
        hip.spmm(rp, cl, x512, b, row_blocks=rb, prepared=prep)
        hip.spmm(rp, cl, x512, c, row_blocks=rb)
        assert torch.equal(b, c)
    finally:
        hip.tuning('spmm_kernel', 3)
