"""GPU: the fused forms of the step's kernels (include/gist_hip.h: dropout folded into producers and
consumers, bias gradients in row chunks, split-K slabs consumed by the loss kernel / the optimiser, the
one-launch extraction) against the un-fused C-ABI calls they replace -- BIT FOR BIT wherever the fused
form promises the same arithmetic, against float64 / the oracle elsewhere."""
import ctypes
import random

import numpy as np
import pytest
import torch

from oracle import gist_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = 1e-4


@pytest.fixture(scope='module')
def hip():
    from gist_amd import hip as h
    assert h.device_count() >= 1
    return h


def block_graph(n, n_blocks, deg_in, deg_out, seed, hub=0):
    """Rows in `n_blocks` consecutive blocks, most neighbours inside the row's block; returns the CSR,
    its transpose and the block boundaries."""
    rs = np.random.RandomState(seed)
    bounds = np.linspace(0, n, n_blocks + 1).astype(np.int64)
    blk = np.searchsorted(bounds, np.arange(n), side='right') - 1
    src, dst = [], []
    for v in range(n):
        lo, hi = bounds[blk[v]], bounds[blk[v] + 1]
        src.append(rs.randint(lo, hi, deg_in))
        dst.append(np.full(deg_in, v))
        src.append(rs.randint(0, n, deg_out))
        dst.append(np.full(deg_out, v))
    if hub:
        src.append(rs.randint(0, n, hub))
        dst.append(np.full(hub, 3))
    src, dst = np.concatenate(src), np.concatenate(dst)
    rp, cl = O.csr_from_edges(src, dst, n)
    trp, tcl = O.csr_from_edges(dst, src, n)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(torch.int32).to(DEV)
    return t(rp), t(cl), t(trp), t(tcl), t(bounds)


@pytest.mark.parametrize('n,d,blocks', [(700, 512, True), (700, 512, False), (300, 602, False), (300, 602, True),
                                        (257, 100, True), (130, 41, False), (900, 1024, True), (200, 7, False),
                                        (700, 2048, True), (1100, 4096, True)])      # (the matrix-core kernel)
def test_spmm_forward_mask_equals_dropout_after(hip, n, d, blocks):
    """mode 1: y = dropout(aggregate(x)) from the aggregation's own store == the plain aggregation
    followed by gist_dropout_f32 on y's columns of the wider tensor, on every kernel that takes it."""
    rp, cl, _, _, rb = block_graph(n, 7, 12, 2, seed=n + d, hub=200)
    gen = torch.Generator(device=DEV).manual_seed(1)
    ld = d if d % 4 == 0 else d + 2
    x = torch.randn(n, ld, device=DEV, generator=gen)[:, :d]
    norm = hip.in_degree_norm(rp)
    z_ref = torch.zeros(n, 2 * d, device=DEV)
    z_new = torch.zeros(n, 2 * d, device=DEV)
    if not hip.spmm_drop_takes(1, d, x, z_new[:, d:], blocks):
        pytest.skip('this shape keeps the separate dropout pass')
    hip.spmm(rp, cl, x, z_ref[:, d:], out_scale=norm, row_blocks=rb if blocks else None)
    hip.dropout_(z_ref, 0.3, 17, 1000)
    hip.spmm_drop(rp, cl, x, z_new[:, d:], 1, 0.3, 17, 1000 + d, 0, 2 * d, out_scale=norm,
                  row_blocks=rb if blocks else None)
    assert torch.equal(z_new[:, d:], z_ref[:, d:])
    frac = float((z_new[:, d:] == 0).float().mean().item())
    assert 0.2 < frac < 0.45          # the mask is really applied


@pytest.mark.parametrize('n,d', [(700, 512), (333, 128), (900, 1024)])
def test_spmm_backward_masks_equal_dropout_before(hip, n, d):
    """mode 2: the reverse aggregation of a gradient whose dropout pass has not run == dropout on
    [dZ_left | dZ_right], then dZ_left += A^T (norm * dZ_right)."""
    rp, cl, trp, tcl, rb = block_graph(n, 6, 10, 3, seed=n, hub=300)
    gen = torch.Generator(device=DEV).manual_seed(2)
    dz = torch.randn(n, 2 * d, device=DEV, generator=gen)
    norm = hip.in_degree_norm(rp)
    ref = dz.clone()
    new = dz.clone()
    assert hip.spmm_drop_takes(2, d, new[:, d:], new[:, :d], True)
    hip.dropout_(ref, 0.25, 5, 64)
    hip.spmm(trp, tcl, ref[:, d:], ref[:, :d], src_scale=norm, accumulate=True, row_blocks=rb)
    hip.spmm_drop(trp, tcl, new[:, d:], new[:, :d], 2, 0.25, 5, 64, 64 + d, 2 * d, src_scale=norm,
                  accumulate=True, row_blocks=rb)
    assert torch.equal(new[:, :d], ref[:, :d])


@pytest.mark.parametrize('n,d,use_ln', [(513, 512, True), (100, 96, True), (77, 1024, False), (300, 4096, True),
                                        (50, 37, True)])
def test_ln_relu_fwd_drop(hip, n, d, use_ln):
    """out = dropout(relu(LN(y))) under the next layer's mask, out2 = the undropped activations, yhat
    in place: equal to gist_ln_relu_fwd_f32 + gist_dropout_f32 on the left half of the next [h | ah]."""
    gen = torch.Generator(device=DEV).manual_seed(3)
    y = torch.randn(n, d, device=DEV, generator=gen) * 3 + 1
    y2 = y.clone()
    z_ref = torch.zeros(n, 2 * d, device=DEV)
    z_new = torch.zeros(n, 2 * d, device=DEV)
    h = torch.zeros(n, d, device=DEV)
    rstd_a, rstd_b = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    hip.ln_relu_fwd(y, z_ref[:, :d], rstd_a, use_ln, True)
    undropped = z_ref[:, :d].clone()
    hip.dropout_(z_ref, 0.2, 9, 4242)
    hip.ln_relu_fwd_drop(y2, z_new[:, :d], h, rstd_b, use_ln, True, 0.2, 9, 4242, 2 * d)
    assert torch.equal(z_new[:, :d], z_ref[:, :d])
    assert torch.equal(h, undropped)
    assert torch.equal(y2, y) and torch.equal(rstd_a, rstd_b)
    # p = 0 with a second output: two copies of the plain result
    y3 = torch.randn(n, d, device=DEV, generator=gen)
    y4 = y3.clone()
    a, b, c = (torch.zeros(n, d, device=DEV) for _ in range(3))
    hip.ln_relu_fwd(y3, a, None, False, True)
    hip.ln_relu_fwd_drop(y4, b, c, None, False, True, 0.0, 0, 0, d)
    assert torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.parametrize('m,n,k,p', [(2046, 256, 512, 0.0), (1140, 512, 1024, 0.3), (700, 2048, 260, 0.2),
                                     (333, 96, 1204, 0.0), (513, 4096, 192, 0.25), (130, 37, 300, 0.0)])      # (37: scalar path)
def test_ln_from_projection_slabs(hip, m, n, k, p):
    """gist_gemm_slabs_f32 (NT; the model's slice count, then 4 forced) + gist_ln_relu_fwd_slabs_f32 == the slabs
    summed in slab order + bias (with the same slice count: gist_gemm_nt_f32's own result) followed by
    gist_ln_relu_fwd_drop_f32, bit for bit: yhat, rstd, the dropped output and the undropped copy."""
    prev = hip.gemm_mode()
    hip.gemm_mode('f32')
    gen = torch.Generator(device=DEV).manual_seed(m + n)
    a = torch.randn(m, k, device=DEV, generator=gen)
    w = torch.randn(n, k, device=DEV, generator=gen) / np.sqrt(k)
    bias = torch.randn(n, device=DEV, generator=gen)
    try:
        for splits in (0, 4):
            hip.tuning('gemm_tile', 64 if splits else 0)
            hip.tuning('gemm_splits', splits)
            y = torch.full((m, n), float('nan'), device=DEV)
            slabs = torch.full((8 * m * n,), float('nan'), device=DEV)
            ns = hip.gemm_slabs('nt', a, w, bias, y, slabs.view(torch.uint8))
            assert (ns in (3, 4)) if splits else ns >= 1      # (k = 260: slices of 128 -> 3)
            if ns > 1:       # the slabs in slab order, then the bias: gist_gemm's own reduce pass
                y_ref = torch.zeros(m, n, device=DEV)
                for q in range(ns):
                    y_ref = y_ref + slabs[q * m * n:(q + 1) * m * n].view(m, n)
                y_ref = y_ref + bias
                if splits:   # ... which the self-reducing call with the same slice count reproduces
                    y_call = torch.empty(m, n, device=DEV)
                    hip.gemm_nt(a, w, bias, y_call)
                    assert torch.equal(y_call, y_ref)
            else:
                y_ref = y.clone()
            o_ref, o2_ref, r_ref = torch.zeros(m, 2 * n, device=DEV), torch.zeros(m, n, device=DEV), torch.zeros(m, device=DEV)
            hip.ln_relu_fwd_drop(y_ref, o_ref[:, :n], o2_ref, r_ref, True, True, p, 9, 128, 2 * n)
            o, o2, r = torch.zeros(m, 2 * n, device=DEV), torch.zeros(m, n, device=DEV), torch.zeros(m, device=DEV)
            if ns > 1:
                hip.ln_relu_fwd_slabs(y, slabs, ns, bias, o[:, :n], o2, r, True, True, p, 9, 128, 2 * n)
            else:
                hip.ln_relu_fwd_slabs(y, None, 0, None, o[:, :n], o2, r, True, True, p, 9, 128, 2 * n)
            assert torch.equal(y, y_ref) and torch.equal(r, r_ref)
            assert torch.equal(o, o_ref) and torch.equal(o2, o2_ref)
    finally:
        hip.tuning('gemm_tile', 0)
        hip.tuning('gemm_splits', 0)
        hip.gemm_mode(prev)


@pytest.mark.parametrize('n,d,use_ln', [(2046, 512, True), (1000, 1024, True), (37, 96, True), (300, 256, False),
                                        (200, 2048, True), (65, 50, True)])
def test_ln_relu_bwd_colsum(hip, n, d, use_ln):
    """dy equal to gist_ln_relu_bwd_f32's up to the contraction of its row sums (another instruction
    schedule, same formula: a few ulp); the 16-row chunk sums add up to the column sums of dy (float64
    check) and gist_colsum_chunks_f32 is their sum."""
    from gist_amd import _lib
    L = _lib.load()
    gen = torch.Generator(device=DEV).manual_seed(4)
    g = torch.randn(n, d, device=DEV, generator=gen)
    yhat = torch.randn(n, d, device=DEV, generator=gen)
    rstd = torch.rand(n, device=DEV, generator=gen) + 0.5
    dy_ref = torch.empty(n, d, device=DEV)
    dy_new = torch.empty(n, d, device=DEV)
    chunks = int(L.gist_row_chunks16(n))
    part = torch.full((chunks * d,), float('nan'), device=DEV)
    hip.ln_relu_bwd(g, yhat, rstd if use_ln else None, dy_ref, use_ln, True)
    hip.ln_relu_bwd_colsum(g, yhat, rstd if use_ln else None, dy_new, use_ln, True, part)
    assert (dy_new - dy_ref).abs().max().item() <= 2e-6 * max(1.0, dy_ref.abs().max().item())
    p2 = part.view(chunks, d).double()
    pad = torch.zeros(chunks * 16, d, dtype=torch.float64, device=DEV)
    pad[:n] = dy_new.double()
    want = pad.view(chunks, 16, d).sum(1)
    assert (p2 - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    out = torch.empty(d, device=DEV)
    hip.colsum_chunks(part, chunks, d, out)
    ref = dy_new.double().sum(0)
    assert (out.double() - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('m,n,k,layout', [(2046, 41, 1024, 'nt'), (41, 1024, 2046, 'tn'), (512, 1204, 2046, 'tn'),
                                          (300, 64, 96, 'nt'), (2046, 47, 8192, 'nt')])
def test_gemm_slabs_sum_to_the_projection(hip, m, n, k, layout):
    """gist_gemm_slabs_f32: the slabs summed in slab order (+ bias) are BIT-equal to the same call through
    gist_gemm_nt/tn_f32 (which runs the reduce pass itself)."""
    from gist_amd import _lib
    L = _lib.load()
    prev = hip.gemm_mode()
    hip.gemm_mode('f32')
    try:
        gen = torch.Generator(device=DEV).manual_seed(5)
        if layout == 'nt':
            a = torch.randn(m, k, device=DEV, generator=gen)
            b = torch.randn(n, k, device=DEV, generator=gen)
            bias = torch.randn(n, device=DEV, generator=gen)
        else:
            a = torch.randn(k, m, device=DEV, generator=gen)
            b = torch.randn(k, n, device=DEV, generator=gen)
            bias = None
        ref = torch.empty(m, n, device=DEV)
        (hip.gemm_nt(a, b, bias, ref) if layout == 'nt' else hip.gemm_tn(a, b, ref))
        need = int(L.gist_gemm_workspace_bytes(m, n, k))
        slabs = torch.full((max(need, 16) // 4,), float('nan'), device=DEV)
        c = torch.full((m, n), float('nan'), device=DEV)
        ns = hip.gemm_slabs(layout, a, b, bias, c, slabs.view(torch.uint8) if need else None)
        if need == 0:
            assert ns == 1
        if ns == 1:
            assert torch.equal(c, ref)
        else:
            acc = torch.zeros(m * n, device=DEV)
            for s in range(ns):
                acc = acc + slabs[s * m * n:(s + 1) * m * n]
            got = acc.view(m, n)
            if bias is not None:
                got = got + bias
            assert torch.equal(got, ref)
    finally:
        hip.gemm_mode(prev)


def test_xent_from_slabs_and_loss_in_adam(hip):
    """The loss kernel fed with split-K slabs == the loss kernel fed with the reduced logits; the loss
    left to gist_adam_segments_f32 == the one gist_softmax_xent_f32 reduces itself."""
    n, c, ns = 1234, 41, 5
    gen = torch.Generator(device=DEV).manual_seed(6)
    slabs = torch.randn(ns, n, c, device=DEV, generator=gen)
    bias = torch.randn(c, device=DEV, generator=gen)
    labels = torch.randint(0, c, (n,), device=DEV, generator=gen).to(torch.int32)
    acc = torch.zeros(n, c, device=DEV)
    for s in range(ns):
        acc = acc + slabs[s]
    logits_ref = torch.zeros(n, 44, device=DEV)
    logits_ref[:, :c] = acc + bias
    rl_a, rl_b = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    loss_a, loss_b = torch.zeros(1, device=DEV), torch.full((1,), float('nan'), device=DEV)
    dl_a, dl_b = torch.zeros(n, 44, device=DEV), torch.zeros(n, 44, device=DEV)
    hip.softmax_xent(logits_ref[:, :c], labels, None, n, rl_a, loss_a, dl_a)
    logits_new = torch.zeros(n, 44, device=DEV)
    hip.softmax_xent_slabs(logits_new[:, :c], slabs.view(-1), ns, bias, labels, None, n, rl_b, None, dl_b)
    assert torch.equal(logits_new, logits_ref) and torch.equal(rl_a, rl_b) and torch.equal(dl_a, dl_b)
    # Adam with no segment but the loss: parameters as gist_adam_f32's, loss as the loss kernel's
    P = 5000
    prm = torch.randn(P, device=DEV, generator=gen)
    grd = torch.randn(P, device=DEV, generator=gen)
    st = [torch.zeros(P, device=DEV) for _ in range(4)]
    p1, p2 = prm.clone(), prm.clone()
    hip.adam_(p1, grd, st[0], st[1], 3, 0.01, weight_decay=5e-4)
    hip.adam_segments_(p2, grd.clone(), st[2], st[3], 3, 0.01, [], row_loss=rl_b, n_loss_rows=n, loss_count=n,
                       loss=loss_b, weight_decay=5e-4)
    assert torch.equal(p1, p2) and torch.equal(loss_a, loss_b)
    ref = torch.nn.functional.cross_entropy(logits_ref[:, :c], labels.long())
    assert abs(loss_b.item() - ref.item()) < 1e-5


def test_adam_segments_equal_reduce_then_adam(hip):
    """Slab and chunk-sum segments inside the optimiser == reducing them first (slabs in slab order,
    chunks by gist_colsum_chunks_f32) and running gist_adam_f32; grad holds the reduced gradient."""
    gen = torch.Generator(device=DEV).manual_seed(7)
    o, i2, c, chunks = 96, 200, 41, 128
    sizes = [o * i2, o, c * 64, c, 777]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    P = int(offs[-1])
    prm = torch.randn(P, device=DEV, generator=gen)
    grd = torch.randn(P, device=DEV, generator=gen)
    slabs_w = torch.randn(6, o * i2, device=DEV, generator=gen)
    part_b = torch.randn(chunks, o, device=DEV, generator=gen)
    slabs_w2 = torch.randn(3, c * 64, device=DEV, generator=gen)
    part_c = torch.randn(17, c, device=DEV, generator=gen)
    # reference: reduce, then plain Adam
    g_ref = grd.clone()
    acc = torch.zeros(o * i2, device=DEV)
    for s in range(6):
        acc = acc + slabs_w[s]
    g_ref[offs[0]:offs[1]] = acc
    tmp = torch.empty(o, device=DEV)
    hip.colsum_chunks(part_b.view(-1), chunks, o, tmp)
    g_ref[offs[1]:offs[2]] = tmp
    acc = torch.zeros(c * 64, device=DEV)
    for s in range(3):
        acc = acc + slabs_w2[s]
    g_ref[offs[2]:offs[3]] = acc
    tmp = torch.empty(c, device=DEV)
    hip.colsum_chunks(part_c.view(-1), 17, c, tmp)
    g_ref[offs[3]:offs[4]] = tmp
    m1, v1, m2, v2 = (torch.rand(P, device=DEV, generator=gen) * 0.01 for _ in range(4))
    m2.copy_(m1)
    v2.copy_(v1)
    p1, p2 = prm.clone(), prm.clone()
    hip.adam_(p1, g_ref, m1, v1, 2, 0.01, weight_decay=1e-3)
    g_new = grd.clone()
    segs = [(int(offs[0]), int(offs[1]), slabs_w, o * i2, 6), (int(offs[1]), int(offs[2]), part_b, o, chunks),
            (int(offs[2]), int(offs[3]), slabs_w2, c * 64, 3), (int(offs[3]), int(offs[4]), part_c, c, 17)]
    hip.adam_segments_(p2, g_new, m2, v2, 2, 0.01, segs, weight_decay=1e-3)
    assert torch.equal(g_new, g_ref)
    assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2)
    want = part_b.double().sum(0)
    assert (g_new[offs[1]:offs[2]].double() - want).abs().max().item() < 1e-4 * want.abs().max().item()


def test_class_layer_dz_kernel_leaves_bias_chunks(hip):
    from gist_amd import _lib
    L = _lib.load()
    n, c, k2 = 1000, 41, 1024
    gen = torch.Generator(device=DEV).manual_seed(8)
    dy = torch.zeros(n, 44, device=DEV)
    dy[:, :c] = torch.randn(n, c, device=DEV, generator=gen)
    w = torch.randn(c, k2, device=DEV, generator=gen)
    z1, z2 = torch.empty(n, k2, device=DEV), torch.empty(n, k2, device=DEV)
    chunks = int(L.gist_row_chunks16(n))
    part = torch.full((chunks * c,), float('nan'), device=DEV)
    hip.gemm_nn_dropout_(dy[:, :c], w, z1, 0.2, 3, 10)
    hip.gemm_nn_dropout_colsum_(dy[:, :c], w, z2, 0.2, 3, 10, part)
    assert torch.equal(z1, z2)
    pad = torch.zeros(chunks * 16, c, dtype=torch.float64, device=DEV)
    pad[:n] = dy[:, :c].double()
    want = pad.view(chunks, 16, c).sum(1)
    assert (part.view(chunks, c).double() - want).abs().max().item() < 1e-5


def _toy_iter(seed, n, n_blocks, n_feats, batch):
    from gist_amd import datasets
    from gist_amd.sampler import EngineClusterIter
    ds = datasets.toy(seed=seed, n=n, n_blocks=n_blocks, n_feats=n_feats, n_classes=6, train_frac=1.0)
    g = ds.g
    random.seed(4)
    it = EngineClusterIter('toy', g, len(ds.par_li), batch, np.arange(g.number_of_nodes(), dtype=np.int64),
                           par_li=[p.copy() for p in ds.par_li], device=torch.device(DEV))
    return ds, it


@pytest.mark.parametrize('n_feats,with_drop', [(50, False), (50, True), (301, True), (64, False)])
def test_one_launch_extraction_equals_five_launches(hip, n_feats, with_drop):
    """gist_extract_parts_batch (membership from the part tables, count + scan + fill behind one grid
    barrier) == gist_extract_batch / gist_extract_batch_drop: row pointers, columns, norm, labels,
    features and the dropped features bit for bit, for every batch of two epochs."""
    from gist_amd import _lib
    L = _lib.load()
    ds, it = _toy_iter(9, 3000, 30, n_feats, 5)
    bt = it.batcher
    n_max = it.n_max
    assert L.gist_extract_parts_supported(n_max) == 1
    scratch = torch.zeros(int(L.gist_extract_parts_scratch_bytes(n_max)) // 8 + 1, dtype=torch.int64, device=DEV)
    i32 = dict(dtype=torch.int32, device=DEV)
    rp2, trp2 = torch.zeros(n_max + 1, **i32), torch.zeros(n_max + 1, **i32)
    cl2, tcl2 = torch.zeros(bt.col.numel(), **i32), torch.zeros(bt.col.numel(), **i32)
    norm2 = torch.zeros(n_max, device=DEV)
    lab2 = torch.zeros(n_max, **i32)
    g = bt.g
    ld = n_feats + (2 if n_feats % 4 else 0)
    for epoch in range(2):
        it.__iter__()
        node_part, tab = it._node_part, it._part_tables
        assert tab is not None
        for j in range(len(it)):
            a, b = int(it._offsets[j]), int(it._offsets[j + 1])
            ids = it._epoch_ids[a:b]
            n = b - a
            z_ref = torch.zeros(n, 2 * n_feats, device=DEV)
            z_new = torch.zeros(n, 2 * n_feats, device=DEV)
            x_ref = torch.zeros(n, ld, device=DEV)
            x_new = torch.zeros(n, ld, device=DEV)
            drop = (x_ref[:, :n_feats], 0.3, 11, 1000 * j, 2 * n_feats) if with_drop else None
            ref = bt.extract(ids, z_ref[:, :n_feats], drop=drop)
            fp, ldf = bt.feat.data_ptr(), bt.feat.stride(0)
            rc = L.gist_extract_parts_batch(
                g.rowptr.data_ptr(), g.col.data_ptr(), g.t_rowptr.data_ptr(), g.t_col.data_ptr(),
                ids.data_ptr(), n, n_max, node_part.data_ptr(), tab.data_ptr(), j, rp2.data_ptr(), cl2.data_ptr(), trp2.data_ptr(), tcl2.data_ptr(), cl2.numel(),
                norm2.data_ptr(), fp, ldf, n_feats, z_new.data_ptr(), 2 * n_feats, bt.labels.data_ptr(),
                lab2.data_ptr(), x_new.data_ptr() if with_drop else None, ld, 0.3, 11, 1000 * j, 2 * n_feats,
                scratch.data_ptr(), hip._stream())
            _lib.check(rc, 'gist_extract_parts_batch')
            nnz = int(ref.rowptr[n].item())
            assert torch.equal(rp2[:n + 1], ref.rowptr[:n + 1]) and torch.equal(trp2[:n + 1], ref.t_rowptr[:n + 1])
            assert nnz > 0 and torch.equal(cl2[:nnz], ref.col[:nnz])
            assert torch.equal(tcl2[:int(trp2[n].item())], ref.t_col[:int(trp2[n].item())])
            assert torch.equal(norm2[:n], ref.norm[:n]) and torch.equal(lab2[:n], ref.labels[:n])
            assert torch.equal(z_new, z_ref) and torch.equal(x_new, x_ref)
        random.shuffle(it.par_li)        # what the end of an epoch does (sampler.py:92)
    assert int(scratch[1].item()) == 0           # no workgroup ever gave up at the barrier


def test_one_launch_extraction_large_batch(hip):
    """More than 128 workgroups per CSR (n > 2048 rows: the look-back's third and later slots per lane)
    and a hub row with more kept neighbours than the LDS stash holds."""
    from gist_amd import _lib
    L = _lib.load()
    from gist_amd import datasets
    from gist_amd.sampler import EngineClusterIter
    ds = datasets.make_block_dataset('hubs', 6000, 12, 16, 6, intra_deg=8, inter_deg=2, seed=3, hub_frac=0.002,
                                     hub_mult=60)
    random.seed(4)
    it = EngineClusterIter('hubs', ds.g, len(ds.par_li), 6, np.arange(6000, dtype=np.int64),
                           par_li=[p.copy() for p in ds.par_li], device=torch.device(DEV))
    bt = it.batcher
    n_max = it.n_max
    assert n_max > 2500 and L.gist_extract_parts_supported(n_max) == 1
    scratch = torch.zeros(int(L.gist_extract_parts_scratch_bytes(n_max)) // 8 + 1, dtype=torch.int64, device=DEV)
    i32 = dict(dtype=torch.int32, device=DEV)
    rp2, trp2 = torch.zeros(n_max + 1, **i32), torch.zeros(n_max + 1, **i32)
    cl2, tcl2 = torch.zeros(bt.col.numel(), **i32), torch.zeros(bt.col.numel(), **i32)
    norm2, lab2 = torch.zeros(n_max, device=DEV), torch.zeros(n_max, **i32)
    g = bt.g
    it.__iter__()
    for j in range(len(it)):
        a, b = int(it._offsets[j]), int(it._offsets[j + 1])
        ids, n = it._epoch_ids[a:b], b - a
        z_ref, z_new = torch.zeros(n, 32, device=DEV), torch.zeros(n, 32, device=DEV)
        ref = bt.extract(ids, z_ref[:, :16])
        _lib.check(L.gist_extract_parts_batch(
            g.rowptr.data_ptr(), g.col.data_ptr(), g.t_rowptr.data_ptr(), g.t_col.data_ptr(), ids.data_ptr(), n,
            n_max, it._node_part.data_ptr(), it._part_tables.data_ptr(), j, rp2.data_ptr(), cl2.data_ptr(),
            trp2.data_ptr(), tcl2.data_ptr(), cl2.numel(), norm2.data_ptr(), bt.feat.data_ptr(), bt.feat.stride(0),
            16, z_new.data_ptr(), 32, bt.labels.data_ptr(), lab2.data_ptr(), None, 0, 0.0, 0, 0, 0,
            scratch.data_ptr(), hip._stream()), 'gist_extract_parts_batch')
        nnz = int(ref.rowptr[n].item())
        assert int((ref.rowptr[1:n + 1] - ref.rowptr[:n]).max().item()) > 256      # a row beyond the stash
        assert torch.equal(rp2[:n + 1], ref.rowptr[:n + 1]) and torch.equal(cl2[:nnz], ref.col[:nnz])
        assert torch.equal(trp2[:n + 1], ref.t_rowptr[:n + 1])
        assert torch.equal(tcl2[:int(trp2[n].item())], ref.t_col[:int(trp2[n].item())])
        assert torch.equal(z_new, z_ref) and torch.equal(norm2[:n], ref.norm[:n])
    assert int(scratch[1].item()) == 0


@pytest.mark.parametrize('p_drop', [0.0, 0.2])
def test_native_step_fused_equals_unfused(hip, monkeypatch, p_drop):
    """gist_sage_step with the fused sequence against the un-fused one (GIST_STEP_FUSE=0), GEMM mode f32,
    4 steps: the dropout folds, slab deferrals and the one-launch extraction are exact rewrites; the bias
    gradients and the k slices of the deferred projections change their summation order."""
    from gist_amd.engine import SageEngine, dims_for
    prev = hip.gemm_mode()
    hip.gemm_mode('f32')
    try:
        res = []
        for fuse in ('0', '1'):
            monkeypatch.setenv('GIST_STEP_FUSE', fuse)
            ds, it = _toy_iter(9, 3000, 30, 302, 5)
            dims = dims_for(302, 512, 6, 2)
            eng = SageEngine(dims, True, p_drop, it.n_max, torch.device(DEV), seed=11)
            gen = torch.Generator().manual_seed(1)
            for k, (i, o) in enumerate(dims):
                eng.arena.W[k].copy_((torch.rand(o, 2 * i, generator=gen) - 0.5) * 0.3)
                eng.arena.b[k].copy_((torch.rand(o, generator=gen) - 0.5) * 0.3)
            it.bind(eng)
            assert bool(eng.plan.fuse) == (fuse == '1')
            losses = []
            for j, b in enumerate(it):
                losses.append(eng.train_step(b, 0.01, 5e-4).clone())
                if j == 3:
                    break
            eng.check_extract()
            assert (eng._extract_scratch is not None) == (fuse == '1')      # the one-launch extraction ran
            res.append((eng.arena.params.clone(), torch.stack(losses), eng.arena.grads.clone()))
        # (the hidden layers' forward projections may run with another k-slice count when their slabs are left
        # to the LayerNorm: fp32 sums in another order -- the first loss agrees to rounding, later ones at 1e-4)
        dl = (res[0][1] - res[1][1]).abs()
        assert dl[0].item() < 2e-6 * max(1.0, res[0][1][0].abs().item()) and dl.max().item() < 1e-4
        assert (res[0][2] - res[1][2]).abs().max().item() < 1e-4 * max(1.0, res[0][2].abs().max().item())
        # Adam normalises by sqrt(v): a rounding-level difference of a tiny bias gradient may move its
        # parameter by a fraction of lr in the first steps; everything else must agree closely
        d = (res[0][0] - res[1][0]).abs()
        assert float((d > 1e-5).float().mean().item()) < 1e-3 and d.max().item() < 0.02
    finally:
        hip.gemm_mode(prev)


def test_blocked_aggregation_only_with_locality(hip):
    """The sampler measures whether the parts are locality blocks (share of edges inside a part, expected
    outside neighbours per batch row).  Reddit-like parts: yes -- batches carry their row blocks and the wide
    aggregations run on the blocked kernels.  The SAME graph with its nodes dealt to parts at random: no --
    batches carry no row blocks, every aggregation of the step runs on the row-split kernel, and the step
    still matches the oracle (the guard the round-2 review asked for: the matrix-core kernel is 1.6x slower
    than row-split on a batch without locality)."""
    from gist_amd import datasets
    from gist_amd.engine import SageEngine, dims_for
    from gist_amd.sampler import EngineClusterIter
    from oracle import train_oracle as TO
    ds = datasets.make_block_dataset('loc', 6000, 60, 64, 5, intra_deg=12, inter_deg=3, seed=2)
    g = ds.g
    nid = np.arange(6000, dtype=np.int64)
    rs = np.random.RandomState(0)
    random_parts = np.array_split(rs.permutation(6000).astype(np.int64), 60)
    for par_li, expect in ((ds.par_li, True), (random_parts, False)):
        random.seed(3)
        it = EngineClusterIter('loc', g, 60, 6, nid, par_li=[p.copy() for p in par_li], device=torch.device(DEV))
        assert it.locality is expect, it.locality_stats
        dims = dims_for(64, 2048, 5, 1)
        eng = SageEngine(dims, True, 0.0, it.n_max, torch.device(DEV))
        gen = torch.Generator().manual_seed(1)
        params = [(((torch.rand(o, 2 * i, generator=gen) - 0.5) * (2.0 / np.sqrt(2 * i))).numpy(),
                   ((torch.rand(o, generator=gen) - 0.5) * 0.1).numpy()) for (i, o) in dims]
        eng.arena.load(params)
        it.bind(eng)
        batch = next(iter(it))
        assert (batch.row_blocks is not None) == expect
        loss = eng.train_step(batch, 0.01)
        assert (eng.plan.n_row_blocks > 0) == expect            # what the step's aggregations were given
        tg = TO.TrainGraph(g.rowptr.numpy().astype(np.int64), g.col.numpy().astype(np.int64),
                           g.ndata['feat'].numpy(), g.ndata['label'].numpy().astype(np.int64))
        b = tg.batch(it.batch_ids(0))
        opt = O.new_opt_state(params)
        ref, _, _ = O.train_step(b[0], b[1], b[2], b[3], b[4], b[5], params, opt, True, 0.01)
        assert abs(float(loss.item()) - float(ref)) < 1e-4


@pytest.mark.parametrize('m,n,k', [(2046, 1024, 512), (1140, 1024, 512), (2046, 512, 256), (300, 192, 96), (2046, 2048, 1024)])
def test_gemm_nn_tn_dual_equals_the_two_launches(hip, m, n, k):
    """gist_gemm_nn_tn_dual_f32 (dz = dy . w and dW = dy^T . z as slabs, ONE launch of the fp32 kernel's tiles) against
    gist_gemm_nn_f32 and gist_gemm_slabs_f32 on the fp32 kernel: bit for bit, slab by slab; shapes the call does not take
    are refused."""
    from gist_amd import _lib
    L = _lib.load()
    gen = torch.Generator(device=DEV).manual_seed(m + n)
    dy = torch.randn(m, k, device=DEV, generator=gen)
    w = torch.randn(k, n, device=DEV, generator=gen)
    z = torch.randn(m, n, device=DEV, generator=gen)
    dz, dz_ref = torch.zeros(m, n, device=DEV), torch.zeros(m, n, device=DEV)
    dw, dw_ref = torch.zeros(k, n, device=DEV), torch.zeros(k, n, device=DEV)
    nb = max(int(L.gist_gemm_workspace_bytes(k, n, m)), 32 * k * n * 4)
    slabs, slabs_ref = torch.zeros(nb // 4, device=DEV), torch.zeros(nb // 4, device=DEV)
    takes = hip.gemm_dual_takes(dy, w, z, dz)
    assert takes == ((m, n, k) != (2046, 2048, 1024))          # (1536 workgroups: each product fills the chip alone)
    if not takes:
        with pytest.raises(Exception):
            hip.gemm_nn_tn_dual(dy, w, dz, z, dw, slabs)
        return
    prev = hip.gemm_mode()
    hip.gemm_mode('f32')
    try:
        ns = hip.gemm_nn_tn_dual(dy, w, dz, z, dw, slabs)
        hip.gemm_nn(dy, w, dz_ref)
        ns_ref = hip.gemm_slabs('tn', dy, z, None, dw_ref, slabs_ref)
    finally:
        hip.gemm_mode(prev)
    assert torch.equal(dz, dz_ref)
    assert ns == ns_ref
    if ns > 1:
        assert torch.equal(slabs[:ns * k * n], slabs_ref[:ns * k * n])
    else:
        assert torch.equal(dw, dw_ref)
    ref = dy.double().t() @ z.double()
    got = slabs[:ns * k * n].view(ns, k, n).double().sum(0) if ns > 1 else dw.double()
    assert (got - ref).abs().max().item() < 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize('n,d,use_ln,n_blocks', [(700, 256, True, 7), (2046, 256, True, 20), (500, 128, True, 5), (900, 192, False, 9),
                                                 (600, 256, True, 4)])      # (blocks of 150 rows: beyond the staged tile)
def test_reverse_aggregation_with_the_layernorm_backward_in_its_store(hip, n, d, use_ln, n_blocks):
    """gist_spmm_csr_drop_lnbwd_f32 == the mode-2 reverse aggregation followed by gist_ln_relu_bwd_colsum_f32: dy and the bias
    gradient (sum of the partial rows) to fp32 rounding (the same formulas; the compiler contracts them into FMAs its own way in
    each kernel), bit for bit without LayerNorm."""
    rp, cl, trp, tcl, rb = block_graph(n, n_blocks, 10, 1, seed=n + d, hub=300)
    gen = torch.Generator(device=DEV).manual_seed(12)
    dz = torch.randn(n, 2 * d, device=DEV, generator=gen)
    yhat = torch.randn(n, d, device=DEV, generator=gen)
    rstd = torch.rand(n, device=DEV, generator=gen) + 0.5
    norm = hip.in_degree_norm(rp)
    p, seed, off = 0.25, 5, 64
    ref = dz.clone()
    hip.spmm_drop(trp, tcl, ref[:, d:], ref[:, :d], 2, p, seed, off, off + d, 2 * d, src_scale=norm, accumulate=True, row_blocks=rb)
    dy_ref = torch.zeros(n, d, device=DEV)
    chunks = torch.zeros((n + 15) // 16, d, device=DEV)
    hip.ln_relu_bwd_colsum(ref[:, :d], yhat, rstd if use_ln else None, dy_ref, use_ln, True, chunks)
    new = dz.clone()
    units = hip.spmm_lnb_units(n_blocks)
    assert units >= n_blocks
    parts = torch.full((units, d), float('nan'), device=DEV)
    y_io = yhat.clone()                                   # in place, as the step runs it
    hip.spmm_drop_lnbwd(trp, tcl, new[:, d:], new[:, :d], p, seed, off, off + d, 2 * d, y_io, y_io, parts, src_scale=norm,
                        row_blocks=rb, rstd=rstd if use_ln else None)
    if use_ln:
        err_dy = (y_io - dy_ref).abs().max().item()
        assert err_dy <= 2e-6 * max(1.0, dy_ref.abs().max().item()), err_dy
    else:
        assert torch.equal(y_io, dy_ref)
    assert torch.equal(new, dz)                           # (y is only read)
    db_ref = dy_ref.double().sum(0)
    err = (parts.double().sum(0) - db_ref).abs().max().item()
    assert err <= 1e-5 * max(1.0, db_ref.abs().max().item()), err
