"""GPU parity: every HIP kernel, called through the C ABI, against the CPU oracle on the
same seeded inputs.  fp32 values: 1e-4 (north_star tolerance, relative to the
magnitude of the result for long reductions); integer / index work: bit exact."""
import numpy as np
import pytest
import torch

from oracle import gist_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = 'cuda:0'


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def rand_graph(n, avg_deg, seed, hub=0):
    rs = np.random.RandomState(seed)
    m = n * avg_deg
    src = rs.randint(0, n, m)
    dst = rs.randint(0, max(n - 1, 1), m)      # last node has zero in-degree
    if hub:
        src = np.concatenate([src, rs.randint(0, n, hub)])
        dst = np.concatenate([dst, np.full(hub, 1)])
    return O.csr_from_edges(src, dst, n)


def close(a, b, tol=TOL):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, 'max err %g (scale %g)' % (err, scale)


@pytest.fixture(scope='module')
def hip():
    from gist_amd import hip as h
    assert h.device_count() >= 1
    return h


# ------------------------------------------------------------------ SpMM
@pytest.mark.parametrize('n,d,deg,hub', [
    (7, 5, 3, 0), (64, 41, 4, 0), (257, 100, 6, 300), (300, 602, 8, 0),
    (513, 256, 16, 700), (1000, 1024, 20, 0), (2046, 4096, 64, 3000), (129, 2, 5, 0),
    (90, 12, 7, 0), (200, 3, 2, 0)])
def test_spmm_forward(hip, n, d, deg, hub):
    rowptr, col = rand_graph(n, deg, seed=n + d, hub=hub)
    rs = np.random.RandomState(1)
    x = rs.randn(n, d).astype(np.float32)
    norm = O.in_degree_norm(rowptr)
    ref = O.spmm_sum(rowptr, col, x, out_scale=norm)
    rp, cl = dev(rowptr, torch.int32), dev(col, torch.int32)
    nd = hip.in_degree_norm(rp)
    assert np.array_equal(nd.cpu().numpy(), norm)
    y = torch.full((n, d), 7.0, device=DEV)
    hip.spmm(rp, cl, dev(x), y, out_scale=nd)
    close(y, ref)
    assert (y[np.diff(rowptr) == 0] == 0).all()       # zero in-degree rows -> exactly 0


@pytest.mark.parametrize('n,d', [(64, 10), (300, 602), (700, 256), (1500, 1024), (130, 258), (90, 1026)])
def test_spmm_concat_window_and_backward_form(hip, n, d):
    """x/y as the two halves of one [n, 2d] buffer; backward = src_scale + accumulate."""
    rowptr, col = rand_graph(n, 9, seed=d)
    t_rowptr, t_col = O.transpose_csr(rowptr, col)
    rs = np.random.RandomState(2)
    z = rs.randn(n, 2 * d).astype(np.float32)
    norm = O.in_degree_norm(rowptr)
    zt = dev(z)
    hip.spmm(dev(rowptr, torch.int32), dev(col, torch.int32), zt[:, :d], zt[:, d:],
             out_scale=dev(norm))
    ref = z.copy()
    ref[:, d:] = O.spmm_sum(rowptr, col, np.ascontiguousarray(z[:, :d]), out_scale=norm)
    close(zt, ref)
    # backward: dh = dz[:, :d] + A^T (norm * dz[:, d:])
    gz = rs.randn(n, 2 * d).astype(np.float32)
    gt = dev(gz)
    hip.spmm(dev(t_rowptr, torch.int32), dev(t_col, torch.int32), gt[:, d:], gt[:, :d],
             src_scale=dev(norm), accumulate=True)
    dh = np.ascontiguousarray(gz[:, :d])
    O.spmm_sum(t_rowptr, t_col, gz[:, d:], src_scale=norm, out=dh, accumulate=True)
    close(gt[:, :d], dh)
    assert np.array_equal(gt[:, d:].cpu().numpy(), gz[:, d:])


def test_spmm_empty(hip):
    rp = torch.zeros(1, dtype=torch.int32, device=DEV)
    cl = torch.zeros(0, dtype=torch.int32, device=DEV)
    hip.spmm(rp, cl, torch.zeros(0, 8, device=DEV), torch.zeros(0, 8, device=DEV))


# ------------------------------------------------------------------ GEMM
GEMM_SHAPES = [(1, 1, 1), (5, 7, 3), (128, 128, 32), (130, 129, 33), (257, 41, 1204),
               (300, 256, 1204), (2046, 41, 1024), (64, 512, 200), (500, 1024, 2048),
               (41, 300, 2046), (2046, 1024, 512)]


@pytest.mark.parametrize('m,n,k', GEMM_SHAPES)
def test_gemm_nt(hip, m, n, k):
    rs = np.random.RandomState(m + n + k)
    a = rs.randn(m, k).astype(np.float32)
    w = rs.randn(n, k).astype(np.float32)
    b = rs.randn(n).astype(np.float32)
    y = torch.full((m, n), np.nan, device=DEV)
    hip.gemm_nt(dev(a), dev(w), dev(b), y)
    ref = (a.astype(np.float64) @ w.T.astype(np.float64) + b).astype(np.float32)
    close(y, ref, tol=2e-6 * np.sqrt(k) + 1e-6)
    y2 = torch.full((m, n), np.nan, device=DEV)
    hip.gemm_nt(dev(a), dev(w), None, y2)
    close(y2, ref - b, tol=2e-6 * np.sqrt(k) + 1e-6)


@pytest.mark.parametrize('m,n,k', GEMM_SHAPES)
def test_gemm_nn(hip, m, n, k):
    rs = np.random.RandomState(m + 2 * n + k)
    g = rs.randn(m, k).astype(np.float32)
    w = rs.randn(k, n).astype(np.float32)
    z = torch.full((m, n), np.nan, device=DEV)
    hip.gemm_nn(dev(g), dev(w), z)
    close(z, (g.astype(np.float64) @ w.astype(np.float64)).astype(np.float32),
          tol=2e-6 * np.sqrt(k) + 1e-6)


@pytest.mark.parametrize('m,n,k', GEMM_SHAPES)
def test_gemm_tn(hip, m, n, k):
    rs = np.random.RandomState(m + n + 3 * k)
    g = rs.randn(k, m).astype(np.float32)
    a = rs.randn(k, n).astype(np.float32)
    d = torch.full((m, n), np.nan, device=DEV)
    hip.gemm_tn(dev(g), dev(a), d)
    close(d, (g.T.astype(np.float64) @ a.astype(np.float64)).astype(np.float32),
          tol=2e-6 * np.sqrt(k) + 1e-6)


def test_gemm_strided_unaligned_operands(hip):
    """Operands that are column windows with odd leading dimensions (scalar-load path),
    asymmetric data so a transposed store cannot pass."""
    rs = np.random.RandomState(9)
    m, n, k = 70, 45, 37
    abuf = rs.randn(m, k + 5).astype(np.float32)
    wbuf = rs.randn(n, k + 3).astype(np.float32)
    ybuf = torch.zeros(m, n + 7, device=DEV)
    at, wt = dev(abuf), dev(wbuf)
    hip.gemm_nt(at[:, 1:1 + k], wt[:, 2:2 + k], None, ybuf[:, 3:3 + n])
    ref = abuf[:, 1:1 + k].astype(np.float64) @ wbuf[:, 2:2 + k].T.astype(np.float64)
    close(ybuf[:, 3:3 + n], ref.astype(np.float32), tol=2e-5)
    assert (ybuf[:, :3] == 0).all() and (ybuf[:, 3 + n:] == 0).all()


def test_gemm_output_window_untouched_around_ragged_tiles(hip):
    """The store epilogue relies on the hardware range check for ragged columns and on a
    vector-offset path for ragged rows: nothing outside the [m, n] window may be written,
    for all three forms, with m and n that are not tile multiples."""
    rs = np.random.RandomState(4)
    m, n, k = 130, 70, 96
    guard = 5
    for form in ('nt', 'nn', 'tn'):
        ybuf = torch.full((m + guard, n + 6), 123.0, device=DEV)
        y = ybuf[:m, 2:2 + n]
        if form == 'nt':
            a, w = rs.randn(m, k).astype(np.float32), rs.randn(n, k).astype(np.float32)
            hip.gemm_nt(dev(a), dev(w), None, y)
            ref = a.astype(np.float64) @ w.T.astype(np.float64)
        elif form == 'nn':
            a, w = rs.randn(m, k).astype(np.float32), rs.randn(k, n).astype(np.float32)
            hip.gemm_nn(dev(a), dev(w), y)
            ref = a.astype(np.float64) @ w.astype(np.float64)
        else:
            a, w = rs.randn(k, m).astype(np.float32), rs.randn(k, n).astype(np.float32)
            hip.gemm_tn(dev(a), dev(w), y)
            ref = a.T.astype(np.float64) @ w.astype(np.float64)
        close(y, ref.astype(np.float32), tol=2e-5)
        assert (ybuf[m:] == 123.0).all(), form
        assert (ybuf[:, :2] == 123.0).all() and (ybuf[:, 2 + n:] == 123.0).all(), form


def test_gemm_rows_beyond_4gib(hip):
    """Tile-relative 32-bit offsets: an operand of 4.9 GB (300 000 rows x 4096), whose last
    rows lie beyond 4 GiB from its base, as in full-graph evaluation."""
    m, k, n = 300000, 4096, 64
    gen = torch.Generator(device=DEV).manual_seed(1)
    a = torch.randn(m, k, device=DEV, generator=gen)
    w = torch.randn(n, k, device=DEV, generator=gen)
    y = torch.empty(m, n, device=DEV)
    hip.gemm_nt(a, w, None, y)
    for lo in (0, 150000, m - 300):
        ref = (a[lo:lo + 300].double() @ w.double().t()).float()
        assert (y[lo:lo + 300] - ref).abs().max().item() < 2e-3


def test_gemm_rejects_huge_leading_dimension(hip):
    from gist_amd import _lib
    L = _lib.load()
    a = torch.zeros(4, 4, device=DEV)
    rc = L.gist_gemm_nt_f32(a.data_ptr(), 1 << 22, a.data_ptr(), 4, None, a.data_ptr(), 4, 4, 4, 4,
                            None, 0, None)
    assert rc < 0 and b'leading dimension' in L.gist_last_error()


def test_gemm_random_shapes_all_forms(hip):
    """60 random (m, n, k) per form, including windows with odd offsets / leading dimensions,
    against a float64 product: exercises full and ragged tiles, the DMA and the register
    staging paths, both tile sizes and split-K as the chooser picks them."""
    rs = np.random.RandomState(123)
    for it in range(60):
        m = int(rs.choice([1, 7, 33, 64, 100, 128, 129, 200, 257, 511, 700, 1030]))
        n = int(rs.choice([1, 5, 41, 64, 96, 128, 130, 300, 513, 1024]))
        k = int(rs.choice([1, 4, 31, 32, 33, 64, 65, 127, 128, 200, 513, 1204, 2050]))
        aligned = it % 3 != 0
        pad = 0 if aligned else int(rs.choice([1, 2, 3]))
        for form in ('nt', 'nn', 'tn'):
            if form == 'nt':
                A, W = rs.randn(m, k + pad), rs.randn(n, k + pad)
                a, w = dev(A.astype(np.float32))[:, pad:], dev(W.astype(np.float32))[:, pad:]
                ref = A[:, pad:].astype(np.float32).astype(np.float64) @ W[:, pad:].astype(np.float32).astype(np.float64).T
            elif form == 'nn':
                A, W = rs.randn(m, k + pad), rs.randn(k, n + pad)
                a, w = dev(A.astype(np.float32))[:, pad:], dev(W.astype(np.float32))[:, pad:]
                ref = A[:, pad:].astype(np.float32).astype(np.float64) @ W[:, pad:].astype(np.float32).astype(np.float64)
            else:
                A, W = rs.randn(k, m + pad), rs.randn(k, n + pad)
                a, w = dev(A.astype(np.float32))[:, pad:], dev(W.astype(np.float32))[:, pad:]
                ref = A[:, pad:].astype(np.float32).astype(np.float64).T @ W[:, pad:].astype(np.float32).astype(np.float64)
            ybuf = torch.full((m, n + pad + 1), 9.0, device=DEV)
            y = ybuf[:, pad:pad + n]
            bias = None
            if form == 'nt' and it % 2 == 0:
                b = rs.randn(n).astype(np.float32)
                bias = dev(b)
                ref = ref + b.astype(np.float64)
            if form == 'nt':
                hip.gemm_nt(a, w, bias, y)
            elif form == 'nn':
                hip.gemm_nn(a, w, y)
            else:
                hip.gemm_tn(a, w, y)
            err = np.abs(y.cpu().numpy().astype(np.float64) - ref).max()
            assert err <= 3e-6 * np.sqrt(k) * max(1.0, np.abs(ref).max()) + 1e-6, (form, m, n, k, pad, err)
            assert (ybuf[:, pad + n:] == 9.0).all() and (ybuf[:, :pad] == 9.0).all(), (form, m, n, k, pad)


def test_gemm_identity_asymmetric(hip):
    """A = I against an asymmetric B catches a swapped C/D register map."""
    n = 96
    b = np.arange(n * n, dtype=np.float32).reshape(n, n) / 100.0
    out = torch.zeros(n, n, device=DEV)
    hip.gemm_nn(dev(np.eye(n, dtype=np.float32)), dev(b), out)
    assert np.array_equal(out.cpu().numpy(), b)


@pytest.mark.parametrize('m,n,k,p,offset', [
    (2046, 8192, 41, 0.2, 0), (300, 1024, 6, 0.5, 123456), (513, 2052, 47, 0.0, 0),
    (17, 1028, 64, 0.3, 2), (100, 1024, 41, 0.2, 7),      # odd offset: two-kernel path
    (100, 1030, 41, 0.2, 0),                              # n % 4 != 0: two-kernel path
    (64, 512, 65, 0.2, 0)])                               # k > 64: two-kernel path
def test_gemm_nn_dropout_fused(hip, m, n, k, p, offset):
    """z = dropout(g @ w) in one call == gist_gemm_nn_f32 followed by gist_dropout_f32: the
    same kept/dropped pattern bit for bit, values equal up to the order of a <= 64-term sum."""
    rs = np.random.RandomState(m + n + k)
    g = dev(rs.randn(m, k).astype(np.float32))
    w = dev(rs.randn(k, n).astype(np.float32))
    zbuf = torch.full((m + 2, n + 4), 5.0, device=DEV)
    z = zbuf[:m, :n]
    hip.gemm_nn_dropout_(g, w, z, p, 99, offset)
    ref = torch.empty(m, n, device=DEV)
    hip.gemm_nn(g, w, ref)
    if p > 0:
        hip.dropout_(ref, p, 99, offset)
    assert torch.equal(z == 0, ref == 0)
    scale = max(1.0, ref.abs().max().item())
    assert (z - ref).abs().max().item() <= 2e-6 * scale * np.sqrt(k)
    assert (zbuf[m:] == 5.0).all() and (zbuf[:, n:] == 5.0).all()
    if p > 0:
        frac = (z == 0).float().mean().item()
        assert abs(frac - p) < 0.02


# ------------------------------------------------------------------ LN / ReLU / dropout / colsum
@pytest.mark.parametrize('n,d', [(5, 7), (64, 41), (300, 256), (257, 1024), (100, 4096), (33, 1030)])
@pytest.mark.parametrize('ln,relu', [(True, True), (True, False), (False, True)])
def test_ln_relu_fwd_bwd(hip, n, d, ln, relu):
    rs = np.random.RandomState(n + d)
    y = (rs.randn(n, d) * 3 + 0.5).astype(np.float32)
    g = rs.randn(n, d).astype(np.float32)
    yt = dev(y)
    nxt = torch.zeros(n, 2 * d, device=DEV)           # write into a left half
    rstd = torch.zeros(n, device=DEV)
    hip.ln_relu_fwd(yt, nxt[:, :d], rstd, ln, relu)
    if ln:
        mu = y.mean(1, keepdims=True, dtype=np.float32)
        var = ((y - mu) ** 2).mean(1, keepdims=True, dtype=np.float32)
        rs_ref = 1.0 / np.sqrt(var + 1e-5)
        yhat = (y - mu) * rs_ref
        close(rstd, rs_ref[:, 0], tol=1e-5)
    else:
        yhat = y
    out = np.maximum(yhat, 0) if relu else yhat
    close(yt, yhat)
    close(nxt[:, :d], out)
    assert (nxt[:, d:] == 0).all()
    # backward (oracle formula)
    gg = g * (yhat > 0) if relu else g
    if ln:
        m1 = gg.mean(1, keepdims=True, dtype=np.float32)
        m2 = (gg * yhat).mean(1, keepdims=True, dtype=np.float32)
        gg = rs_ref * (gg - m1 - yhat * m2)
    dy = torch.empty(n, d, device=DEV)
    hip.ln_relu_bwd(dev(g), yt, rstd, dy, ln, relu)
    close(dy, gg.astype(np.float32))
    hip.ln_relu_bwd(dev(g), yt, rstd, yt, ln, relu)     # in place over yhat
    close(yt, gg.astype(np.float32))


def _dropout_mask_ref(n, d, p, seed, offset):
    idx = np.arange(n * d, dtype=np.uint64) + np.uint64(offset)
    with np.errstate(over='ignore'):
        z = (idx >> np.uint64(1)) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    w = np.where(idx & np.uint64(1), z >> np.uint64(32), z & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    u = (w >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u >= np.float32(p)).reshape(n, d)


@pytest.mark.parametrize('n,d,p', [(50, 33, 0.2), (300, 1204, 0.5), (2046, 512, 0.2)])
def test_dropout(hip, n, d, p):
    rs = np.random.RandomState(3)
    z = rs.randn(n, d + 3).astype(np.float32)
    zt = dev(z)
    hip.dropout_(zt[:, :d], p, seed=1234, offset=77)
    mask = _dropout_mask_ref(n, d, p, 1234, 77)
    ref = z.copy()
    ref[:, :d] = z[:, :d] * mask * np.float32(1.0 / (1.0 - p))
    close(zt, ref, tol=1e-6)
    keep = mask.mean()
    assert abs(keep - (1 - p)) < 0.02
    # a different offset gives a different mask; same call regenerates the same one
    z2 = dev(z)
    hip.dropout_(z2[:, :d], p, seed=1234, offset=77)
    assert torch.equal(z2, zt)


@pytest.mark.parametrize('n,d', [(1, 1), (127, 41), (300, 1024), (2046, 4096)])
def test_colsum(hip, n, d):
    rs = np.random.RandomState(n)
    g = rs.randn(n, d).astype(np.float32)
    out = torch.zeros(d, device=DEV)
    hip.colsum(dev(g), out)
    close(out, g.sum(0, dtype=np.float64).astype(np.float32), tol=1e-5)


# ------------------------------------------------------------------ loss / Adam / accuracy
@pytest.mark.parametrize('n,c,ldg', [(10, 5, 8), (300, 41, 44), (2046, 47, 48), (70, 130, 132)])
def test_softmax_xent(hip, n, c, ldg):
    rs = np.random.RandomState(c)
    logits = (rs.randn(n, c) * 4).astype(np.float32)
    labels = rs.randint(0, c, n)
    mask = rs.rand(n) < 0.8
    for msk in (None, mask):
        loss_ref, d_ref = O.cross_entropy(logits, labels, msk)
        cnt = n if msk is None else int(msk.sum())
        dl = torch.full((n, ldg), np.nan, device=DEV)
        loss = torch.zeros(1, device=DEV)
        hip.softmax_xent(dev(logits), dev(labels, torch.int32),
                         None if msk is None else dev(msk.astype(np.uint8)), cnt,
                         torch.empty(n, device=DEV), loss, dl)
        assert abs(loss.item() - loss_ref) < TOL
        close(dl[:, :c], d_ref, tol=1e-6)
        assert (dl[:, c:] == 0).all()


@pytest.mark.parametrize('wd', [0.0, 5e-4])
def test_adam(hip, wd):
    rs = np.random.RandomState(4)
    n = 10007
    p = rs.randn(n).astype(np.float32)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    pt, mt, vt = dev(p), dev(m), dev(v)
    for step in range(1, 6):
        g = rs.randn(n).astype(np.float32)
        O.adam_step(p, g, m, v, step, 0.01, weight_decay=wd)
        hip.adam_(pt, dev(g), mt, vt, step, 0.01, weight_decay=wd)
    close(pt, p, tol=1e-6)
    close(mt, m, tol=1e-6)
    close(vt, v, tol=1e-6)


def test_argmax_correct(hip):
    rs = np.random.RandomState(5)
    n, c = 1000, 41
    logits = rs.randn(n, c).astype(np.float32)
    logits[::7, 3] = logits[::7].max(1)          # ties: first max wins like numpy
    labels = rs.randint(0, c, n)
    labels[:200] = np.argmax(logits[:200], 1)
    mask = rs.rand(n) < 0.6
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.argmax_correct(dev(logits), dev(labels, torch.int32), dev(mask.astype(np.uint8)), cnt)
    assert cnt.item() == int((np.argmax(logits[mask], 1) == labels[mask]).sum())


# ------------------------------------------------------------------ cluster batch extraction
@pytest.mark.parametrize('n,deg,nb,hub', [(50, 4, 20, 0), (3000, 30, 700, 500), (20000, 60, 2046, 4000)])
def test_induced_subgraph(hip, n, deg, nb, hub):
    rowptr, col = rand_graph(n, deg, seed=n, hub=hub)
    rs = np.random.RandomState(6)
    ids = rs.permutation(n)[:nb].astype(np.int64)
    if hub:
        ids[0] = 1                                  # include the hub row
        ids = np.unique(ids)
        rs.shuffle(ids)
        nb = ids.shape[0]
    ref_rp, ref_cl = O.induced_subgraph(rowptr, col, ids)
    rp, cl, idt = dev(rowptr, torch.int32), dev(col, torch.int32), dev(ids, torch.int32)
    remap = torch.empty(n, dtype=torch.int32, device=DEV)
    hip.fill_i32_(remap, -1)
    hip.induced_mark(idt, remap)
    srp = torch.empty(nb + 1, dtype=torch.int32, device=DEV)
    hip.induced_rowptr(rp, cl, idt, remap, srp)
    assert np.array_equal(srp.cpu().numpy(), ref_rp)
    scl = torch.full((int(ref_rp[-1]) + 5,), -7, dtype=torch.int32, device=DEV)
    hip.induced_fill(rp, cl, idt, remap, srp, scl)
    assert np.array_equal(scl.cpu().numpy()[:-5], ref_cl)
    assert (scl[-5:] == -7).all()
    hip.induced_mark(idt, remap, unmark=True)
    assert (remap == -1).all()
    # feature / label gather
    feat = rs.randn(n, 602).astype(np.float32)
    z0 = torch.zeros(nb, 1204, device=DEV)
    hip.gather_rows(dev(feat), idt, z0[:, :602])
    assert np.array_equal(z0[:, :602].cpu().numpy(), feat[ids])
    lab = rs.randint(0, 41, n).astype(np.int32)
    lo = torch.empty(nb, dtype=torch.int32, device=DEV)
    hip.gather_i32(dev(lab), idt, lo)
    assert np.array_equal(lo.cpu().numpy(), lab[ids])


def test_scan_large(hip):
    """train-induced subgraph size: > 1024 rows exercises the multi-chunk scan."""
    n = 150000
    rowptr, col = rand_graph(n, 3, seed=8)
    ids = np.arange(0, n, 2, dtype=np.int64)
    ref_rp, _ = O.induced_subgraph(rowptr, col, ids)
    remap = torch.full((n,), -1, dtype=torch.int32, device=DEV)
    idt = dev(ids, torch.int32)
    hip.induced_mark(idt, remap)
    srp = torch.empty(ids.shape[0] + 1, dtype=torch.int32, device=DEV)
    hip.induced_rowptr(dev(rowptr, torch.int32), dev(col, torch.int32), idt, remap, srp)
    assert np.array_equal(srp.cpu().numpy(), ref_rp)


# ------------------------------------------------------------------ IST blocks
def test_block_gather_scatter(hip):
    rs = np.random.RandomState(7)
    H, h = 64, 16
    W = rs.randn(H, 2 * H).astype(np.float32)
    rows = rs.permutation(H)[:h]
    cols = np.concatenate([rs.permutation(H)[:h]] * 2)
    cols[h:] += H
    Wt = dev(W)
    blk = torch.empty(h, 2 * h, device=DEV)
    hip.block_gather(Wt, dev(rows, torch.int32), dev(cols, torch.int32), blk)
    assert np.array_equal(blk.cpu().numpy(), W[np.ix_(rows, cols)])
    blk2 = blk + 1
    hip.block_scatter(blk2, dev(rows, torch.int32), dev(cols, torch.int32), Wt)
    ref = W.copy()
    ref[np.ix_(rows, cols)] += 1
    assert np.array_equal(Wt.cpu().numpy(), ref)
    # identity index forms
    out = torch.empty(5, 2 * h, device=DEV)
    hip.block_gather(Wt, None, dev(cols, torch.int32), out)
    assert np.array_equal(out.cpu().numpy(), ref[:5][:, cols])
    src = torch.stack([torch.arange(41, dtype=torch.float32, device=DEV) * (s + 1) for s in range(4)])
    o = torch.empty(41, device=DEV)
    hip.mean_rows(src.reshape(-1), 41, 4, 41, o)
    close(o, (np.arange(41, dtype=np.float32) * 2.5), tol=1e-6)


@pytest.mark.parametrize('tile,splits', [(128, 1), (128, 4), (64, 1), (64, 2), (64, 8)])
def test_gemm_forced_configs(hip, tile, splits):
    """Every (block tile, split-K) configuration the chooser can pick is correct on ragged
    shapes for all three layouts (tuning hooks gemm_tile / gemm_splits)."""
    dev = torch.device(DEV)
    hip.workspace(64 << 20, dev)
    prev_mode = hip.gemm_mode()
    hip.gemm_mode('f32')
    hip.tuning('gemm_tile', tile)
    hip.tuning('gemm_splits', splits)
    try:
        rs = np.random.RandomState(0)
        worst = 0.0
        for (m, n, k) in [(130, 129, 300), (257, 41, 1204), (70, 520, 2046), (2046, 96, 200)]:
            a = rs.randn(m, k).astype(np.float32); w = rs.randn(n, k).astype(np.float32)
            b = rs.randn(n).astype(np.float32)
            y = torch.full((m, n), float('nan'), device=dev)
            hip.gemm_nt(torch.from_numpy(a).to(dev), torch.from_numpy(w).to(dev),
                        torch.from_numpy(b).to(dev), y)
            worst = max(worst, float(np.abs(y.cpu().numpy() - (a.astype(np.float64) @ w.T + b)).max()))
            g = rs.randn(m, k).astype(np.float32); w2 = rs.randn(k, n).astype(np.float32)
            z = torch.full((m, n), float('nan'), device=dev)
            hip.gemm_nn(torch.from_numpy(g).to(dev), torch.from_numpy(w2).to(dev), z)
            worst = max(worst, float(np.abs(z.cpu().numpy() - g.astype(np.float64) @ w2).max()))
            gt = rs.randn(k, m).astype(np.float32); at = rs.randn(k, n).astype(np.float32)
            d = torch.full((m, n), float('nan'), device=dev)
            hip.gemm_tn(torch.from_numpy(gt).to(dev), torch.from_numpy(at).to(dev), d)
            worst = max(worst, float(np.abs(d.cpu().numpy() - gt.T.astype(np.float64) @ at).max()))
    finally:
        hip.tuning('gemm_tile', 0)
        hip.tuning('gemm_splits', 0)
        hip.gemm_mode(prev_mode)
    assert worst < 5e-4, worst          # |sum of 2046 N(0,1) products| ~ 45, fp32 accumulation


@pytest.mark.parametrize('n,d,deg,hub', [(300, 602, 8, 0), (513, 256, 16, 700), (1000, 1024, 20, 0),
                                         (2046, 4096, 64, 3000), (700, 130, 5, 0), (90, 12, 7, 0),
                                         (777, 300, 12, 0), (260, 2052, 9, 400)])       # ragged last column tile
@pytest.mark.parametrize('blocks', ['uniform', 'parts', 'oversize'])
@pytest.mark.parametrize('kernel', ['lds_gather', 'mfma_block_dense'])
def test_spmm_blocked_lds(hip, n, d, deg, hub, blocks, kernel):
    """Both kernels behind gist_spmm_csr_blocked_f32 (the LDS gather kernel and the block-dense
    matrix-core kernel, forced through the tuning hook) == plain kernel semantics, for uniform
    128-row blocks, ragged 'METIS part' blocks and blocks larger than the 128 rows that fit in LDS;
    multigraph with strong in-block locality plus remote edges and a hub (a row with thousands of
    neighbours, which leaves the dense product and is gathered in full)."""
    hip.tuning('spmm_kernel', 1 if kernel == 'lds_gather' else 2)
    try:
        _spmm_blocked_case(hip, n, d, deg, hub, blocks)
    finally:
        hip.tuning('spmm_kernel', 0)


def test_spmm_block_dense_exact_counts_and_order():
    """The matrix-core kernel is exact fp32 aggregation: on integer-valued features (every partial sum
    exact in fp32, whatever the order) it must equal the oracle BIT FOR BIT -- including an edge
    listed 300 times (a count above 256 is not exact in bf16: that row leaves the dense product) and
    features with 21 significant bits (the three bf16 pieces must carry all of them)."""
    from gist_amd import hip
    rs = np.random.RandomState(5)
    n, d = 400, 2048
    src = rs.randint(0, n, 6000); dst = (src // 100) * 100 + rs.randint(0, 100, 6000)
    dst = np.minimum(dst, n - 1)
    src = np.concatenate([src, np.full(300, 7)]); dst = np.concatenate([dst, np.full(300, 9)])
    rowptr, col = O.csr_from_edges(src, dst, n)
    x = np.zeros((n, 2 * d), np.float32)
    # multiples of 2^-12 below 512 (21 significant bits: all three bf16 pieces are needed), sparse enough
    # that every sum stays below 2^12, i.e. exact in fp32 in any order; the 300-fold neighbour: small integers
    x[:, :d] = rs.randint(-(1 << 21), 1 << 21, (n, d)).astype(np.float32) / 4096.0
    x[:, :d] *= (rs.rand(n, d) < 0.05)
    x[7, :d] = rs.randint(-8, 9, d).astype(np.float32)
    rb = dev(np.array([0, 100, 200, 300, 400]), torch.int32)
    xt = dev(x)
    hip.tuning('spmm_kernel', 2)
    try:
        hip.spmm(dev(rowptr, torch.int32), dev(col, torch.int32), xt[:, :d], xt[:, d:], row_blocks=rb, blocked=True)
    finally:
        hip.tuning('spmm_kernel', 0)
    want = O.spmm_sum(rowptr, col, np.ascontiguousarray(x[:, :d]))
    ref64 = np.zeros((n, d))
    np.add.at(ref64, np.repeat(np.arange(n), np.diff(rowptr)), x[col, :d].astype(np.float64))
    assert np.array_equal(want.astype(np.float64), ref64), 'test data: the fp32 sums are not exact'
    assert np.array_equal(xt[:, d:].cpu().numpy(), want)


@pytest.mark.parametrize('n,d,deg,hub,blocks', [(2046, 4096, 64, 3000, 'parts'), (1000, 1024, 20, 0, 'uniform'),
                                                (260, 2052, 9, 400, 'parts'), (1500, 1536, 30, 0, 'uniform'),
                                                (513, 256, 16, 700, 'oversize'), (300, 602, 8, 0, 'parts')])
def test_spmm_prepared_blocks(hip, n, d, deg, hub, blocks):
    """gist_spmm_blocks_prepare + gist_spmm_csr_prepared_f32 == the oracle (forward and backward form),
    and bit-identical to the unprepared matrix-core kernel: the block structure built once per
    graph is the one every workgroup would build for itself."""
    _spmm_blocked_case(hip, n, d, deg, hub, blocks, prepared=True)


def test_spmm_blocks_prepare_rejects_bad_buffers(hip):
    """The prepared-block entry points fail loudly: a buffer smaller than gist_spmm_blocks_bytes, a
    misaligned one, a missing one."""
    from gist_amd import _lib
    L = _lib.load()
    rp = torch.arange(0, 301, dtype=torch.int32, device=DEV)
    cl = torch.zeros(300, dtype=torch.int32, device=DEV)
    need = L.gist_spmm_blocks_bytes(3)
    # per block: count image, rem_cnt, rem_col, the pair descriptor; then (batches of <= 256 blocks) two pair images each
    assert need == 3 * (16 * 128 * 16 + 128 * 4 + 128 * 8 * 4 + 16) + 3 * 2 * 16 * 128 * 16
    buf = torch.empty(need + 16, dtype=torch.uint8, device=DEV)
    assert L.gist_spmm_blocks_prepare(rp.data_ptr(), cl.data_ptr(), 300, None, 0, buf.data_ptr(), need - 1, None) < 0
    assert b'buffer too small' in L.gist_last_error()
    assert L.gist_spmm_blocks_prepare(rp.data_ptr(), cl.data_ptr(), 300, None, 0, buf.data_ptr() + 4, need, None) < 0
    assert L.gist_spmm_blocks_prepare(rp.data_ptr(), cl.data_ptr(), 300, None, 0, None, need, None) < 0
    assert L.gist_spmm_blocks_prepare(rp.data_ptr(), cl.data_ptr(), 300, None, 0, buf.data_ptr(), need, None) == 0
    x = torch.zeros(300, 2048, device=DEV)
    y = torch.zeros(300, 2048, device=DEV)
    assert L.gist_spmm_csr_prepared_f32(rp.data_ptr(), cl.data_ptr(), x.data_ptr(), 2048, y.data_ptr(), 2048, 300,
                                        2048, None, None, 0, None, 0, None, None) < 0
    torch.cuda.synchronize()


def _sibling_graph(rs, n_parts=20, part=102, siblings=((3, 7), (10, 11), (11, 15), (10, 15)), cross_per_row=45, extra_hub=0):
    """A cluster batch whose parts are dense inside (40 in-part neighbours per row), with a few neighbours anywhere in the
    batch, and SIBLING parts: every row of one has `cross_per_row` neighbours in the other (one community cut in two /
    three) -- far more than the per-row list of outside neighbours holds."""
    sizes = part + rs.randint(-3, 4, n_parts)
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    n = int(cuts[-1])
    src, dst = [], []
    for p in range(n_parts):
        rows = np.arange(cuts[p], cuts[p + 1])
        d_ = np.repeat(rows, 40)
        src.append(rs.randint(cuts[p], cuts[p + 1], d_.size)); dst.append(d_)
        d_ = np.repeat(rows, 1)
        src.append(rs.randint(0, n, d_.size)); dst.append(d_)
    for (p, q) in siblings:
        for (a, b) in ((p, q), (q, p)):
            d_ = np.repeat(np.arange(cuts[a], cuts[a + 1]), cross_per_row)
            src.append(rs.randint(cuts[b], cuts[b + 1], d_.size)); dst.append(d_)
    if extra_hub:
        src.append(rs.randint(0, n, extra_hub)); dst.append(np.full(extra_hub, int(cuts[3]) + 5))
    return n, cuts, np.concatenate(src), np.concatenate(dst)


@pytest.mark.parametrize('d', [2048, 4096, 1540])
def test_spmm_sibling_parts_as_dense_pairs(hip, d):
    """Two (three) parts of one community in a batch: the off-diagonal blocks between them are multiplied like diagonal
    ones (the prepare kernel finds them, gist_amd/csrc/spmm_mfma.hip).  Against the oracle in both forms, with the dropout
    mask in the store, BIT FOR BIT on integer-valued features (counts x bf16x3 pieces are exact), the pairs really found,
    and the same prepared structure read by the kernel that has no pair product (F = 602: such rows are gathered in full)."""
    from gist_amd import _lib
    rs = np.random.RandomState(d)
    n, cuts, src, dst = _sibling_graph(rs, extra_hub=500)
    rowptr, col = O.csr_from_edges(src, dst, n)
    norm = O.in_degree_norm(rowptr)
    rb = dev(cuts, torch.int32)
    rp, cl = dev(rowptr, torch.int32), dev(col, torch.int32)
    prep = hip.spmm_prepare(rp, cl, rb)
    nb = len(cuts) - 1
    stride = int(_lib.load().gist_spmm_block_image_bytes())
    rec = prep[:nb * stride].view(nb, stride)
    pinfo = rec[:, stride - 16:].contiguous().view(torch.int32).view(nb, 4).cpu().numpy()
    want_pairs = {3: {7}, 7: {3}, 10: {11, 15}, 11: {10, 15}, 15: {10, 11}}
    for b in range(nb):
        got = {int(np.searchsorted(cuts, pinfo[b, 2 * j], side='right') - 1) for j in range(2) if pinfo[b, 2 * j + 1] > 0}
        assert got == want_pairs.get(b, set()), (b, pinfo[b])
        for j in range(2):
            if pinfo[b, 2 * j + 1] > 0:
                q = int(np.searchsorted(cuts, pinfo[b, 2 * j], side='right') - 1)
                assert pinfo[b, 2 * j] == cuts[q] and pinfo[b, 2 * j + 1] == cuts[q + 1] - cuts[q]
    x = rs.randn(n, 2 * d).astype(np.float32)
    xt = dev(x)
    hip.spmm(rp, cl, xt[:, :d], xt[:, d:], out_scale=dev(norm), row_blocks=rb, blocked=True, prepared=prep)
    ref = x.copy()
    ref[:, d:] = O.spmm_sum(rowptr, col, np.ascontiguousarray(x[:, :d]), out_scale=norm)
    close(xt, ref)
    # backward form on the reversed graph: src_scale + accumulate
    t_rp, t_cl = O.transpose_csr(rowptr, col)
    trp, tcl = dev(t_rp, torch.int32), dev(t_cl, torch.int32)
    prep_t = hip.spmm_prepare(trp, tcl, rb)
    g = rs.randn(n, 2 * d).astype(np.float32)
    gt = dev(g)
    hip.spmm(trp, tcl, gt[:, d:], gt[:, :d], src_scale=dev(norm), accumulate=True, row_blocks=rb, blocked=True, prepared=prep_t)
    dh = np.ascontiguousarray(g[:, :d])
    O.spmm_sum(t_rp, t_cl, g[:, d:], src_scale=norm, out=dh, accumulate=True)
    close(gt[:, :d], dh)
    # the forward dropout mask in the store
    if d % 4 == 0:
        xd = dev(x)
        p_, seed, off = 0.3, 11, 1000
        hip.spmm_drop(rp, cl, xd[:, :d], xd[:, d:], 1, p_, seed, off + d, off, 2 * d, out_scale=dev(norm), row_blocks=rb,
                      prepared=prep)
        mask = _dropout_mask_ref(n, 2 * d, p_, seed, off)[:, d:]
        close(xd[:, d:], ref[:, d:] * mask / (1 - p_))
    # integers: every partial sum exact in fp32 whatever the order -> bit for bit
    xi = np.zeros((n, 2 * d), np.float32)
    xi[:, :d] = rs.randint(-(1 << 20), 1 << 20, (n, d)).astype(np.float32) / 4096.0 * (rs.rand(n, d) < 0.05)
    xit = dev(xi)
    hip.spmm(rp, cl, xit[:, :d], xit[:, d:], row_blocks=rb, blocked=True, prepared=prep)
    want = O.spmm_sum(rowptr, col, np.ascontiguousarray(xi[:, :d]))
    ref64 = np.zeros((n, d))
    np.add.at(ref64, np.repeat(np.arange(n), np.diff(rowptr)), xi[col, :d].astype(np.float64))
    assert np.array_equal(want.astype(np.float64), ref64), 'test data: the fp32 sums are not exact'
    assert np.array_equal(xit[:, d:].cpu().numpy(), want)
    # the unprepared kernel (no pairs: those rows walk their edge lists) agrees to rounding
    x2 = dev(x)
    hip.tuning('spmm_kernel', 2)
    try:
        hip.spmm(rp, cl, x2[:, :d], x2[:, d:], out_scale=dev(norm), row_blocks=rb, blocked=True)
    finally:
        hip.tuning('spmm_kernel', 0)
    close(x2, ref)
    # F = 602: the fp32 block-dense kernel reads the same structure
    f = 602
    xf = rs.randn(n, 2 * f).astype(np.float32)
    xft = dev(xf)
    hip.spmm(rp, cl, xft[:, :f], xft[:, f:], out_scale=dev(norm), row_blocks=rb, blocked=True, prepared=prep)
    reff = xf.copy()
    reff[:, f:] = O.spmm_sum(rowptr, col, np.ascontiguousarray(xf[:, :f]), out_scale=norm)
    close(xft, reff)


def _spmm_blocked_case(hip, n, d, deg, hub, blocks, prepared=False):
    rs = np.random.RandomState(n + d)
    # locality: most edges inside chunks of ~100 rows
    m = n * deg
    dst = rs.randint(0, max(n - 1, 1), m)
    near = (dst // 100) * 100 + rs.randint(0, 100, m)
    src = np.where(rs.rand(m) < 0.9, np.minimum(near, n - 1), rs.randint(0, n, m))
    if hub:
        src = np.concatenate([src, rs.randint(0, n, hub)])
        dst = np.concatenate([dst, np.full(hub, 1)])
    rowptr, col = O.csr_from_edges(src, dst, n)
    x = rs.randn(n, 2 * d).astype(np.float32)
    norm = O.in_degree_norm(rowptr)
    if blocks == 'uniform':
        rb = None
    else:
        step = 100 if blocks == 'parts' else 300
        cuts = list(range(0, n, step)) + [n]
        cuts = sorted(set(cuts + ([57] if n > 57 else [])))
        rb = dev(np.array(cuts), torch.int32)
    rp, cl = dev(rowptr, torch.int32), dev(col, torch.int32)
    xt = dev(x)
    prep = hip.spmm_prepare(rp, cl, rb) if prepared else None
    hip.spmm(rp, cl, xt[:, :d], xt[:, d:], out_scale=dev(norm), row_blocks=rb, blocked=True, prepared=prep)
    ref = x.copy()
    ref[:, d:] = O.spmm_sum(rowptr, col, np.ascontiguousarray(x[:, :d]), out_scale=norm)
    close(xt, ref)
    if prepared and d % 4 == 0 and d >= 1536:      # (the widths the prepared entry point takes) same arithmetic
        x2 = dev(x)
        hip.tuning('spmm_kernel', 2)
        try:
            hip.spmm(rp, cl, x2[:, :d], x2[:, d:], out_scale=dev(norm), row_blocks=rb, blocked=True)
        finally:
            hip.tuning('spmm_kernel', 0)
        # (blocks cut at row 57 make [0, 57) and [57, 100) sibling parts: the prepared structure then multiplies their
        # off-diagonal blocks as pairs, another order of the same fp32 sums; without pairs the arithmetic is the same)
        if blocks == 'parts' and n > 57:
            close(x2, xt.cpu().numpy(), tol=2e-5)
        else:
            assert torch.equal(x2, xt)
    # backward form: src_scale + accumulate, reversed graph
    t_rp, t_cl = O.transpose_csr(rowptr, col)
    g = rs.randn(n, 2 * d).astype(np.float32)
    gt = dev(g)
    trp, tcl = dev(t_rp, torch.int32), dev(t_cl, torch.int32)
    prep_t = hip.spmm_prepare(trp, tcl, rb) if prepared else None
    hip.spmm(trp, tcl, gt[:, d:], gt[:, :d], src_scale=dev(norm), accumulate=True, row_blocks=rb, blocked=True,
             prepared=prep_t)
    dh = np.ascontiguousarray(g[:, :d])
    O.spmm_sum(t_rp, t_cl, g[:, d:], src_scale=norm, out=dh, accumulate=True)
    close(gt[:, :d], dh)


@pytest.mark.parametrize('n,f,n_fit', [(5000, 602, 3300), (700, 37, 700), (300, 100, 17)])
def test_standard_scaler_on_device(hip, n, f, n_fit):
    """gist_standard_scaler_f32 == the oracle's sklearn StandardScaler restatement (pinned to
    sklearn itself on the CPU): statistics over the fit rows only, float64 mean / population
    variance, zero-variance column -> scale 1, transform of every row."""
    from oracle import gist_oracle as O
    rs = np.random.RandomState(n + f)
    x = (rs.randn(n, f) * rs.uniform(0.01, 30, f) + rs.uniform(-50, 50, f)).astype(np.float32)
    x[:, f // 2] = -3.5
    rows = np.sort(rs.choice(n, n_fit, replace=False))
    mask = np.zeros(n, bool)
    mask[rows] = True
    want, mean, var = O.standard_scaler(x, mask)
    xt = torch.from_numpy(x).to(DEV)
    m, v = hip.standard_scale_(xt, torch.from_numpy(rows.astype(np.int32)).to(DEV))
    assert np.allclose(m.cpu().numpy(), mean, rtol=1e-12, atol=1e-12)
    assert np.allclose(v.cpu().numpy(), var, rtol=1e-9, atol=1e-12)
    assert np.abs(xt.cpu().numpy() - want).max() <= 1e-6 * max(1.0, np.abs(want).max())
    assert np.array_equal(xt[:, f // 2].cpu().numpy(), want[:, f // 2])
    if n_fit == n:                                     # rows = NULL: all rows
        xt2 = torch.from_numpy(x).to(DEV)
        hip.standard_scale_(xt2)
        assert torch.equal(xt2, xt)


def test_use_pp_preaggregation():
    """--use-pp (sampler.py:58-69, modules.py:100-159): ClusterIter(use_pp=True) turns the train
    graph's features into [X | A^X] once, on the device == the oracle; a GraphSAGELayer built with
    use_pp=True then skips its own aggregation in training and gives the same output as the layer
    that aggregates itself."""
    import random
    import torch.nn.functional as F
    from gist_amd import datasets
    from gist_amd.modules import GraphSAGELayer
    from gist_amd.sampler import ClusterIter
    from oracle import gist_oracle as O
    ds = datasets.toy(seed=8, n_feats=30)
    g = ds.g
    train_nid = np.nonzero(g.ndata['train_mask'].numpy())[0].astype(np.int64)
    random.seed(1)
    it_pp = ClusterIter('toy', g, len(ds.par_li), 4, train_nid, use_pp=True,
                        par_li=[p.copy() for p in ds.par_li], device=torch.device(DEV))
    random.seed(1)
    it = ClusterIter('toy', g, len(ds.par_li), 4, train_nid, use_pp=False,
                     par_li=[p.copy() for p in ds.par_li], device=torch.device(DEV))
    tg = it.g
    want = O.preaggregate(tg.rowptr.cpu().numpy().astype(np.int64), tg.col.cpu().numpy().astype(np.int64),
                          tg.ndata['feat'].cpu().numpy())
    got = it_pp.g.ndata['feat'].cpu().numpy()
    assert got.shape == (tg.number_of_nodes(), 60) and np.abs(got - want).max() < 1e-5
    # the layer on the WHOLE train graph: pre-aggregated input == own aggregation
    torch.manual_seed(0)
    pp = GraphSAGELayer(30, 16, F.relu, 0.0, use_pp=True).to(DEV)
    own = GraphSAGELayer(30, 16, F.relu, 0.0, use_pp=False).to(DEV)
    own.load_state_dict(pp.state_dict())
    pp.train(); own.train()
    a = pp(it_pp.g, it_pp.g.ndata['feat'])
    b = own(tg, tg.ndata['feat'])
    assert (a - b).abs().max().item() < 1e-4
    pp.eval()                                          # eval: aggregates itself (modules.py:133)
    c = pp(tg, tg.ndata['feat'])
    assert (c - b).abs().max().item() < 1e-4
    sub = next(iter(it_pp))                            # batches carry the 2F-wide features
    assert sub.ndata['feat'].shape[1] == 60
