"""GPU tests of the convert-on-load bf16x3 projection (gist_amd/csrc/gemm_b3c.hip) through the C ABI: the
shapes below the pre-split path's thresholds (per-rank widths 1024 and 512 of the N = 4 / 8 points, ragged
edges, k tails, split-K, output windows) in GEMM mode bf16x3 against float64 at the fp32-MFMA kernel's error
level, exactness where fp32 is exact, and the deferred-slab form."""
import numpy as np
import pytest
import torch

from tests.test_gemm_b3_gpu import _operands, _ref64, _run

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def hip():
    from gist_amd import hip as h
    assert h.device_count() >= 1
    prev = h.gemm_mode()
    h.tuning('b3c', 2)               # also below the production flop threshold
    yield h
    h.gemm_mode(prev)
    h.tuning('b3c', 0)


def _taken(hip, form, m, n, k, gen):
    """The call really runs on gemm_b3c_kernel: with the path switched off (tuning hook) the result
    differs in its low bits from the one with it on (fp32 products vs six bf16 cross terms)."""
    a, w = _operands(form, m, n, k, gen, 'normal')
    on = _run(hip, form, a, w, None, m, n)
    hip.tuning('b3c', 1)
    off = _run(hip, form, a, w, None, m, n)
    hip.tuning('b3c', 2)
    return not torch.equal(on, off)


SHAPES = [('nt', 2046, 1024, 1204), ('nt', 2046, 1024, 2048), ('nn', 2046, 2048, 1024), ('tn', 1024, 2048, 2046),
          ('tn', 1024, 1204, 2046), ('nt', 2046, 512, 1204), ('nn', 2046, 1024, 512), ('tn', 512, 1024, 2046),
          ('nt', 1140, 512, 200), ('tn', 512, 200, 1140),          # Amazon-like per-rank shapes
          ('nt', 333, 130, 100), ('nn', 77, 68, 72), ('tn', 68, 200, 333)]


@pytest.mark.parametrize('form,m,n,k', SHAPES)
@pytest.mark.parametrize('kind', ['normal', 'train', 'grad', 'cancel', 'range', 'edge'])
def test_b3c_error_at_fp32_mfma_level(hip, form, m, n, k, kind):
    hip.gemm_mode('bf16x3')
    gen = torch.Generator(device=DEV).manual_seed(m + 3 * n + 7 * k)
    if kind == 'normal':
        assert _taken(hip, form, m, n, k, gen)
    a, w = _operands(form, m, n, k, gen, kind)
    bias = torch.randn(n, device=DEV, generator=gen) * 1e-3 if form == 'nt' and kind == 'normal' else None
    rows = torch.arange(0, m, max(1, m // 192), device=DEV)
    ref, den = _ref64(form, a, w, rows)
    if bias is not None:
        ref = ref + bias.double()
    y3 = _run(hip, form, a, w, bias, m, n)[rows].double()
    hip.gemm_mode('f32')
    y1 = _run(hip, form, a, w, bias, m, n)[rows].double()
    hip.gemm_mode('bf16x3')
    assert torch.isfinite(y3).all()
    den = den.clamp(min=1e-300)
    e3 = ((y3 - ref).abs() / den).max().item()
    e1 = ((y1 - ref).abs() / den).max().item()
    r3 = ((y3 - ref).pow(2).mean().sqrt() / den.pow(2).mean().sqrt()).item()
    r1 = ((y1 - ref).pow(2).mean().sqrt() / den.pow(2).mean().sqrt()).item()
    # same bars as the pre-split kernel (tests/test_gemm_b3_gpu.py): the arithmetic is the same
    assert e3 <= max(3.0 * e1, 6e-7), (kind, e3, e1)
    assert r3 <= max((2.5 if kind == 'range' else 1.25) * r1, 5e-8), (kind, r3, r1)


def test_b3c_is_exact_where_fp32_is(hip):
    hip.gemm_mode('bf16x3')
    gen = torch.Generator(device=DEV).manual_seed(3)
    m, n, k = 530, 300, 250
    a = torch.randint(-8, 9, (m, k), device=DEV, generator=gen).float()
    w = torch.randint(-8, 9, (n, k), device=DEV, generator=gen).float()
    assert torch.equal(_run(hip, 'nt', a, w, None, m, n), (a.double() @ w.double().t()).float())
    wt = w.t().contiguous()
    assert torch.equal(_run(hip, 'nn', a, wt, None, m, n), (a.double() @ w.double().t()).float())
    at = a.t().contiguous()
    assert torch.equal(_run(hip, 'tn', at, wt, None, m, n), (a.double() @ w.double().t()).float())
    b = torch.randn(256, 320, device=DEV, generator=gen) * torch.exp2(
        torch.randint(-30, 31, (256, 320), device=DEV, generator=gen).float())
    eye = torch.eye(256, device=DEV)
    assert torch.equal(_run(hip, 'nn', eye, b, None, 256, 320), b)      # every one of the 24 bits of b
    assert torch.equal(_run(hip, 'tn', eye, b, None, 256, 320), b)
    bt = b.t().contiguous()
    assert torch.equal(_run(hip, 'nt', eye, bt, None, 256, 320), b)
    assert torch.equal(_run(hip, 'nt', bt, eye, None, 320, 256), bt)


def test_b3c_output_window_k_tail_and_split_k(hip):
    """A result written into a window of a larger buffer touches nothing around it; a k that is no multiple
    of 32 or 8; forced split-K (the slabs reduced by the call) equals one k slice up to summation order."""
    hip.gemm_mode('bf16x3')
    gen = torch.Generator(device=DEV).manual_seed(5)
    m, n, k = 530, 300, 203 * 4
    a = torch.randn(m, k, device=DEV, generator=gen)
    w = torch.randn(n, k, device=DEV, generator=gen)
    ybuf = torch.full((m + 3, n + 8), 7.0, device=DEV)
    hip.gemm_nt(a, w, None, ybuf[:m, 4:4 + n])
    ref = (a.double() @ w.double().t()).float()
    assert (ybuf[:m, 4:4 + n] - ref).abs().max().item() < 2e-6 * ref.abs().max().item() * np.sqrt(k)
    assert (ybuf[m:] == 7.0).all() and (ybuf[:, :4] == 7.0).all() and (ybuf[:, 4 + n:] == 7.0).all()
    one = _run(hip, 'nt', a, w, None, m, n)
    hip.tuning('gemm_tile', 64)
    hip.tuning('gemm_splits', 4)
    try:
        four = _run(hip, 'nt', a, w, None, m, n)
    finally:
        hip.tuning('gemm_tile', 0)
        hip.tuning('gemm_splits', 0)
    assert (four - one).abs().max().item() < 1e-5 * ref.abs().max().item()
    # a k-major operand whose k is not a multiple of 8 (the batch rows of a dW projection)
    k2 = 2046
    g = torch.randn(k2, 130, device=DEV, generator=gen)
    z = torch.randn(k2, 260, device=DEV, generator=gen)
    got = _run(hip, 'tn', g, z, None, 130, 260)
    ref2 = (g.double().t() @ z.double()).float()
    assert (got - ref2).abs().max().item() < 1e-5 * ref2.abs().max().item()


def test_b3c_deferred_slabs(hip):
    """gist_gemm_slabs_f32 in mode bf16x3: the slabs of a split weight-gradient projection summed in slab
    order are BIT-equal to the call that reduces them itself."""
    from gist_amd import _lib
    L = _lib.load()
    hip.gemm_mode('bf16x3')
    gen = torch.Generator(device=DEV).manual_seed(7)
    m, n, k = 512, 1204, 2046
    a = torch.randn(k, m, device=DEV, generator=gen)
    b = torch.randn(k, n, device=DEV, generator=gen)
    ref = torch.empty(m, n, device=DEV)
    hip.tuning('gemm_splits', 4)          # (the kernel's own choice for this shape is one k slice)
    try:
        hip.gemm_tn(a, b, ref)
        need = int(L.gist_gemm_workspace_bytes(m, n, k))
        assert need >= 4 * m * n * 4
        slabs = torch.full((need // 4,), float('nan'), device=DEV)
        c = torch.full((m, n), float('nan'), device=DEV)
        ns = hip.gemm_slabs('tn', a, b, None, c, slabs.view(torch.uint8))
    finally:
        hip.tuning('gemm_splits', 0)
    assert ns == 4
    acc = torch.zeros(m * n, device=DEV)
    for s in range(ns):
        acc = acc + slabs[s * m * n:(s + 1) * m * n]
    assert torch.equal(acc.view(m, n), ref)


@pytest.mark.parametrize('form,m,n,k', [('nt', 2046, 1024, 2048), ('nn', 2046, 1024, 512), ('tn', 512, 1204, 2046),
                                        ('nt', 333, 130, 100), ('nn', 77, 68, 72), ('tn', 68, 200, 333),
                                        ('nt', 530, 300, 812), ('nt', 129, 257, 96)])
@pytest.mark.parametrize('splits', [1, 3])
def test_b3c_tiles_agree_bit_for_bit(hip, form, m, n, k, splits):
    """The 128 x 128 tile (one fragment set refilled piece by piece, bare barriers), the 128 x 64 and the 64 x 64
    tile (two fragment sets) give every output element the same MFMAs in the same order: equal bits, for one
    k slice and for the same number of slices."""
    hip.gemm_mode('bf16x3')
    gen = torch.Generator(device=DEV).manual_seed(11 * m + n + k)
    a, w = _operands(form, m, n, k, gen, 'normal')
    bias = torch.randn(n, device=DEV, generator=gen) if form == 'nt' else None
    out = {}
    hip.tuning('gemm_splits', splits)
    try:
        for tile in (64, 128, 128128):
            hip.tuning('gemm_tile', tile)
            out[tile] = _run(hip, form, a, w, bias, m, n)
    finally:
        hip.tuning('gemm_tile', 0)
        hip.tuning('gemm_splits', 0)
    assert torch.isfinite(out[64]).all()
    assert torch.equal(out[64], out[128])
    assert torch.equal(out[64], out[128128])


@pytest.mark.parametrize('form,m,n,k', [('nt', 2100, 1024, 2048), ('nt', 2100, 1024, 1204), ('nn', 2100, 2048, 1024),
                                        ('nt', 2080, 1024, 500), ('nn', 2150, 1032, 1024)])
def test_b3c_tail_units(hip, form, m, n, k):
    """More tiles than the chip holds at once, by a few (a batch of 2049-2112 rows: 528 tiles of 64 x 64 on 512 workgroup
    slots, 272 of 128 x 128 on 256): the tiles of the last round run as k slices summed in slice order by a follow-up
    launch.  Same result as whole tiles up to the order of the fp32 partial sums, bias once, the same bits on every run,
    exact on integers, nothing written outside C."""
    hip.gemm_mode('bf16x3')
    gen = torch.Generator(device=DEV).manual_seed(m + 5 * n + 11 * k)
    a, w = _operands(form, m, n, k, gen, 'normal')
    bias = torch.randn(n, device=DEV, generator=gen) if form == 'nt' else None
    y_tail = _run(hip, form, a, w, bias, m, n)
    for _ in range(3):
        assert torch.equal(_run(hip, form, a, w, bias, m, n), y_tail)
    hip.tuning('b3_tail', 1)
    try:
        y_whole = _run(hip, form, a, w, bias, m, n)
    finally:
        hip.tuning('b3_tail', 0)
    assert torch.isfinite(y_tail).all()
    assert not torch.equal(y_tail, y_whole), 'not a tail-unit shape (the hook changed nothing)'
    rows = torch.cat([torch.arange(0, m, max(1, m // 96), device=DEV), torch.arange(m - 70, m, device=DEV)])
    ref, den = _ref64(form, a, w, rows)
    if bias is not None:
        ref = ref + bias.double()
    e_tail = ((y_tail[rows].double() - ref).abs() / den).max().item()
    e_whole = ((y_whole[rows].double() - ref).abs() / den).max().item()
    assert e_tail <= max(1.5 * e_whole, 4e-7), (e_tail, e_whole)
    assert (y_tail - y_whole).abs().max().item() <= 4e-6 * den.max().item()
    sa, sb = {'nt': ((m, k), (n, k)), 'nn': ((m, k), (k, n))}[form]
    ai = torch.randint(-8, 9, sa, device=DEV, generator=gen).float()
    wi = torch.randint(-8, 9, sb, device=DEV, generator=gen).float()
    ybuf = torch.full((m + 2, n + 8), 7.0, device=DEV)
    out = ybuf[:m, 4:4 + n]
    if form == 'nt':
        hip.gemm_nt(ai, wi, None, out)
        exact = ai.double() @ wi.double().t()
    else:
        hip.gemm_nn(ai, wi, out)
        exact = ai.double() @ wi.double()
    assert torch.equal(out, exact.float())
    assert (ybuf[m:] == 7.0).all() and (ybuf[:, :4] == 7.0).all() and (ybuf[:, 4 + n:] == 7.0).all()
