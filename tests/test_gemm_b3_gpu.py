"""GPU tests of the bf16x3 split projection path (gist_amd/csrc/gemm_b3.hip) through the C ABI:
every fp32 operand is carried with ALL 24 significant bits (three bf16 pieces, the six cross terms
down to 2^-16), so its error against a float64 product must be at the fp32-MFMA kernel's level
(mode 'f32': rms within 1.25x, max within 3x) on the same operands -- on the shapes the training step runs, on operands of very
different magnitudes, and on ADVERSARIAL operands: heavy cancellation, rows whose in-row dynamic
range exceeds 2^17 (and 2^40), values at the edges of the fp32 exponent range."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def hip():
    from gist_amd import hip as h
    assert h.device_count() >= 1
    prev = h.gemm_mode()
    h.tuning('h3_min_gflop', 1)      # also exercise shapes below the production threshold
    h.tuning('h3_min_tiles', 16)
    yield h
    h.gemm_mode(prev)
    h.tuning('h3_min_gflop', 0)
    h.tuning('h3_min_tiles', 0)


def _run(hip, form, a, w, bias, m, n):
    out = torch.full((m, n), float('nan'), device=DEV)
    if form == 'nt':
        hip.gemm_nt(a, w, bias, out)
    elif form == 'nn':
        hip.gemm_nn(a, w, out)
    else:
        hip.gemm_tn(a, w, out)
    return out


def _shape(form, m, n, k):
    if form == 'nt':
        return (m, k), (n, k)
    if form == 'nn':
        return (m, k), (k, n)
    return (k, m), (k, n)


def _operands(form, m, n, k, gen, kind):
    sa, sb = _shape(form, m, n, k)
    a = torch.randn(*sa, device=DEV, generator=gen)
    b = torch.randn(*sb, device=DEV, generator=gen)
    kdim_a = 1 if form in ('nt', 'nn') else 0           # which axis of a is k
    kdim_b = 1 if form == 'nt' else 0
    if kind == 'train':           # post-LayerNorm/ReLU/dropout activations x U(-b, b) weights
        a = torch.relu(a) * 1.25 * (torch.rand(*sa, device=DEV, generator=gen) > 0.2)
        b = (torch.rand(*sb, device=DEV, generator=gen) - 0.5) * 0.022
    elif kind == 'grad':          # tiny gradients, row scales spread over e^(+-4)
        scale = torch.exp(2 * torch.randn(sa[0], 1, device=DEV, generator=gen)) if kdim_a == 1 else \
            torch.exp(2 * torch.randn(1, sa[1], device=DEV, generator=gen))
        a = a * 1e-6 * scale
    elif kind == 'cancel':        # heavy cancellation: large terms that sum to (almost) nothing
        half = k // 2
        if kdim_a == 1:
            a = torch.cat([a[:, :half], -a[:, :half] * (1 + 1e-6 * torch.randn(sa[0], half, device=DEV, generator=gen))], 1)
            a = torch.cat([a, torch.zeros(sa[0], k - 2 * half, device=DEV)], 1) * 1e3
        else:
            a = torch.cat([a[:half], -a[:half] * (1 + 1e-6 * torch.randn(half, sa[1], device=DEV, generator=gen))], 0)
            a = torch.cat([a, torch.zeros(k - 2 * half, sa[1], device=DEV)], 0) * 1e3
        if kdim_b == 1:
            b = torch.cat([b[:, :half], b[:, :half], b[:, 2 * half:]], 1)
        else:
            b = torch.cat([b[:half], b[:half], b[2 * half:]], 0)
    elif kind == 'range':         # in-row dynamic range 2^40: every element keeps its 24 bits
        e = torch.randint(-20, 21, sa, device=DEV, generator=gen).float()
        a = a * torch.exp2(e)
        e2 = torch.randint(-20, 21, sb, device=DEV, generator=gen).float()
        b = b * torch.exp2(e2)
    elif kind == 'edge':          # magnitudes near the ends of the fp32 exponent range
        a = a * 2.0 ** 60
        b = b * 2.0 ** -70
    return a.contiguous(), b.contiguous()


def _ref64(form, a, w, rows):
    a64, w64 = a.double(), w.double()
    if form == 'nt':
        return a64[rows] @ w64.t(), a64[rows].abs() @ w64.abs().t()
    if form == 'nn':
        return a64[rows] @ w64, a64[rows].abs() @ w64.abs()
    return a64[:, rows].t() @ w64, a64[:, rows].abs().t() @ w64.abs()


SHAPES = [('nt', 2046, 4096, 1204), ('nt', 2046, 4096, 8192), ('nn', 2046, 8192, 4096),
          ('tn', 4096, 8192, 2046), ('tn', 4096, 1204, 2046), ('nt', 1030, 1100, 1000),
          ('nn', 1500, 1024, 777), ('tn', 1024, 1204, 2046)]


@pytest.mark.parametrize('form,m,n,k', SHAPES)
@pytest.mark.parametrize('kind', ['normal', 'train', 'grad', 'cancel', 'range', 'edge'])
def test_bf16x3_error_at_fp32_mfma_level(hip, form, m, n, k, kind):
    from gist_amd import _lib
    L = _lib.load()
    hip.gemm_mode('bf16x3')
    assert L.gist_gemm_workspace_bytes(m, n, k) >= (m + n) * k * 6, 'shape not on the split path'
    gen = torch.Generator(device=DEV).manual_seed(m + 3 * n + 7 * k)
    a, w = _operands(form, m, n, k, gen, kind)
    bias = torch.randn(n, device=DEV, generator=gen) * 1e-3 if form == 'nt' and kind == 'normal' else None
    rows = torch.arange(0, m, max(1, m // 192), device=DEV)      # a sample of output rows
    ref, den = _ref64(form, a, w, rows)
    if bias is not None:
        ref = ref + bias.double()
    y3 = _run(hip, form, a, w, bias, m, n)[rows].double()
    hip.gemm_mode('f32')
    y1 = _run(hip, form, a, w, bias, m, n)[rows].double()
    hip.gemm_mode('bf16x3')
    assert torch.isfinite(y3).all()
    den = den.clamp(min=1e-300)
    e3 = ((y3 - ref).abs() / den).max().item()          # error relative to sum |a||b|
    e1 = ((y1 - ref).abs() / den).max().item()
    r3 = ((y3 - ref).pow(2).mean().sqrt() / den.pow(2).mean().sqrt()).item()
    r1 = ((y1 - ref).pow(2).mean().sqrt() / den.pow(2).mean().sqrt()).item()
    # 24-bit operands, 6 roundings per 32 products: the fp32 kernel's error LEVEL -- rms within
    # 1.25x of it on every class, max error (over ~10^6 outputs) within 3x.  Where mode 'f32' picks
    # split-K (k <= 2048) its summation chains are short and it is exceptionally accurate (max
    # 1.0-1.6e-7 of sum |a||b| against 3.4-5.6e-7 here = 3-5 ulp, whichever of the two term orders
    # the kernel has used), hence the floors.
    # 'range' (in-row dynamic range 2^40): a handful of products dominate every sum, accumulation
    # error vanishes and what shows is the dropped a2.b3 + a3.b2 (<= 2^-23 of a product) against
    # the fp32 kernel's single product rounding (2^-24): rms 1.5-2.2x, still ~2e-7 of sum |a||b|.
    assert e3 <= max(3.0 * e1, 6e-7), (kind, e3, e1)
    assert r3 <= max((2.5 if kind == 'range' else 1.25) * r1, 5e-8), (kind, r3, r1)


def test_bf16x3_is_exact_where_fp32_is(hip):
    """(1) small integers: every partial sum is an integer below 2^24 -> the exact product, bit for
    bit; (2) operands with full 24-bit mantissas against the identity: x = b1 + b2 + b3 must
    reproduce EVERY bit of x (the 22-bit f16x3 split cannot); also catches a swapped fragment or
    C/D map."""
    hip.gemm_mode('bf16x3')
    gen = torch.Generator(device=DEV).manual_seed(3)
    m, n, k = 1030, 1100, 1000
    a = torch.randint(-8, 9, (m, k), device=DEV, generator=gen).float()
    w = torch.randint(-8, 9, (n, k), device=DEV, generator=gen).float()
    y = _run(hip, 'nt', a, w, None, m, n)
    assert torch.equal(y, (a.double() @ w.double().t()).float())
    b = torch.randn(1024, 1152, device=DEV, generator=gen) * torch.exp2(
        torch.randint(-30, 31, (1024, 1152), device=DEV, generator=gen).float())
    eye = torch.eye(1024, device=DEV)
    assert torch.equal(_run(hip, 'nn', eye, b, None, 1024, 1152), b)
    assert torch.equal(_run(hip, 'tn', eye, b, None, 1024, 1152), b)
    bt = b.t().contiguous()
    assert torch.equal(_run(hip, 'nt', eye, bt, None, 1024, 1152), b)
    assert torch.equal(_run(hip, 'nt', bt, eye, None, 1152, 1024), bt)


def test_bf16x3_zero_operand_and_output_window(hip):
    hip.gemm_mode('bf16x3')
    gen = torch.Generator(device=DEV).manual_seed(5)
    m, n, k = 1030, 1100, 520
    a = torch.randn(m, k, device=DEV, generator=gen)
    w = torch.zeros(n, k, device=DEV)
    assert (_run(hip, 'nt', a, w, None, m, n) == 0).all()
    w = torch.randn(n, k, device=DEV, generator=gen)
    ybuf = torch.full((m + 3, n + 8), 7.0, device=DEV)
    hip.gemm_nt(a, w, None, ybuf[:m, 4:4 + n])
    ref = (a.double() @ w.double().t()).float()
    assert (ybuf[:m, 4:4 + n] - ref).abs().max().item() < 1e-5 * ref.abs().max().item()
    assert (ybuf[m:] == 7.0).all() and (ybuf[:, :4] == 7.0).all() and (ybuf[:, 4 + n:] == 7.0).all()


def test_bf16x3_non_finite_inputs_stay_in_their_row(hip):
    """A NaN or an Inf in one row of an operand makes that output row (A) / column (B) non-finite
    and leaves every other output exactly as without it (there are no shared scales at all)."""
    hip.gemm_mode('bf16x3')
    gen = torch.Generator(device=DEV).manual_seed(11)
    m, n, k = 1030, 1100, 520
    a = torch.randn(m, k, device=DEV, generator=gen)
    w = torch.randn(n, k, device=DEV, generator=gen)
    clean = _run(hip, 'nt', a, w, None, m, n)
    a2, w2 = a.clone(), w.clone()
    a2[7, 100] = float('nan')
    a2[500, 3] = float('inf')
    w2[33, 17] = float('nan')
    y = _run(hip, 'nt', a2, w2, None, m, n)
    touched = torch.zeros(m, n, dtype=torch.bool, device=DEV)
    touched[[7, 500]] = True
    touched[:, 33] = True
    assert not torch.isfinite(y[7]).any() and not torch.isfinite(y[:, 33]).any()
    assert not torch.isfinite(y[500]).any()
    assert torch.equal(y[~touched], clean[~touched])


def test_mode_switch(hip):
    from gist_amd import _lib
    L = _lib.load()
    hip.gemm_mode('f32')
    assert L.gist_gemm_workspace_bytes(2046, 4096, 8192) == 0
    hip.gemm_mode('bf16x3')
    assert hip.gemm_mode() == 'bf16x3'
    assert L.gist_gemm_workspace_bytes(2046, 4096, 8192) >= (2046 + 4096) * 8192 * 6
    hip.tuning('h3_min_tiles', 0)         # production thresholds: a skinny output is mostly tile padding
    try:
        small = L.gist_gemm_workspace_bytes(2046, 41, 8192)
        wide = L.gist_gemm_workspace_bytes(2046, 2048, 4096)
        hip.gemm_mode('f32')
        assert L.gist_gemm_workspace_bytes(2046, 41, 8192) == small      # fp32 split-K either way
        assert wide >= (2046 + 2048) * 4096 * 6 + 2 * 2046 * 2048 * 4    # 128 tiles: two k slices + slabs
    finally:
        hip.tuning('h3_min_tiles', 16)


@pytest.mark.parametrize('form,m,n,k', [('nt', 2046, 2048, 4096), ('nt', 2046, 2048, 1204),
                                        ('tn', 2048, 1204, 2046), ('nn', 700, 640, 3000)])
def test_bf16x3_split_k_equals_one_slice(hip, form, m, n, k):
    """Outputs with few 256 x 128 tiles are computed as k slices (one workgroup each) summed from
    fp32 slabs: same result as the single-slice kernel up to the order of the fp32 partial sums,
    bias added exactly once, uneven slice lengths (1204 -> 38 k tiles = 20 + 18) included."""
    hip.gemm_mode('bf16x3')
    gen = torch.Generator(device=DEV).manual_seed(5 * m + n + k)
    a, w = _operands(form, m, n, k, gen, 'normal')
    bias = torch.randn(n, device=DEV, generator=gen) if form == 'nt' else None
    from gist_amd import _lib
    L = _lib.load()
    assert L.gist_gemm_workspace_bytes(m, n, k) >= (m + n) * k * 6 + 2 * m * n * 4, 'not a split-K shape'
    y_auto = _run(hip, form, a, w, bias, m, n)
    hip.tuning('gemm_splits', 1)
    try:
        y_one = _run(hip, form, a, w, bias, m, n)
    finally:
        hip.tuning('gemm_splits', 0)
    rows = torch.arange(0, m, max(1, m // 128), device=DEV)
    ref, den = _ref64(form, a, w, rows)
    if bias is not None:
        ref = ref + bias.double()
    assert torch.isfinite(y_auto).all()
    e_auto = ((y_auto[rows].double() - ref).abs() / den).max().item()
    e_one = ((y_one[rows].double() - ref).abs() / den).max().item()
    assert e_auto <= max(1.5 * e_one, 4e-7), (e_auto, e_one)
    assert (y_auto - y_one).abs().max().item() <= 4e-6 * den.max().item()


TAIL_SHAPES = [('nt', 2100, 4096, 8192), ('nt', 2100, 4096, 1204), ('nn', 2100, 8192, 4096),
               ('nt', 2049, 4096, 1024), ('nt', 2304, 4096, 640), ('nt', 300, 16500, 512),
               ('tn', 4200, 2050, 1000)]


@pytest.mark.parametrize('form,m,n,k', TAIL_SHAPES)
def test_bf16x3_tail_units(hip, form, m, n, k):
    """A tile count just above a multiple of 256 (a batch of 2049-2304 rows has a ninth row tile): the tiles past
    the last full round run as k slices that meet through partials, the last slice to arrive summing them in slice
    order.  Same result as whole tiles up to the order of the fp32 partial sums, bias added exactly once, the
    same bits on every run (the sum does not depend on which slice arrives last), exact on integers, nothing
    written outside C (one valid row in the tail tile included)."""
    hip.gemm_mode('bf16x3')
    from gist_amd import _lib
    L = _lib.load()
    hip.tuning('b3_tail', 1)
    try:
        ws_off = L.gist_gemm_workspace_bytes(m, n, k)
    finally:
        hip.tuning('b3_tail', 0)
    assert L.gist_gemm_workspace_bytes(m, n, k) > ws_off >= (m + n) * k * 6, 'not a tail-unit shape'
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
    rows = torch.cat([torch.arange(0, m, max(1, m // 96), device=DEV), torch.arange(m - min(m, 60), m, device=DEV)])
    ref, den = _ref64(form, a, w, rows)
    if bias is not None:
        ref = ref + bias.double()
    e_tail = ((y_tail[rows].double() - ref).abs() / den).max().item()
    e_whole = ((y_whole[rows].double() - ref).abs() / den).max().item()
    assert e_tail <= max(1.5 * e_whole, 4e-7), (e_tail, e_whole)
    assert (y_tail - y_whole).abs().max().item() <= 4e-6 * den.max().item()
    # the full rounds are the same workgroups doing the same work (where the tail is exactly the ninth row tile)
    tn_ = -(-n // 128)
    if 2048 < m <= 2304 and (9 * tn_) % 256 == tn_:
        assert torch.equal(y_tail[:2048], y_whole[:2048])
    # integers: every partial sum exact -> bit for bit, in an output window with guard bands
    sa, sb = _shape(form, m, n, k)
    ai = torch.randint(-8, 9, sa, device=DEV, generator=gen).float()
    wi = torch.randint(-8, 9, sb, device=DEV, generator=gen).float()
    ybuf = torch.full((m + 2, n + 8), 7.0, device=DEV)
    out = ybuf[:m, 4:4 + n]
    if form == 'nt':
        hip.gemm_nt(ai, wi, None, out)
        exact = ai.double() @ wi.double().t()
    elif form == 'nn':
        hip.gemm_nn(ai, wi, out)
        exact = ai.double() @ wi.double()
    else:
        hip.gemm_tn(ai, wi, out)
        exact = ai.double().t() @ wi.double()
    assert torch.equal(out, exact.float())
    assert (ybuf[m:] == 7.0).all() and (ybuf[:, :4] == 7.0).all() and (ybuf[:, 4 + n:] == 7.0).all()
