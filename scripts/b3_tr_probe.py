"""Dev tool: the bf16x3 GEMM with B given as [k][n] -- B's row form read transposed (ds_read_b64_tr_b16, default) against
the transposed split (tuning hook b3_tr = 1): equality of the results and time per call (pre-pass included)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gist_amd import hip
dev = torch.device('cuda', 0)
hip.gemm_mode('bf16x3')
for (m, n, k) in [(2046, 8192, 4096), (2046, 4096, 2048), (1000, 1280, 1000), (2046, 1204, 4096)]:
    gen = torch.Generator(device=dev).manual_seed(m + n + k)
    a = torch.randn(m, k, device=dev, generator=gen)
    w = torch.randn(k, n, device=dev, generator=gen)
    ref = (a.double() @ w.double())
    outs, times = [], []
    for knob in (0, 2):
        hip.tuning('b3_tr', knob)
        y = torch.empty(m, n, device=dev)
        for _ in range(3):
            hip.gemm_nn(a, w, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            hip.gemm_nn(a, w, y)
        e1.record(); torch.cuda.synchronize()
        outs.append(y.clone()); times.append(e0.elapsed_time(e1) / 20 * 1e3)
    err = [(o.double() - ref).abs().max().item() / ref.abs().max().item() for o in outs]
    print('%d x %d x %d: transposed split %.1f us, transposed READ %.1f us; bit-equal %s; max err / max |ref| %.2e %.2e'
          % (m, n, k, times[0], times[1], bool(torch.equal(outs[0], outs[1])), err[0], err[1]), flush=True)
hip.tuning('b3_tr', 0)
