"""Error of the f16x3 split path and of the fp32-MFMA kernel against float64 (dev tool):
max |err| / sum_k|a||b|, rms(err)/rms(ref), and the mean signed error (rounding bias)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from gist_amd import hip
import test_gemm_h3_gpu as T

for kind in ('normal', 'train', 'grad'):
    for (form, m, n, k) in T.SHAPES:
        gen = torch.Generator(device=T.DEV).manual_seed(m + 3 * n + 7 * k)
        a, w = T._operands(form, m, n, k, gen, kind)
        rows = torch.arange(0, m, max(1, m // 192), device=T.DEV)
        ref, den = T._ref64(form, a, w, rows)
        out = {}
        for mode in ('f32', 'f16x3'):
            hip.gemm_mode(mode)
            y = T._run(hip, form, a, w, None, m, n)[rows].double()
            e = (y - ref)
            out[mode] = ((e.abs() / den).max().item(), (e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(),
                         (e / den).mean().item())
        print('%-6s %s m=%d n=%d k=%d | f32 max %.2e rms %.2e bias %+.2e | f16x3 max %.2e rms %.2e bias %+.2e' % (
            (kind, form, m, n, k) + out['f32'] + out['f16x3']), flush=True)
