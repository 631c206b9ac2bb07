"""GEMM microbenchmark (dev tool): times the three layouts at the step's shapes with and
without leading-dimension padding.  Interleaved rounds in one process (guide rule 24)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gist_amd import hip

dev = torch.device('cuda', 0)
torch.manual_seed(0)


def buf(r, c, pad):
    b = torch.randn(r, c + pad, device=dev)
    return b[:, :c]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


shapes = [('nt', 2046, 4096, 8192), ('nn', 2046, 8192, 4096), ('tn', 4096, 8192, 2046),
          ('nt', 2046, 4096, 1204), ('tn', 4096, 1204, 2046), ('nt', 2048, 4096, 8192),
          ('nt', 4096, 4096, 4096)]
for pad in (0, 32, 36):
    for (lay, m, n, k) in shapes:
        if lay == 'nt':
            a, w, y = buf(m, k, pad), buf(n, k, pad), buf(m, n, pad)
            f = lambda: hip.gemm_nt(a, w, None, y)
        elif lay == 'nn':
            g, w, z = buf(m, k, pad), buf(k, n, pad), buf(m, n, pad)
            f = lambda: hip.gemm_nn(g, w, z)
        else:
            g, a, d = buf(k, m, pad), buf(k, n, pad), buf(m, n, pad)
            f = lambda: hip.gemm_tn(g, a, d)
        med, mn = timeit(f)
        print('pad %2d %s m=%d n=%d k=%d  median %.3f ms  %.1f TF   (min %.3f ms %.1f TF)' % (
            pad, lay, m, n, k, med, 2.0 * m * n * k / med / 1e9, mn, 2.0 * m * n * k / mn / 1e9), flush=True)
