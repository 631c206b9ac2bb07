"""f16x3 split GEMM vs the fp32-MFMA kernel on the training step's shapes (dev tool).
Interleaved rounds in one process; the time of a call includes the split pre-pass."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gist_amd import hip

dev = torch.device('cuda', 0)
torch.manual_seed(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


shapes = [('nt', 2046, 4096, 8192), ('nn', 2046, 8192, 4096), ('tn', 4096, 8192, 2046),
          ('nt', 2046, 4096, 1204), ('tn', 4096, 1204, 2046),
          ('nt', 2046, 1024, 2048), ('nn', 2046, 2048, 1024), ('tn', 1024, 2048, 2046),
          ('nt', 2046, 1024, 1204), ('nt', 2046, 512, 1024)]
for (lay, m, n, k) in shapes:
    if lay == 'nt':
        a, w, y = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), torch.empty(m, n, device=dev)
        f = lambda: hip.gemm_nt(a, w, None, y)
    elif lay == 'nn':
        a, w, y = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev), torch.empty(m, n, device=dev)
        f = lambda: hip.gemm_nn(a, w, y)
    else:
        a, w, y = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev), torch.empty(m, n, device=dev)
        f = lambda: hip.gemm_tn(a, w, y)
    res = {}
    for mode in ('f32', 'f16x3', 'bf16x3', 'f32', 'f16x3', 'bf16x3'):
        hip.gemm_mode(mode)
        res.setdefault(mode, []).append(timeit(f))
    t1, t3, tb = min(res['f32']), min(res['f16x3']), min(res['bf16x3'])
    gf = 2.0 * m * n * k / 1e9
    print('%s m=%d n=%d k=%d  f32 %.3f ms %.1f TF | f16x3 %.3f ms %.1f TF-equiv x%.2f | bf16x3 %.3f ms %.1f TF-equiv x%.2f' % (
        lay, m, n, k, t1, gf / t1, t3, gf / t3, t1 / t3, tb, gf / tb, t1 / tb), flush=True)
