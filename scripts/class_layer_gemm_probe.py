import sys, os
sys.path.insert(0, '/root/repo')
import torch
from gist_amd import hip, _lib
dev = torch.device('cuda', 0)
hip.gemm_mode('bf16x3')
L = _lib.load()
def t(f, it=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for n in (1024, 2048, 4096, 8192):
    m, k = 41, 2046
    a = torch.randn(k, 44, device=dev)[:, :m]
    b = torch.randn(k, n, device=dev)
    c = torch.empty(m, n, device=dev)
    out = []
    for sp in (0, 1, 2, 4, 8, 16):
        hip.tuning('gemm_splits', sp)
        out.append('%d: %.1f' % (sp, t(lambda: hip.gemm_tn(a, b, c))))
    hip.tuning('gemm_splits', 0)
    print('tn m=41 n=%d k=2046  splits->us  ' % n + '  '.join(out), flush=True)
    # the logits projection NT: 2046 x 41 x n
    z = torch.randn(2046, n, device=dev); w = torch.randn(m, n, device=dev); y = torch.empty(2046, 44, device=dev)[:, :m]
    out = []
    for sp in (0, 1, 2, 4, 8, 16):
        hip.tuning('gemm_splits', sp)
        out.append('%d: %.1f' % (sp, t(lambda: hip.gemm_nt(z, w, None, y))))
    hip.tuning('gemm_splits', 0)
    print('nt 2046 x 41 x %d     splits->us  ' % n + '  '.join(out), flush=True)
