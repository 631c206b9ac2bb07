"""Dev tool: bf16x3 GEMM time on the big NT shape with the library in GIST_LIB_PATH."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gist_amd import hip
dev = torch.device('cuda', 0)
hip.gemm_mode('bf16x3')
out = []
for (m, n, k) in [(2046, 4096, 8192), (4096, 8192, 2046)]:
    a, w, y = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), torch.empty(m, n, device=dev)
    for _ in range(3): hip.gemm_nt(a, w, None, y)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); hip.gemm_nt(a, w, None, y); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    out.append('%dx%dx%d %.3f ms' % (m, n, k, ts[len(ts) // 2]))
print(os.environ.get('GIST_LIB_PATH', 'default')[-14:], ' | '.join(out))
