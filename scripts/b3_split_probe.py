"""Dev tool: the bf16x3 pre-split GEMM on the dW_0 shape of the H = 4096 step (160 tiles on 256 CUs) with forced k
slices; standalone NT calls (per-call pre-pass included: compare the differences)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gist_amd import hip
dev = torch.device('cuda', 0)
hip.gemm_mode('bf16x3')
for (m, n, k) in [(4096, 1204, 2046), (4096, 8192, 2046)]:
    a, w, y = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), torch.empty(m, n, device=dev)
    out = []
    for sp in (0, 1, 2, 3, 4):
        hip.tuning('gemm_splits', sp)
        for _ in range(3):
            hip.gemm_nt(a, w, None, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            hip.gemm_nt(a, w, None, y)
        e1.record(); torch.cuda.synchronize()
        out.append('splits %d: %.1f us' % (sp, e0.elapsed_time(e1) / 20 * 1e3))
    hip.tuning('gemm_splits', 0)
    print('%d x %d x %d  ' % (m, n, k) + ' | '.join(out), flush=True)
