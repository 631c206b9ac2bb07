"""Dev tool: one GEMM shape in a given mode, repeated (for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gist_amd import hip
dev = torch.device('cuda', 0)
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
m, n, k = 2046, 4096, 8192
a, w, y = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), torch.empty(m, n, device=dev)
hip.gemm_mode(mode)
for _ in range(20):
    hip.gemm_nt(a, w, None, y)
torch.cuda.synchronize()
