"""Dev tool: one GEMM shape under the tile/split override (GIST_GEMM_TILE / GIST_GEMM_SPLITS in
the environment of THIS script are passed to the library's tuning hooks)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gist_amd import hip
dev = torch.device('cuda', 0)
if os.environ.get('GIST_GEMM_TILE') and os.environ.get('GIST_GEMM_SPLITS'):
    hip.tuning('gemm_tile', int(os.environ['GIST_GEMM_TILE']))
    hip.tuning('gemm_splits', int(os.environ['GIST_GEMM_SPLITS']))
lay, m, n, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
torch.manual_seed(0)
if lay == 'nt':
    a, w, y = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), torch.empty(m, n, device=dev)
    f = lambda: hip.gemm_nt(a, w, None, y)
elif lay == 'nn':
    a, w, y = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev), torch.empty(m, n, device=dev)
    f = lambda: hip.gemm_nn(a, w, y)
else:
    a, w, y = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev), torch.empty(m, n, device=dev)
    f = lambda: hip.gemm_tn(a, w, y)
for _ in range(5):
    f()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ts.sort()
med = ts[len(ts) // 2]
print('%s %dx%dx%d tile=%s splits=%s: median %.1f us  %.1f TF' % (
    lay, m, n, k, os.environ.get('GIST_GEMM_TILE', 'auto'), os.environ.get('GIST_GEMM_SPLITS', 'auto'),
    med * 1e3, 2.0 * m * n * k / med / 1e9))
