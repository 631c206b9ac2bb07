"""Dev tool (run under rocprofv3 --pmc WRITE_SIZE, and FETCH_SIZE in a second pass): what the block-dense aggregation
writes.  One dispatch each, D = 4096, on a Reddit-like cluster batch:
  0 forward form into a dense y (pitch D)          1 forward form into the right half of Z (pitch 2 D)
  2 backward form (y += ..., pitch 2 D)            3 as 1 with uniform 128-row blocks instead of the parts
The rocprof CSV lists them in this order after the warm-up dispatches (4 of them)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gist_amd import datasets, hip
from gist_amd.engine import SageEngine, dims_for
from gist_amd.sampler import EngineClusterIter
dev = torch.device('cuda', 0)
random.seed(0)
ds = datasets.reddit_synth(seed=0)
g = ds.g
it = EngineClusterIter('r', g, len(ds.par_li), 20, np.arange(g.number_of_nodes(), dtype=np.int64),
                       par_li=[p.copy() for p in ds.par_li], device=dev)
eng = SageEngine(dims_for(602, 64, 41, 1), True, 0.0, it.n_max, dev)
it.bind(eng, native=False)
b = next(iter(it))
n, d = b.n, 4096
sizes = [len(p) for p in it.par_li[:20]]
rb = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
ub = torch.tensor(list(range(0, n, 128)) + [n], dtype=torch.int32, device=dev)
x = torch.randn(n, d, device=dev)
y = torch.empty(n, d, device=dev)
z = torch.randn(n, 2 * d, device=dev)
dz = torch.randn(n, 2 * d, device=dev)
calls = [lambda: hip.spmm(b.rowptr, b.col, x, y, out_scale=b.norm, row_blocks=rb),
         lambda: hip.spmm(b.rowptr, b.col, z[:, :d], z[:, d:], out_scale=b.norm, row_blocks=rb),
         lambda: hip.spmm(b.t_rowptr, b.t_col, dz[:, d:], dz[:, :d], src_scale=b.norm, accumulate=True, row_blocks=rb),
         lambda: hip.spmm(b.rowptr, b.col, z[:, :d], z[:, d:], out_scale=b.norm, row_blocks=ub)]
for c in calls:
    c()
torch.cuda.synchronize()
for c in calls:
    c()
    torch.cuda.synchronize()
print('n=%d nnz=%d output bytes %.1f MB; rows per part: min %d max %d' % (n, int(b.rowptr[n]), n * d * 4 / 1e6, min(sizes), max(sizes)))
