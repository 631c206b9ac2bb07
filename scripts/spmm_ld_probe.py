"""Dev tool: sensitivity of the SpMM to the source matrix's leading dimension / residency."""
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
b = next(iter(it)); n = b.n
def timeit(f, it_=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(it_):
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); c.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(c))
    ts.sort(); return ts[len(ts) // 2]
d = 4096
z = torch.randn(n, 2 * d, device=dev)
for ldx in (4096, 4096 + 64, 4096+256, 6144, 8192, 8192 + 64):
    x = torch.randn(n, ldx, device=dev)
    t = timeit(lambda: hip.spmm(b.rowptr, b.col, x[:, :d], z[:, d:], out_scale=b.norm))
    print('x ld %5d -> z right: %.1f us' % (ldx, t * 1e3), flush=True)
t = timeit(lambda: hip.spmm(b.rowptr, b.col, z[:, :d], z[:, d:], out_scale=b.norm))
print('z left -> z right: %.1f us' % (t * 1e3))
y = torch.empty(n, d, device=dev)
t = timeit(lambda: hip.spmm(b.rowptr, b.col, z[:, :d], y, out_scale=b.norm))
print('z left -> y ld 4096: %.1f us' % (t * 1e3))
