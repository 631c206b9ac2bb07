"""Dev tool: plain vs LDS-staged SpMM on a real Reddit-like cluster batch."""
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
n = b.n
sizes = [len(p) for p in it.par_li[:20]]
rb = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
assert int(rb[-1]) == n
nnz = int(b.rowptr[n].item())
def timeit(f, it_=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(it_):
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); c.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(c))
    ts.sort(); return ts[len(ts) // 2]
for d in (602, 256, 512, 1024, 2048, 4096):
    z = torch.randn(n, 2 * d, device=dev)
    t0 = timeit(lambda: hip.spmm(b.rowptr, b.col, z[:, :d], z[:, d:], out_scale=b.norm))
    ref = z[:, d:].clone()
    t1 = timeit(lambda: hip.spmm(b.rowptr, b.col, z[:, :d], z[:, d:], out_scale=b.norm, blocked=True))
    e1 = (z[:, d:] - ref).abs().max().item()
    t2 = timeit(lambda: hip.spmm(b.rowptr, b.col, z[:, :d], z[:, d:], out_scale=b.norm, row_blocks=rb))
    e2 = (z[:, d:] - ref).abs().max().item()
    alg = 4.0 * (n + 1) + 4.0 * nnz + 8.0 * n * d
    print('D=%4d plain %.1f us (%.0f GB/s alg) | lds uniform128 %.1f us | lds parts %.1f us (%.0f GB/s alg)  err %.1e %.1e'
          % (d, t0 * 1e3, alg / t0 / 1e6, t1 * 1e3, t2 * 1e3, alg / t2 / 1e6, e1, e2), flush=True)
