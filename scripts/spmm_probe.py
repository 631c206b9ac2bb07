"""Dev tool: time gist_spmm_csr_blocked_f32 (second LDS design) at one width on a Reddit-like
batch with the library named by GIST_LIB_PATH (ablation builds: GIST_EXTRA_FLAGS=-DL2_PROBE_*)."""
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
def timeit(f, it_=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(it_):
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); c.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(c))
    ts.sort(); return ts[len(ts) // 2]
out = []
for d in [int(x) for x in (sys.argv[1:] or ['4096'])]:
    z = torch.randn(n, 2 * d, device=dev)
    for R in (0, 1, 2, 3, 4, 6):
        T = 1
        hip.tuning('spmm_split', R)
        t = timeit(lambda: hip.spmm(b.rowptr, b.col, z[:, :d], z[:, d:], out_scale=b.norm, row_blocks=rb), 10)
        out.append('D=%d T=%d R=%d %.1f us' % (d, T, R, t * 1e3))
print(os.environ.get('GIST_LIB_PATH', 'default'), ' | '.join(out), flush=True)
