"""Dev tool: row-split (plain) vs the blocked SpMM entry point (block-dense matrix-core kernel for
D >= 1536, LDS gather kernel below; scripts/spmm_mf_probe.py separates the two) on a Reddit-like
cluster batch; forward form (out_scale) and backward form (reversed CSR, src_scale, accumulate);
also with the batch rows randomly permuted (no block locality: every neighbour cross-block)."""
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
rp, cl = b.rowptr.cpu().numpy(), b.col[:nnz].cpu().numpy()
deg = np.diff(rp)
rows = np.repeat(np.arange(n), deg)
blk = np.searchsorted(rb.cpu().numpy(), np.arange(n), side='right') - 1
print('batch: n=%d nnz=%d max deg %d; neighbours inside the row\'s own part: %.1f%%'
      % (n, nnz, deg.max(), 100.0 * (blk[rows] == blk[cl]).mean()), flush=True)


def timeit(f, it_=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(it_):
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); c.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(c))
    ts.sort(); return ts[len(ts) // 2]


def run(tag, rowptr, col, t_rowptr, t_col, norm, blocks):
    for d in (512, 1024, 2048, 4096):
        z = torch.randn(n, 2 * d, device=dev)
        alg = 4.0 * (n + 1) + 4.0 * nnz + 8.0 * n * d
        out = {}
        for form in ('fwd', 'bwd'):
            if form == 'fwd':
                call = lambda **kw: hip.spmm(rowptr, col, z[:, :d], z[:, d:], out_scale=norm, **kw)
                res = lambda: z[:, d:]
            else:
                dz0 = torch.randn(n, 2 * d, device=dev)
                dz = dz0.clone()
                def call(**kw):
                    return hip.spmm(t_rowptr, t_col, dz[:, d:], dz[:, :d], src_scale=norm,
                                    accumulate=True, **kw)
                def res():
                    return dz[:, :d]
            def once(**kw):
                if form == 'bwd':
                    dz.copy_(dz0)
                call(**kw)
                return res().clone()
            ref = once()
            t0 = timeit(lambda: call())
            e2 = (once(row_blocks=blocks) - ref).abs().max().item()
            t2 = timeit(lambda: call(row_blocks=blocks))
            e3 = (once(blocked=True) - ref).abs().max().item()
            t3 = timeit(lambda: call(blocked=True))
            print('%s D=%4d %s: row-split %.1f us (%.0f GB/s) | blocked, blocks = parts %.1f us '
                  '(%.0f GB/s = %.3f of 8 TB/s) | blocked, uniform 128-row blocks %.1f us | max |diff| vs '
                  'row-split %.1e %.1e   [HIP-event time of one call from Python: includes ~15 us of launch path]'
                  % (tag, d, form, t0 * 1e3, alg / t0 / 1e6, t2 * 1e3, alg / t2 / 1e6,
                     alg / t2 / 1e6 / 8000, t3 * 1e3, e2, e3), flush=True)
        if d == 4096:
            hip.tuning('spmm_kernel', 1)
            for R in (1, 2, 3, 4):
                hip.tuning('spmm_split', R)
                t = timeit(lambda: hip.spmm(rowptr, col, z[:, :d], z[:, d:], out_scale=norm, row_blocks=blocks))
                print('   LDS gather kernel D=4096 fwd row_split=%d: %.1f us' % (R, t * 1e3), flush=True)
            hip.tuning('spmm_split', 0)
            hip.tuning('spmm_kernel', 0)


run('clustered', b.rowptr, b.col, b.t_rowptr, b.t_col, b.norm, rb)
# no locality: relabel the batch rows by a random permutation
perm = np.random.RandomState(0).permutation(n)
from gist_amd.graph import Graph
src = perm[cl]; dst = perm[rows]
gp = Graph.from_edges(src, dst, n).to(dev)
run('permuted ', gp.rowptr, gp.col, gp.t_rowptr, gp.t_col, gp.norm(), rb)
