"""Dev tool: the prepared block-dense aggregation, one workgroup per CU (tuning hook spmm_kernel = 4) against two
(default) on a Reddit-like batch: bit equality and time per call (forward form, backward form), optionally the
column groups per block forced (spmm_split)."""
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
n, rb = b.n, b.row_blocks
prep, prep_t = hip.spmm_prepare(b.rowptr, b.col, rb), hip.spmm_prepare(b.t_rowptr, b.t_col, rb)


def timed(fn, reps=60):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


splits = [int(s) for s in os.environ.get('SPLITS', '0').split(',')]
for d in [int(x) for x in (sys.argv[1:] or ['4096'])]:
    x = torch.randn(n, 2 * d, device=dev)
    out = {}
    for kern in (0, 5):
        for sp in (splits if kern == 5 else [0]):
            hip.tuning('spmm_kernel', kern)
            hip.tuning('spmm_split', sp)
            z = x.clone()
            hip.spmm(b.rowptr, b.col, z[:, :d], z[:, d:], out_scale=b.norm, row_blocks=rb, prepared=prep)
            zf = z.clone()
            hip.spmm(b.t_rowptr, b.t_col, z[:, d:], z[:, :d], src_scale=b.norm, accumulate=True, row_blocks=rb, prepared=prep_t)
            tf = timed(lambda: hip.spmm(b.rowptr, b.col, z[:, :d], z[:, d:], out_scale=b.norm, row_blocks=rb, prepared=prep))
            tb = timed(lambda: hip.spmm(b.t_rowptr, b.t_col, z[:, d:], z[:, :d], src_scale=b.norm, row_blocks=rb, prepared=prep_t))
            tba = timed(lambda: hip.spmm(b.t_rowptr, b.t_col, x[:, d:], z[:, :d], src_scale=b.norm, accumulate=True, row_blocks=rb, prepared=prep_t))
            out[(kern, sp)] = zf
            alg = 4.0 * (n + 1) + 4.0 * int(b.rowptr[-1]) + 8.0 * n * d
            print('D %d kernel %s groups %s: fwd %.1f us (%.2f of 8 TB/s)  bwd %.1f  bwd accumulate %.1f' %
                  (d, 'one/CU' if kern == 0 else 'producer/consumer', sp or 'auto', tf, alg / tf / 8e6, tb, tba), flush=True)
    ref = out[(0, 0)]
    for k, v in out.items():
        if k != (0, 0):
            print('   bit-equal to one/CU:', bool(torch.equal(v, ref)), k)
hip.tuning('spmm_kernel', 0)
hip.tuning('spmm_split', 0)
print('blocks', int(rb.numel()) - 1, 'rows', n)
