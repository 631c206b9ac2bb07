"""Dev probe (round 5): the blocked aggregation of one Reddit-like batch at the narrow widths on each kernel the launcher
can be forced onto (GIST_TUNE_SPMM_KERNEL: 1 = LDS gather, 2 = block-dense bf16x3, 3 = fp32 block-dense from memory), forward
and reversed-accumulate forms, 200 back-to-back launches between two events (launch-to-launch time, not a single launch's)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gist_amd import datasets, hip
from gist_amd.engine import SageEngine, dims_for
from gist_amd.sampler import EngineClusterIter
dev = torch.device('cuda', 0)
random.seed(0)
ds = datasets.reddit_synth(seed=0)
it = EngineClusterIter('r', ds.g, len(ds.par_li), 20, np.arange(ds.g.number_of_nodes(), dtype=np.int64),
                       par_li=[p.copy() for p in ds.par_li], device=dev)
eng = SageEngine(dims_for(602, 64, 41, 1), True, 0.0, it.n_max, dev)
it.bind(eng, native=False)
b = next(iter(it))
n = b.n


def stream_time(f, reps=200):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    c.record(); torch.cuda.synchronize()
    return a.elapsed_time(c) / reps * 1e3


prep_f = hip.spmm_prepare(b.rowptr, b.col, b.row_blocks)
prep_b = hip.spmm_prepare(b.t_rowptr, b.t_col, b.row_blocks)
for d in (256, 512, 1024):
    z = torch.randn(n, 2 * d, device=dev)
    dz = torch.randn(n, 2 * d, device=dev)
    line = 'D=%d:' % d
    for kern, name in ((1, 'lds'), (2, 'bf16x3 dense'), (3, 'fp32 dense')):
        hip.tuning('spmm_kernel', kern)
        try:
            tf = stream_time(lambda: hip.spmm(b.rowptr, b.col, z[:, :d], z[:, d:], out_scale=b.norm, row_blocks=b.row_blocks, prepared=prep_f))
            tb = stream_time(lambda: hip.spmm(b.t_rowptr, b.t_col, dz[:, d:], dz[:, :d], src_scale=b.norm, accumulate=True, row_blocks=b.row_blocks, prepared=prep_b))
            line += '  %s fwd %.1f bwd %.1f us' % (name, tf, tb)
        except Exception as e:
            line += '  %s: %s' % (name, str(e)[:60])
    hip.tuning('spmm_kernel', 0)
    print(line, flush=True)
