"""Dev tool: the two kernels behind gist_spmm_csr_blocked_f32 (tuning hook spmm_kernel: 1 = LDS
gather, otherwise block-dense MFMA) on a Reddit-like batch at the given widths; run under
scripts/ktrace.sh for kernel durations (KFILTER=spmm)."""
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
rb = b.row_blocks
for d in [int(x) for x in (sys.argv[1:] or ['4096'])]:
    z = torch.randn(n, 2 * d, device=dev)
    prep, prep_t = hip.spmm_prepare(b.rowptr, b.col, rb), hip.spmm_prepare(b.t_rowptr, b.t_col, rb)
    for kern in (1, 2, 3):
        hip.tuning("spmm_kernel", min(kern, 2))
        for _ in range(12):
            hip.spmm(b.rowptr, b.col, z[:, :d], z[:, d:], out_scale=b.norm, row_blocks=rb, prepared=prep if kern == 3 else None)
            hip.spmm(b.t_rowptr, b.t_col, z[:, d:], z[:, :d], src_scale=b.norm, accumulate=True, row_blocks=rb,
                     prepared=prep_t if kern == 3 else None)
        torch.cuda.synchronize()
hip.tuning('spmm_kernel', 0)
print('blocks', int(rb.numel()) - 1, 'rows', n)
