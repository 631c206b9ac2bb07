"""Round 4 probe: what the shared optimiser + extraction grid (adam_extract_kernel) costs against its two halves, at the
per-rank width 512 of the metric's batches.  Back-to-back launches, HIP events, microseconds per launch:
  adam        gist_adam_segments_f32 over the whole arena (no deferred segments)
  extract     gist_extract_parts_desc_batch (features, CSR both ways, layer 0's aggregation)
  fused       gist_adam_segments_extract_f32 = both in one grid
  fused_tiny  the same grid with a 4096-element arena: the extraction beside (almost) no optimiser traffic
"""
import ctypes, os, random, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gist_amd import datasets, hip, _lib
from gist_amd.engine import SageEngine, dims_for
from gist_amd.sampler import EngineClusterIter
DEV = torch.device('cuda:0')
L = _lib.load()
H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ds = datasets.reddit_synth(seed=0)
g = ds.g
random.seed(3)
it = EngineClusterIter(ds.name, g, len(ds.par_li), 20, np.arange(g.number_of_nodes(), dtype=np.int64),
                       par_li=[p.copy() for p in ds.par_li], device=DEV)
dims = dims_for(602, H, 41, 2)
eng = SageEngine(dims, True, 0.2, it.n_max, DEV)
it.bind(eng)
batches = iter(it)
b0 = next(batches)
eng.train_step(b0, 0.01, 5e-4)          # (sets up scratch, plan tables)
b1 = next(batches)
torch.cuda.synchronize()
A, P = eng.arena, eng.plan
n = A.params.numel()
row_loss = torch.rand(b0.n, device=DEV)
loss = torch.zeros(1, device=DEV)
grad = torch.randn(n, device=DEV) * 1e-3


def desc():
    x = _lib.ExtractPartsDesc()
    for f in ('g_rowptr', 'g_col', 'g_t_rowptr', 'g_t_col', 'node_part', 'part_slot', 'rowptr', 'col', 't_rowptr', 't_col',
              'col_capacity', 'norm', 'feat', 'ld_feat', 'labels_all', 'labels'):
        setattr(x, f, getattr(P, f))
    x.ids, x.n, x.n_max, x.batch = b1.ids.data_ptr(), b1.ids.numel(), P.n_max, int(b1.parts[2])
    x.n_feat, x.z0, x.ldz0 = 602, P.layer[0].Z, P.layer[0].ldz
    x.x0, x.ldx0, x.p, x.seed, x.offset, x.mask_ld = P.hsrc[0], P.ld_hsrc[0], 0.2, P.seed, 1000, 2 * 602
    x.scratch = P.extract_scratch
    x.feat_intra, x.ld_intra = P.feat_intra, P.ld_feat_intra
    x.ah = P.layer[0].Z + 602 * 4
    return x


def adam_args(count):
    return (A.params.data_ptr(), grad.data_ptr(), A.exp_avg.data_ptr(), A.exp_avg_sq.data_ptr(), count, 0.01, 0.9, 0.999, 1e-8,
            5e-4, 7, (_lib.GradSegment * 1)(), 0, row_loss.data_ptr(), b0.n, b0.n, loss.data_ptr())


def timeit(f, iters=200):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


x = desc()
st = hip._stream()
res = {
    'adam': timeit(lambda: _lib.check(L.gist_adam_segments_f32(*adam_args(n), st), 'adam')),
    'adam_tiny': timeit(lambda: _lib.check(L.gist_adam_segments_f32(*adam_args(4096), st), 'adam')),
    'extract': timeit(lambda: _lib.check(L.gist_extract_parts_desc_batch(ctypes.byref(x), st), 'extract')),
    'fused': timeit(lambda: _lib.check(L.gist_adam_segments_extract_f32(*adam_args(n), ctypes.byref(x), st), 'fused')),
    'fused_tiny': timeit(lambda: _lib.check(L.gist_adam_segments_extract_f32(*adam_args(4096), ctypes.byref(x), st), 'fused')),
}
eng.check_extract()
print('n_hidden', H, 'parameters', n, 'batch rows', b1.ids.numel())
for k, v in res.items():
    print('%-11s %7.2f us' % (k, v))
