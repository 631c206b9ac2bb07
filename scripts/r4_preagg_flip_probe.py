"""Round 4 probe: the fused step with and without layer 0's aggregation formed by the extraction (GIST_STEP_PREAGG), metric
configuration, one iteration -- how many rows of dW_0 differ beyond rounding (a ReLU input within rounding of zero flips)."""
import os, random, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gist_amd import datasets, hip
from gist_amd.engine import SageEngine, dims_for
from gist_amd.sampler import EngineClusterIter
DEV = torch.device('cuda:0')
hip.gemm_mode('bf16x3')
runs = {}
for pre in ('0', '1'):
    os.environ['GIST_STEP_PREAGG'] = pre
    ds = datasets.reddit_synth(seed=0)
    g = ds.g
    random.seed(3)
    it = EngineClusterIter(ds.name, g, len(ds.par_li), 20, np.arange(g.number_of_nodes(), dtype=np.int64),
                           par_li=[p.copy() for p in ds.par_li], device=DEV)
    dims = dims_for(602, 4096, 41, 2)
    eng = SageEngine(dims, True, 0.2, it.n_max, DEV)
    rs = np.random.RandomState(3)
    params = []
    for (i, o) in dims:
        sc = 1.0 / np.sqrt(2 * i)
        params.append((rs.uniform(-sc, sc, (o, 2 * i)).astype(np.float32), rs.uniform(-sc, sc, o).astype(np.float32)))
    eng.arena.load(params)
    it.bind(eng)
    batch = next(iter(it))
    loss = float(eng.train_step(batch, 0.01, 0.0).item())
    torch.cuda.synchronize()
    runs[pre] = (loss, [w.clone() for w in eng.arena.dW], eng.Z[0][:batch.n].clone())
l0, l1 = runs['0'][0], runs['1'][0]
print('loss', l0, l1, abs(l0 - l1))
z0, z1 = runs['0'][2], runs['1'][2]
print('Z0 max |diff| / max', (z0 - z1).abs().max().item(), z0.abs().max().item())
for k in range(3):
    a, b = runs['0'][1][k], runs['1'][1][k]
    d = (a - b).abs()
    rowmax = d.max(dim=1).values
    bar = 1e-4 * a.abs().max().item()
    print('dW_%d: max diff %.3e  (max |g| %.3e), mean diff %.3e, rows over 1e-4 of max: %d of %d, cols over: %d of %d' % (
        k, d.max().item(), a.abs().max().item(), d.mean().item(), int((rowmax > bar).sum().item()), a.shape[0],
        int((d.max(dim=0).values > bar).sum().item()), a.shape[1]))
