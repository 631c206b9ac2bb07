"""Is the step host-bound? (dev tool) Enqueue time vs total time of the native step at the
per-rank widths of the multi-GPU points."""
import sys, os, random, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from gist_amd import datasets, hip
from gist_amd.engine import SageEngine, dims_for
from gist_amd.sampler import EngineClusterIter
DEV = torch.device('cuda', 0)
ds = datasets.reddit_synth(seed=0); g = ds.g
for H in (512, 1024, 4096):
    random.seed(0)
    it = EngineClusterIter(ds.name, g, len(ds.par_li), 20, np.arange(g.number_of_nodes(), dtype=np.int64), par_li=[p.copy() for p in ds.par_li], device=DEV)
    dims = dims_for(602, H, 41, 2)
    eng = SageEngine(dims, True, 0.2, it.n_max, DEV)
    rs = np.random.RandomState(0)
    eng.arena.load([(rs.uniform(-.03, .03, (o, 2 * i)).astype(np.float32), rs.uniform(-.03, .03, o).astype(np.float32)) for (i, o) in dims])
    it.bind(eng)
    def run(nsteps):
        k = 0
        while k < nsteps:
            for b in it:
                eng.train_step(b, 0.01, 0.0)
                k += 1
                if k >= nsteps: break
    run(40); torch.cuda.synchronize()
    t0 = time.time(); run(300); t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
    print('H=%d: host enqueue %.3f ms/step, total %.3f ms/step' % (H, (t1 - t0) / 300 * 1e3, (t2 - t0) / 300 * 1e3))
    hs = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.time(); run(4); t1 = time.time(); hs.append((t1 - t0) / 4 * 1e3)
    torch.cuda.synchronize()
    print('   host cost of issuing one step with an empty queue: median %.3f ms' % sorted(hs)[5])
