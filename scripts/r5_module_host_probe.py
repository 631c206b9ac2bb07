"""Host cost of one iteration of the module path: the loop issues N steps, the host clock is read BEFORE the final
synchronise (issue time) and after it (wall).  --profile prints the cProfile of the issue loop."""
import argparse
import random
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, '.')
ap = argparse.ArgumentParser()
ap.add_argument('--n-hidden', type=int, default=512)
ap.add_argument('--n-layers', type=int, default=2)
ap.add_argument('--steps', type=int, default=300)
ap.add_argument('--profile', action='store_true')
ap.add_argument('--engine', action='store_true', help='the engine path instead (SageEngine.train_step)')
a = ap.parse_args()
from gist_amd import datasets
from gist_amd.modules import GCN
from gist_amd.nn import CrossEntropyLoss
from gist_amd.optim import Adam
from gist_amd.sampler import ClusterIter, EngineClusterIter
from gist_amd.engine import SageEngine, dims_for
dev = torch.device('cuda:0')
ds = datasets.reddit_synth() if hasattr(datasets, 'reddit_synth') else datasets.load('reddit-synth')
g = ds.g
nid = np.arange(g.number_of_nodes(), dtype=np.int64)
random.seed(0)
F_in, C = g.ndata['feat'].shape[1], ds.num_classes
if a.engine:
    it = EngineClusterIter(ds.name, g, len(ds.par_li), 20, nid, par_li=[p.copy() for p in ds.par_li], device=dev)
    eng = SageEngine(dims_for(F_in, a.n_hidden, C, a.n_layers), True, 0.2, it.n_max, dev)
    for W in eng.arena.W:
        W.uniform_(-0.03, 0.03)
    it.bind(eng)
    eng.prefetch = True

    def batches():
        while True:
            for b in it:
                yield b
    gen = batches()

    def loop(n):
        for _ in range(n):
            eng.train_step(next(gen), 0.01, 0.0)
else:
    it = ClusterIter(ds.name, g, len(ds.par_li), 20, nid, par_li=[p.copy() for p in ds.par_li], device=dev)
    model = GCN(F_in, a.n_hidden, C, a.n_layers, F.relu, 0.2, True, False, False, 1, True).cuda()
    loss_f, opt = CrossEntropyLoss(), Adam(model.parameters(), lr=0.01)

    def batches():
        while True:
            for c in it:
                yield c
    gen = batches()

    def loop(n):
        for _ in range(n):
            cluster = next(gen)
            cluster = cluster.to(torch.cuda.current_device())
            model.train()
            pred = model(cluster)
            lab = cluster.ndata['label']
            m = cluster.ndata['train_mask']
            loss = loss_f(pred[m], lab[m])
            opt.zero_grad()
            loss.backward()
            opt.step()
loop(20)
torch.cuda.synchronize()
import gc
for chunk in range(8):
    st0 = torch.cuda.memory_stats().get('num_device_alloc', 0)
    g0 = gc.get_stats()[2]['collections']
    t0 = time.perf_counter()
    loop(50)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print('chunk %d: issue %.1f us/step, new device allocs %d, gen2 collections %d' % (
        chunk, (t1 - t0) / 50 * 1e6, torch.cuda.memory_stats().get('num_device_alloc', 0) - st0, gc.get_stats()[2]['collections'] - g0), flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    loop(a.steps)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%s h=%d L=%d: issue %.1f us/step, wall %.1f us/step' % ('engine' if a.engine else 'module', a.n_hidden, a.n_layers,
                                                                   (t1 - t0) / a.steps * 1e6, (t2 - t0) / a.steps * 1e6), flush=True)
if a.profile:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    loop(a.steps)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
if not a.engine:
    st = torch.cuda.memory_stats()
    print('device allocs', st.get('num_device_alloc'), 'alloc retries', st.get('num_alloc_retries'))
