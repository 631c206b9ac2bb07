"""Dev probe (round 6): what does a batch extraction cost the training step when it runs on a SECOND stream beside it?

Sizes the double-buffered-batch idea before building it: if the batch buffers existed twice, the extraction of batch t + 1
would depend on nothing of step t and could run on a side stream while step t computes (events already signalled when
they are waited for), and the optimiser launch would stop carrying it (adam_extract_kernel 32 us -> ~10 at h = 512).
Here the step is the product's (its own fused extraction stays), and ONE MORE extraction of the same batch -- into the
buffers of a second iterator -- is issued per step: mode `main` on the step's stream (its full cost), mode `side` on a
second stream with the event pattern the real thing would use, mode `none` not at all.  side - none = what overlap leaves.

python scripts/r6_side_stream_probe.py <n_hidden> [steps]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')
import numpy as np, torch
from gist_amd import datasets, hip
from gist_amd.engine import SageEngine, dims_for
from gist_amd.sampler import EngineClusterIter

H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
hip.gemm_mode('bf16x3')
ds = datasets.reddit_synth(seed=0)
g = ds.g
nid = np.arange(g.number_of_nodes(), dtype=np.int64)


def make(native, hidden):
    random.seed(0)
    it = EngineClusterIter('r', g, len(ds.par_li), 20, nid, par_li=[p.copy() for p in ds.par_li], device=dev)
    eng = SageEngine(dims_for(602, hidden, 41, 2), True, 0.2, it.n_max, dev, seed=0)
    rs = np.random.RandomState(0)
    for k, (i, o) in enumerate(eng.dims):
        s = 1.0 / np.sqrt(2 * i)
        eng.arena.W[k].copy_(torch.from_numpy(rs.uniform(-s, s, (o, 2 * i)).astype(np.float32)))
        eng.arena.b[k].copy_(torch.from_numpy(rs.uniform(-s, s, o).astype(np.float32)))
    it.bind(eng, native=native)
    return it, eng


it, eng = make(True, H)
eng.prefetch = True
it2, eng2 = make(False, 64)          # the second set of batch buffers: its iterator extracts eagerly (one launch)


def batches(i):
    while True:
        for b in i:
            yield b


def run(mode, steps):
    gen, gen2 = batches(it), batches(it2)
    side = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    ev_side, ev_main = torch.cuda.Event(), torch.cuda.Event()
    for _ in range(20):
        eng.train_step(next(gen), 0.01, 0.0)
    torch.cuda.synchronize()
    t0 = time.time()
    for s in range(steps):
        if mode == 'main':
            next(gen2)
        elif mode == 'events':                 # the event traffic alone
            ev_main.record(main)
            side.wait_event(ev_main)
            ev_side.record(side)
        elif mode == 'side':
            ev_main.record(main)               # (what the other buffer set's last reader would have recorded: long done)
            side.wait_event(ev_main)
            with torch.cuda.stream(side):
                next(gen2)
            ev_side.record(side)
        eng.train_step(next(gen), 0.01, 0.0)
        if mode in ('side', 'events'):
            main.wait_event(ev_side)           # the next step reads what the side stream extracted
    t_issue = time.time() - t0
    torch.cuda.synchronize()
    return (time.time() - t0) / steps * 1e3, t_issue / steps * 1e3


for rep in range(2):
    for mode in ('none', 'main', 'events', 'side'):
        print('h = %d  %-5s %.4f ms/step (host issue %.4f)' % ((H, mode) + run(mode, STEPS)), flush=True)
