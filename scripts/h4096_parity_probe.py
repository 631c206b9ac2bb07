"""H=4096 step parity probe (dev tool): GPU step in both GEMM modes and the CPU oracle against a
float64 autograd evaluation of the same step on the GPU (dense adjacency)."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gist_amd import datasets, hip
from gist_amd.engine import SageEngine, dims_for
from gist_amd.sampler import EngineClusterIter
from oracle import gist_oracle as O
from oracle import train_oracle as TO
DEV = torch.device('cuda', 0)


def step64(rowptr, col, feat, labels, params):
    n = len(rowptr) - 1
    A = torch.zeros(n, n, dtype=torch.float64, device=DEV)
    rows = torch.repeat_interleave(torch.arange(n, device=DEV), torch.from_numpy(np.diff(rowptr)).to(DEV))
    A.index_put_((rows, torch.from_numpy(col).to(DEV)), torch.ones(len(col), dtype=torch.float64, device=DEV), accumulate=True)
    deg = A.sum(1, keepdim=True)
    A = torch.where(deg > 0, A / deg.clamp(min=1), torch.zeros_like(A))
    ps = [(torch.from_numpy(W).to(DEV).double().requires_grad_(), torch.from_numpy(b).to(DEV).double().requires_grad_()) for W, b in params]
    h = torch.from_numpy(feat).to(DEV).double()
    ys, yhats = [], []
    for k, (W, b) in enumerate(ps):
        y = torch.cat([h, A @ h], 1) @ W.t() + b
        y.retain_grad(); ys.append(y)
        if k + 1 < len(ps):
            y = torch.nn.functional.layer_norm(y, (y.shape[1],), eps=1e-5)
            yhats.append(y.detach())
            h = torch.relu(y)
    loss = torch.nn.functional.cross_entropy(y, torch.from_numpy(labels).to(DEV).long())
    loss.backward()
    return loss.item(), y.detach(), [(W.grad, b.grad) for W, b in ps], yhats, [t.grad for t in ys]


def rel(a, b):
    return (torch.linalg.norm(a.double() - b) / torch.linalg.norm(b)).item()


ds = datasets.reddit_synth(seed=0); g = ds.g
tg = TO.TrainGraph(g.rowptr.numpy().astype(np.int64), g.col.numpy().astype(np.int64), g.ndata['feat'].numpy(), g.ndata['label'].numpy().astype(np.int64))
dims = dims_for(602, 4096, 41, 2)
for mode in ('f32', 'f16x3'):
    hip.gemm_mode(mode)
    random.seed(3)
    it = EngineClusterIter(ds.name, g, len(ds.par_li), 20, np.arange(g.number_of_nodes(), dtype=np.int64), par_li=[p.copy() for p in ds.par_li], device=DEV)
    eng = SageEngine(dims, True, 0.0, it.n_max, DEV)
    rs = np.random.RandomState(3)
    params = []
    for (i, o) in dims:
        sc = 1.0 / np.sqrt(2 * i)
        params.append((rs.uniform(-sc, sc, (o, 2 * i)).astype(np.float32), rs.uniform(-sc, sc, o).astype(np.float32)))
    eng.arena.load(params); it.bind(eng, native=False)
    batch = next(iter(it))
    b = tg.batch(it.batch_ids(0))
    l64, y64, g64, yh64, dy64 = step64(b[0], b[1], b[4], b[5], params)
    n = batch.n
    eng.forward(batch, True)
    for k in range(2):
        yh = eng.Y[k][:n].double()
        d = (yh - yh64[k]).abs()
        flips = ((yh > 0) != (yh64[k] > 0))
        print('   yhat%d: max err %.2e rms %.2e  relu sign flips %d  (sum |dy64| at flips %.3e of total %.3e)' % (k, d.max().item(), d.pow(2).mean().sqrt().item(), int(flips.sum().item()), 0.0, 0.0))
    loss = eng.loss_and_backward(batch).clone()
    for k in range(2):
        print('   dY%d fro rel %.2e' % (k, rel(eng.Y[k][:n], dy64[k])))
    logits = eng.logits(batch.n)
    print(mode, 'loss err vs f64 %.2e  logits max err %.2e (max %.2f)' % (abs(loss.item() - l64), (logits.double() - y64).abs().max().item(), y64.abs().max().item()))
    for k in range(len(dims)):
        print('   dW%d fro rel %.2e  db%d %.2e' % (k, rel(eng.arena.dW[k], g64[k][0]), k, rel(eng.arena.db[k], g64[k][1])))
    if mode == 'f32':
        opt = O.new_opt_state(params)
        pc = [(W.copy(), bb.copy()) for W, bb in params]
        ol, ologits, og = O.train_step(b[0], b[1], b[2], b[3], b[4], b[5], pc, opt, True, 0.01)
        print('oracle loss err vs f64 %.2e  logits max err %.2e' % (abs(float(ol) - l64), np.abs(ologits - y64.cpu().numpy()).max()))
        for k in range(len(dims)):
            print('   dW%d fro rel %.2e  db%d %.2e' % (k, rel(torch.from_numpy(og[k][0]).to(DEV), g64[k][0]), k, rel(torch.from_numpy(og[k][1]).to(DEV), g64[k][1])))
