"""Dev probe (round 6): the prepared block-dense aggregation at D = 4096, batch by batch over an epoch of the power-law
community graph cut by gist_partition_graph: per batch the pairs the prepare kernel found, the rows in each state of its
per-row lists, the part-to-part edge counts, and the launch time (main + pairs kernels together)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gist_amd import datasets, hip, _lib
from gist_amd.engine import SageEngine, dims_for
from gist_amd.sampler import EngineClusterIter
from gist_amd.dgl_compat.transform import partition_assignment
dev = torch.device('cuda', 0)
dc = datasets.reddit_communities(seed=0)
k = 1500
a = partition_assignment(dc.g, k, seed=0)
o = np.argsort(a, kind='stable')
bnd = np.searchsorted(a[o], np.arange(k + 1))
parts = [o[bnd[i]:bnd[i + 1]].astype(np.int64) for i in range(k)]
random.seed(0)
it = EngineClusterIter('r', dc.g, k, 20, np.arange(dc.g.number_of_nodes(), dtype=np.int64), par_li=parts, device=dev)
eng = SageEngine(dims_for(602, 64, 41, 1), True, 0.0, it.n_max, dev)
it.bind(eng, native=False)
stride = int(_lib.load().gist_spmm_block_image_bytes())


def timeit(f, it_=8):
    for _ in range(2): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(it_):
        x, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x.record(); f(); c.record(); torch.cuda.synchronize(); ts.append(x.elapsed_time(c))
    ts.sort(); return ts[len(ts) // 2] * 1e3


d = 4096
z = torch.randn(it.n_max, 2 * d, device=dev)
tot = []
for j, b in enumerate(it):
    if j >= int(os.environ.get('BATCHES', '40')):
        break
    n = b.n
    nnz = int(b.rowptr[n].item())
    rp, cl = b.rowptr.cpu().numpy().astype(np.int64), b.col[:nnz].cpu().numpy()
    rb = b.row_blocks.cpu().numpy()
    nb = len(rb) - 1
    rows = np.repeat(np.arange(n), np.diff(rp))
    blk = np.searchsorted(rb, np.arange(n), side='right') - 1
    cross = blk[rows] != blk[cl]
    pc = np.bincount(blk[rows][cross] * nb + blk[cl][cross], minlength=nb * nb).reshape(nb, nb)
    top = np.sort(pc.ravel())[::-1][:6]
    prep = hip.spmm_prepare(b.rowptr, b.col, b.row_blocks)
    rec = prep[:nb * stride].view(nb, stride)
    pinfo = rec[:, stride - 16:].contiguous().view(torch.int32).view(nb, 4).cpu().numpy()
    remc = rec[:, 32768:32768 + 512].contiguous().view(torch.int32).view(nb, 128).cpu().numpy()
    n_pairs = int((pinfo[:, 1] > 0).sum() + (pinfo[:, 3] > 0).sum())
    zz = z[:n]
    t = timeit(lambda: hip.spmm(b.rowptr, b.col, zz[:, :d], zz[:, d:], out_scale=b.norm, row_blocks=b.row_blocks, prepared=prep))
    tot.append(t)
    print('batch %2d n=%d nnz=%6d blocks %d | host flag %s | blocks with pairs %d, pairs %d | rows: walk %3d, full %3d, listed>4 %3d | largest part-to-part counts %s | %.1f us'
          % (j, n, nnz, nb, b.siblings, int((pinfo[:, 1] > 0).sum()), n_pairs, int((remc == -2).sum()), int((remc == -1).sum()),
             int(((remc >= 0) & ((remc & 0xff) > 4)).sum()), list(top), t), flush=True)
print('mean %.1f us, median %.1f us' % (np.mean(tot), np.median(tot)))
