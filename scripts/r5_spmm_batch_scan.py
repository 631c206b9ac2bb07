"""Dev probe (round 5): the blocked aggregation at D = 4096 and the one-launch extraction, batch by batch over an epoch of the
power-law community graph cut by gist_partition_graph: which batches are slow, and what they have in common."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gist_amd import datasets, hip
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
if len(sys.argv) > 1 and sys.argv[1] == 'renumber':      # node ids contiguous inside a part
    order = np.concatenate(parts)
    gd = dc.g.to(dev).subgraph(order).to('cpu')
    st = np.concatenate([[0], np.cumsum([len(p) for p in parts])])
    parts = [np.arange(st[i], st[i + 1], dtype=np.int64) for i in range(k)]
    dc = dc._replace(g=gd)
random.seed(0)
it = EngineClusterIter('r', dc.g, k, 20, np.arange(dc.g.number_of_nodes(), dtype=np.int64), par_li=parts, device=dev)
eng = SageEngine(dims_for(602, 64, 41, 1), True, 0.0, it.n_max, dev)
it.bind(eng, native=False)


def timeit(f, it_=8):
    for _ in range(2): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(it_):
        x, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x.record(); f(); c.record(); torch.cuda.synchronize(); ts.append(x.elapsed_time(c))
    ts.sort(); return ts[len(ts) // 2] * 1e3


d = 4096
z = torch.randn(it.n_max, 2 * d, device=dev)
for j, b in enumerate(it):
    if j >= 40:
        break
    n = b.n
    nnz = int(b.rowptr[n].item())
    rp, cl = b.rowptr.cpu().numpy().astype(np.int64), b.col[:nnz].cpu().numpy()
    rb = b.row_blocks.cpu().numpy()
    deg = np.diff(rp)
    rows = np.repeat(np.arange(n), deg)
    blk = np.searchsorted(rb, np.arange(n), side='right') - 1
    same = blk[rows] == blk[cl]
    opr = np.bincount(rows[~same], minlength=n)
    over = np.flatnonzero(opr > 8)
    prep = hip.spmm_prepare(b.rowptr, b.col, b.row_blocks)
    zz = z[:n]
    t = timeit(lambda: hip.spmm(b.rowptr, b.col, zz[:, :d], zz[:, d:], out_scale=b.norm, row_blocks=b.row_blocks, prepared=prep))
    ids = it._epoch_ids[int(it._offsets[j]):int(it._offsets[j + 1])]
    bt = it.batcher
    f_ = bt.feat.shape[1]
    te = timeit(lambda: hip.extract_parts(bt.g, ids, it.n_max, it._node_part, it._part_tables, j, bt.rowptr[:n + 1], bt.col,
                                          bt.t_rowptr[:n + 1], bt.t_col, bt.norm, bt.feat, eng.z0_left(n), bt.labels, bt.lab,
                                          it._extract_scratch, feat_intra=bt.feat_intra, ah=eng.Z[0][:n, f_:2 * f_]))
    te0 = timeit(lambda: hip.extract_parts(bt.g, ids, it.n_max, it._node_part, it._part_tables, j, bt.rowptr[:n + 1], bt.col,
                                           bt.t_rowptr[:n + 1], bt.t_col, bt.norm, bt.feat, eng.z0_left(n), bt.labels, bt.lab,
                                           it._extract_scratch))
    trp = b.t_rowptr.cpu().numpy().astype(np.int64)
    tdeg = np.diff(trp)
    fd = np.diff(dc.g.rowptr.numpy().astype(np.int64))[ids.cpu().numpy()]
    print('batch %2d n=%d nnz=%6d blocks %d max deg %4d | rows >8 outside: %2d (their degrees %s) max outside %3d | full-graph degree max %5d | spmm %.1f us, extraction %.1f us'
          % (j, n, nnz, len(rb) - 1, deg.max(), over.size, list(deg[over][:4]), opr.max(), fd.max(), t, te), 'without aggregation %.1f us; rows with > 256 kept in/out-neighbours %d/%d, > 64 remote %d' % (te0, (deg > 256).sum(), (tdeg > 256).sum(), (opr > 64).sum()), flush=True)
