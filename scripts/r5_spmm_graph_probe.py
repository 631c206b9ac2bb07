"""Dev probe (round 5): the blocked aggregation (prepared block-dense kernel, D = 4096; LDS kernel, D = 512) and the one-launch
extraction on ONE batch of (a) the planted block model, (b) the power-law community graph cut by gist_partition_graph, (c) the
same with its nodes renumbered part by part, (d) the community graph without hubs: which property of the batch costs the time."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gist_amd import datasets, hip
from gist_amd.engine import SageEngine, dims_for
from gist_amd.sampler import EngineClusterIter
from gist_amd.dgl_compat.transform import partition_assignment
dev = torch.device('cuda', 0)


def own_parts(ds, k=1500):
    a = partition_assignment(ds.g, k, seed=0)
    o = np.argsort(a, kind='stable')
    b = np.searchsorted(a[o], np.arange(k + 1))
    return [o[b[i]:b[i + 1]].astype(np.int64) for i in range(k)]


def timeit(f, it_=40):
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(it_):
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); c.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(c))
    ts.sort(); return ts[len(ts) // 2] * 1e3


def probe(tag, ds, parts):
    random.seed(0)
    g = ds.g
    it = EngineClusterIter('r', g, len(parts), 20, np.arange(g.number_of_nodes(), dtype=np.int64),
                           par_li=[p.copy() for p in parts], device=dev)
    eng = SageEngine(dims_for(602, 64, 41, 1), True, 0.0, it.n_max, dev)
    it.bind(eng, native=False)
    b = next(iter(it))
    n = b.n
    nnz = int(b.rowptr[n].item())
    rp, cl = b.rowptr.cpu().numpy().astype(np.int64), b.col[:nnz].cpu().numpy()
    rb = b.row_blocks.cpu().numpy()
    deg = np.diff(rp)
    rows = np.repeat(np.arange(n), deg)
    blk = np.searchsorted(rb, np.arange(n), side='right') - 1
    same = blk[rows] == blk[cl]
    out_per_row = np.bincount(rows[~same], minlength=n)
    epb = np.bincount(blk[rows], minlength=len(rb) - 1)
    # distinct (row, col) pairs inside blocks: how dense the diagonal blocks are
    line = ('%s: n=%d nnz=%d max row degree %d, edges per block max %d mean %.0f; inside own part %.3f; rows with >8 outside %d, '
            'outside per row mean %.2f max %d' % (tag, n, nnz, deg.max(), epb.max(), epb.mean(), same.mean(), (out_per_row > 8).sum(),
                                                 out_per_row.mean(), out_per_row.max()))
    res = []
    for d in (4096, 512):
        z = torch.randn(n, 2 * d, device=dev)
        prep = hip.spmm_prepare(b.rowptr, b.col, b.row_blocks) if d >= 1536 else None
        t_blk = timeit(lambda: hip.spmm(b.rowptr, b.col, z[:, :d], z[:, d:], out_scale=b.norm, row_blocks=b.row_blocks, prepared=prep))
        t_plain = timeit(lambda: hip.spmm(b.rowptr, b.col, z[:, :d], z[:, d:], out_scale=b.norm))
        res.append('D=%d blocked %.1f us, ungrouped %.1f us' % (d, t_blk, t_plain))
    # extraction (one launch, with layer 0's aggregation)
    ids = it._epoch_ids[:n]
    t_ex = timeit(lambda: it._extract_with_aggregation(ids, n))
    print(line + ' | ' + '; '.join(res) + ' | extraction %.1f us' % t_ex, flush=True)


ds = datasets.reddit_synth(seed=0)
probe('planted block model', ds, ds.par_li)
dc = datasets.reddit_communities(seed=0)
parts = own_parts(dc)
probe('communities, own parts', dc, parts)
# renumber the nodes part by part (what a pipeline does after METIS): ids contiguous inside a part
order = np.concatenate(parts)
g2 = dc.g.subgraph(order) if False else None
from gist_amd.graph import Graph
gd = dc.g.to(dev).subgraph(order).to('cpu')
starts = np.concatenate([[0], np.cumsum([len(p) for p in parts])])
parts2 = [np.arange(starts[i], starts[i + 1], dtype=np.int64) for i in range(len(parts))]
probe('communities, own parts, nodes renumbered by part', dc._replace(g=gd), parts2)
dh = datasets.community_dataset('nohub', 153431, 602, 41, seed=0, hub_frac=0.0)
probe('communities without hubs, own parts', dh, own_parts(dh))
