#!/usr/bin/env python3
"""Quality and cost of the library's partitioner (gist_partition_graph, the stand-in for
dgl.transform.metis_partition, cluster_gcn/partition_utils.py:11-18) at the BASELINE scales, against the
PLANTED blocks of the synthetic graphs (the best partition there is, by construction) and a random one:
edge cut (share of non-loop edges between parts), balance (largest part / mean), wall time on the host.
Round 5 adds a torus mesh (ideal cut known) and a power-law community graph (no planted k-way partition).
Host-only (the partitioner is a CPU function, like METIS): python scripts/partition_quality.py [reddit|amazon|torus|communities|both|all] [out.json]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gist_amd import datasets                                   # noqa: E402
from gist_amd.dgl_compat.transform import partition_assignment  # noqa: E402


def cut_stats(g, part, k):
    rp = g.rowptr.numpy().astype(np.int64)
    col = g.col.numpy()
    dst = np.repeat(np.arange(g.number_of_nodes()), np.diff(rp))
    keep = dst != col
    cut = float((part[dst[keep]] != part[col[keep]]).mean())
    sizes = np.bincount(part, minlength=k)
    return dict(edge_cut_fraction=round(cut, 5), largest_over_mean=round(float(sizes.max() / sizes.mean()), 4),
                smallest_over_mean=round(float(sizes.min() / sizes.mean()), 4), empty_parts=int((sizes == 0).sum()))


def stages():
    import ctypes
    from gist_amd import _lib
    st = (ctypes.c_double * 16)()
    _lib.load().gist_partition_last_stats(st, 16)
    names = ('input_graph', 'clustering', 'contraction', 'visiting_orders', 'coarse_levels', 'input_level_refinement',
             'balance_repair')
    d = {k: round(st[i], 2) for i, k in enumerate(names)}
    d['levels'], d['coarsest_vertices'] = int(st[8]), int(st[9])
    return d


def partition(g, k, out, key='gist_partition_graph', **kw):
    t0 = time.time()
    part = partition_assignment(g, k, seed=0, **kw)
    out[key] = dict(cut_stats(g, part.astype(np.int64), k), wall_s=round(time.time() - t0, 2), stage_seconds=stages())
    return part


def run(name):
    ds = datasets.reddit_synth(seed=0) if name == 'reddit' else datasets.amazon_synth(seed=1)
    g = ds.g
    n, k = g.number_of_nodes(), len(ds.par_li)
    planted = np.empty(n, np.int64)
    for b, ids in enumerate(ds.par_li):
        planted[ids] = b
    out = dict(graph=ds.name, nodes=n, edges=int(g.number_of_edges()), parts=k,
               planted=cut_stats(g, planted, k),
               random=cut_stats(g, np.random.RandomState(0).randint(0, k, n), k))
    partition(g, k, out)
    return out


def run_torus(W=96, k=64):
    from gist_amd.graph import Graph
    idx = np.arange(W * W).reshape(W, W)
    src = np.concatenate([idx.ravel(), idx.ravel()])
    dst = np.concatenate([np.roll(idx, 1, 0).ravel(), np.roll(idx, 1, 1).ravel()])
    g = Graph.from_edges(np.concatenate([src, dst]), np.concatenate([dst, src]), W * W)
    side = int(round((W * W / k) ** 0.5))
    out = dict(graph='torus %d x %d' % (W, W), nodes=W * W, edges=int(g.number_of_edges()), parts=k,
               ideal=dict(edge_cut_fraction=round(1.0 / side, 5), note='%d x %d squares' % (side, side)))
    partition(g, k, out)
    out['cut_over_ideal'] = round(out['gist_partition_graph']['edge_cut_fraction'] / out['ideal']['edge_cut_fraction'], 3)
    return out


def run_communities(k=1500):
    """Power-law communities (30-400 nodes, mixing 0.3), Reddit-sized: no planted k-way partition exists.  Reference points:
    the communities themselves (NOT balanced: the floor = the mixing), a partition built from the ground truth (every
    community cut into near-equal pieces that fit a part, pieces bin-packed), a random partition."""
    ds = datasets.reddit_communities(seed=0)
    g = ds.g
    n = g.number_of_nodes()
    comm = np.empty(n, np.int64)
    for q, ids in enumerate(ds.par_li):
        comm[ids] = q
    sizes = np.array([len(p) for p in ds.par_li])
    cap = int(np.ceil(1.03 * n / k))
    rs = np.random.RandomState(0)
    pieces = []
    for ids in ds.par_li:
        s_, p_ = len(ids), int(np.ceil(len(ids) / cap))
        ids = rs.permutation(ids)
        pieces += [ids[q * s_ // p_:(q + 1) * s_ // p_] for q in range(p_)]
    pieces.sort(key=len, reverse=True)
    truth, fill = np.empty(n, np.int64), np.zeros(k, np.int64)
    for pc in pieces:
        cand = np.flatnonzero(fill + len(pc) <= cap)
        while len(pc):
            b = cand[np.argmax(fill[cand])] if len(cand) else int(np.argmin(fill))
            take = pc[:cap - fill[b]]
            truth[take], fill[b], pc, cand = b, fill[b] + len(take), pc[len(take):], np.zeros(0, np.int64)
    out = dict(graph=ds.name, nodes=n, edges=int(g.number_of_edges()), parts=k,
               communities=dict(count=len(sizes), size_min=int(sizes.min()), size_median=int(np.median(sizes)),
                                size_mean=round(float(sizes.mean()), 1), size_max=int(sizes.max()),
                                share_of_nodes_in_communities_larger_than_a_part=round(float(sizes[sizes > cap].sum() / n), 3)),
               communities_as_parts_unbalanced=cut_stats(g, comm, len(sizes)),
               packed_from_ground_truth=cut_stats(g, truth, k),
               random=cut_stats(g, np.random.RandomState(0).randint(0, k, n), k))
    partition(g, k, out)
    partition(g, k, out, key='gist_partition_graph_16_passes', n_passes=16)
    best = min(out['gist_partition_graph']['edge_cut_fraction'], out['gist_partition_graph_16_passes']['edge_cut_fraction'],
               out['packed_from_ground_truth']['edge_cut_fraction'])
    out['cut_over_best_known'] = round(out['gist_partition_graph']['edge_cut_fraction'] / best, 4)
    return out


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    todo = {'both': ['reddit', 'amazon'], 'all': ['reddit', 'amazon', 'torus', 'communities']}.get(which, [which])
    res = []
    for w in todo:
        res.append(run_torus() if w == 'torus' else run_communities() if w == 'communities' else run(w))
    import os
    meta = dict(host_cpus=os.cpu_count(), note='wall times on the host that ran this script; the partition does not depend on it')
    txt = json.dumps(dict(meta=meta, results=res), indent=1)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(txt + '\n')
