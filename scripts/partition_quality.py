#!/usr/bin/env python3
"""Quality and cost of the library's partitioner (gist_partition_graph, the stand-in for
dgl.transform.metis_partition, cluster_gcn/partition_utils.py:11-18) at the BASELINE scales, against the
PLANTED blocks of the synthetic graphs (the best partition there is, by construction) and a random one:
edge cut (share of non-loop edges between parts), balance (largest part / mean), wall time on the host.
Host-only (the partitioner is a CPU function, like METIS): python scripts/partition_quality.py [reddit|amazon|both] [out.json]"""
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
    t0 = time.time()
    part = partition_assignment(g, k, seed=0)
    out['gist_partition_graph'] = dict(cut_stats(g, part.astype(np.int64), k), wall_s=round(time.time() - t0, 2))
    return out


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'both'
    res = [run(w) for w in (['reddit', 'amazon'] if which == 'both' else [which])]
    txt = json.dumps(res, indent=1)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(txt + '\n')
