"""dgl.transform.metis_partition (cluster_gcn/partition_utils.py:7,12).

METIS itself is a third-party library the reference borrows through DGL and is not available
offline.  `metis_partition(g, k)` keeps DGL's call shape -- a dict {part id: object whose
`.ndata[dgl.NID]` are the part's node ids} -- but the partition comes from the library's own host
partitioner (`gist_partition_graph`: restreaming linear-deterministic-greedy, include/gist_hip.h).
Its parts are balanced and keep most edges inside parts on clustered graphs, but they are NOT
METIS's parts: to reproduce a reference run exactly, feed the reference's partition cache
(`../data/{dataset}_{psize}.npy`, sampler.py:44-51) or pass `par_li=` to ClusterIter.
"""
import ctypes

import numpy as np
import torch


class _PartView(object):
    """What the reference reads from a DGL partition subgraph: `.ndata[dgl.NID]`."""

    def __init__(self, nids):
        from . import NID
        self.ndata = {NID: torch.from_numpy(nids)}

    def number_of_nodes(self):
        from . import NID
        return int(self.ndata[NID].numel())


def partition_assignment(g, k, seed=0, n_passes=4, imbalance=0.03):
    """int32 part id per node of `g` (gist_amd.graph.Graph, host or device)."""
    from .. import _lib
    L = _lib.load()
    n = g.number_of_nodes()
    if not 0 < k <= n:
        raise ValueError('gist_amd: metis_partition needs 0 < k <= number of nodes')
    host = [np.ascontiguousarray(t.detach().cpu().numpy().astype(np.int32, copy=False))
            for t in (g.rowptr, g.col, g.t_rowptr, g.t_col)]
    part = np.empty(n, np.int32)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = L.gist_partition_graph(ptr(host[0]), ptr(host[1]), ptr(host[2]), ptr(host[3]), n, int(k),
                                int(seed), int(n_passes), float(imbalance), ptr(part))
    _lib.check(rc, 'gist_partition_graph')
    return part


def metis_partition(g, k, seed=0):
    part = partition_assignment(g, k, seed=seed)
    order = np.argsort(part, kind='stable')
    bounds = np.searchsorted(part[order], np.arange(k + 1))
    return {i: _PartView(order[bounds[i]:bounds[i + 1]].astype(np.int64)) for i in range(k)}
