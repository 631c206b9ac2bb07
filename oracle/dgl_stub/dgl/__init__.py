"""TEST INFRASTRUCTURE ONLY -- a minimal stand-in for the `dgl` package.

GIST's reference code (`/root/reference/cluster_gcn/*.py`) imports DGL 0.5.3,
which is not installed in the build container.  This stub implements, on
torch-CPU, exactly the DGL surface those files touch so
that `oracle/gen_golden.py` can import the reference UNCHANGED and record
golden input/output vectors from the reference's own arithmetic.

It is never imported by the product (`gist_amd/`), by `bench.py` or by the GPU
tests; only `oracle/gen_golden.py` puts this directory on `sys.path`.

Semantics restated (they are mathematically unambiguous):
  * update_all(copy_src, sum): y[v] = sum over in-edges (u->v) of x[u]
    (duplicate edges count twice, self loops are ordinary edges)
  * in_degrees(): number of in-edges per node
  * subgraph(ids): node-induced subgraph, node i of the result is ids[i],
    edges keep their original relative order, ndata rows are gathered
"""
import numpy as np
import torch

from . import function  # noqa: F401
from . import backend  # noqa: F401

NID = '_ID'


class _NData(dict):
    def pop(self, k, *a):
        return dict.pop(self, k, *a)


class DGLGraph(object):
    """Directed multigraph stored as an edge list (src -> dst)."""

    def __init__(self, data=None, num_nodes=None, idtype=torch.int64):
        self.ndata = _NData()
        self._idtype = idtype
        self._device = torch.device('cpu')
        if data is None:
            self._src = np.zeros(0, np.int64)
            self._dst = np.zeros(0, np.int64)
            self._n = int(num_nodes or 0)
        elif isinstance(data, tuple):
            s, d = data
            self._src = np.asarray(s, np.int64)
            self._dst = np.asarray(d, np.int64)
            self._n = int(num_nodes if num_nodes is not None else
                          (max(self._src.max(), self._dst.max()) + 1))
        else:  # networkx graph (gcn/train.py:69)
            import networkx as nx
            g = data if data.is_directed() else data.to_directed()
            nodes = sorted(g.nodes())
            assert nodes == list(range(len(nodes)))
            e = np.array(list(g.edges()), np.int64).reshape(-1, 2)
            self._src, self._dst, self._n = e[:, 0].copy(), e[:, 1].copy(), len(nodes)
        self._adj = None

    # -- structure ---------------------------------------------------------
    def number_of_nodes(self):
        return self._n

    def number_of_edges(self):
        return int(self._src.shape[0])

    def in_degrees(self):
        d = np.bincount(self._dst, minlength=self._n).astype(np.int64)
        return torch.from_numpy(d).to(self._device)

    def _in_adj(self):
        if self._adj is None:
            idx = torch.from_numpy(np.stack([self._dst, self._src]))
            val = torch.ones(idx.shape[1], dtype=torch.float32)
            self._adj = torch.sparse_coo_tensor(
                idx, val, (self._n, self._n)).coalesce()
        return self._adj

    def in_csr(self):
        """CSR of in-edges with the original edge order kept inside a row."""
        order = np.argsort(self._dst, kind='stable')
        col = self._src[order]
        rowptr = np.zeros(self._n + 1, np.int64)
        np.cumsum(np.bincount(self._dst, minlength=self._n), out=rowptr[1:])
        return rowptr, col

    # -- frames ------------------------------------------------------------
    def local_var(self):
        g = DGLGraph.__new__(DGLGraph)
        g.__dict__.update(self.__dict__)
        g.ndata = _NData(self.ndata)
        return g

    def update_all(self, msg, red, apply_fn=None):
        assert msg[0] == 'copy_src' and red[0] == 'sum'
        _, src_field, m_field = msg
        _, m_field2, out_field = red
        assert m_field == m_field2
        x = self.ndata[src_field]
        self.ndata[out_field] = torch.sparse.mm(self._in_adj(), x)

    def subgraph(self, nids):
        if torch.is_tensor(nids):
            nids = nids.cpu().numpy()
        nids = np.asarray(nids, np.int64).reshape(-1)
        remap = np.full(self._n, -1, np.int64)
        remap[nids] = np.arange(nids.shape[0])
        keep = (remap[self._src] >= 0) & (remap[self._dst] >= 0)
        sg = DGLGraph((remap[self._src[keep]], remap[self._dst[keep]]),
                      num_nodes=nids.shape[0], idtype=self._idtype)
        tid = torch.from_numpy(nids)
        for k, v in self.ndata.items():
            sg.ndata[k] = v[tid]
        sg.ndata[NID] = tid
        return sg

    # -- dtype / device ------------------------------------------------------
    def int(self):
        g = self.local_var()
        g._idtype = torch.int32
        return g

    def long(self):
        g = self.local_var()
        g._idtype = torch.int64
        return g

    def to(self, device):
        g = self.local_var()
        g._device = torch.device(device) if not isinstance(device, torch.device) else device
        g.ndata = _NData({k: v.to(g._device) for k, v in self.ndata.items()})
        return g

    def cpu(self):
        return self.to('cpu')


def graph(data, num_nodes=None):
    return DGLGraph(data, num_nodes=num_nodes)


def batch(graphs, edge_attrs=None, node_attrs=None):
    raise NotImplementedError('dgl.batch is out of scope (PPI path)')
