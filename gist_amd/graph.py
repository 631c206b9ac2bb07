"""Device-resident graph exposing the DGL surface GIST's hot path uses.

The reference drives DGL through a handful of calls:
`g.local_var()`, `g.ndata[...]`, `g.update_all(fn.copy_src, fn.sum)`,
`g.in_degrees()`, `g.subgraph(ids)`, `g.to(device)`, `g.int()/long()`,
`g.number_of_nodes()/number_of_edges()`.  This class provides exactly those on
top of two int32 CSR structures kept in HBM:

  rowptr/col      in-edges  (row = destination)  -> forward aggregation
  t_rowptr/t_col  out-edges (row = source)        -> backward aggregation

The reversed CSR of an induced subgraph is the induced subgraph of the reversed
CSR, so both are extracted by the same kernels without any atomics and the
summation order stays deterministic.

Structure can be built on the host (numpy) -- that is graph construction, not the
hot path -- but every compute call requires the graph to be on the GPU.
"""
import weakref

import numpy as np
import torch

from . import hip


class NData(dict):
    pass


def _csr_from_edges_host(src, dst, n):
    """Host (numpy) CSR builder: rows = dst, stable edge order (graph construction only)."""
    src = np.asarray(src, np.int64)
    dst = np.asarray(dst, np.int64)
    order = np.argsort(dst, kind='stable')
    rowptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(dst, minlength=n), out=rowptr[1:])
    return rowptr, src[order]


class Graph(object):
    def __init__(self, rowptr, col, t_rowptr, t_col, n_nodes, idtype=torch.int64):
        self.rowptr, self.col = rowptr, col
        self.t_rowptr, self.t_col = t_rowptr, t_col
        self._n = int(n_nodes)
        self._idtype = idtype
        self.ndata = NData()
        self._norm = None
        self._remap = None
        self._nnz = None

    # -- construction (host) ------------------------------------------------------
    @classmethod
    def from_edges(cls, src, dst, num_nodes):
        """Directed multigraph from an edge list (u -> v); duplicates and self loops kept."""
        src = np.asarray(src, np.int64)
        dst = np.asarray(dst, np.int64)
        if num_nodes >= 2 ** 31 or src.shape[0] >= 2 ** 31:
            raise ValueError('gist_amd: graph too large for int32 device indices')
        rp, cl = _csr_from_edges_host(src, dst, num_nodes)
        trp, tcl = _csr_from_edges_host(dst, src, num_nodes)
        g = cls(torch.from_numpy(rp.astype(np.int32)), torch.from_numpy(cl.astype(np.int32)),
                torch.from_numpy(trp.astype(np.int32)), torch.from_numpy(tcl.astype(np.int32)),
                num_nodes)
        g._nnz = int(src.shape[0])
        return g

    @classmethod
    def from_scipy(cls, adj):
        coo = adj.tocoo()
        return cls.from_edges(coo.row, coo.col, adj.shape[0])

    # -- DGL surface ------------------------------------------------------------------
    @property
    def device(self):
        return self.rowptr.device

    def number_of_nodes(self):
        return self._n

    num_nodes = number_of_nodes

    def number_of_edges(self):
        if self._nnz is None:
            self._nnz = int(self.rowptr[-1].item())
        return self._nnz

    num_edges = number_of_edges

    def local_var(self):
        """Shallow copy with its own feature frame (cluster_gcn/modules.py:219)."""
        g = Graph.__new__(Graph)
        g.__dict__.update(self.__dict__)
        g.ndata = NData(self.ndata)
        return g

    def int(self):
        g = self.local_var()
        g._idtype = torch.int32
        return g

    def long(self):
        g = self.local_var()
        g._idtype = torch.int64
        return g

    def to(self, device):
        device = torch.device(device) if not isinstance(device, torch.device) else device
        if device.type == 'cuda' and device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        if device == self.device:
            g = self.local_var()
        else:
            g = Graph(self.rowptr.to(device), self.col.to(device), self.t_rowptr.to(device),
                      self.t_col.to(device), self._n, self._idtype)
            g._nnz = self._nnz
            if getattr(self, 'node_blocks', None) is not None:
                g.node_blocks = self.node_blocks      # (host boundaries of the id-ordered parts, if known)
        g.ndata = NData({k: v.to(device) for k, v in self.ndata.items()})
        return g

    def cpu(self):
        return self.to('cpu')

    def cuda(self):
        return self.to('cuda')

    def in_degrees(self):
        d = (self.rowptr[1:] - self.rowptr[:-1])
        return d.to(self._idtype)

    def norm(self):
        """1/in_degree, inf -> 0 (cluster_gcn/modules.py:239-243), cached per structure."""
        if self._norm is None:
            self._norm = hip.in_degree_norm(self.rowptr)
        return self._norm

    def update_all(self, message_func, reduce_func, apply_node_func=None):
        """Only the builtin pair the reference uses: copy_src + sum
        (cluster_gcn/modules.py:224-225, sampler.py:64-66)."""
        from .dgl_compat import function as fn
        if not (isinstance(message_func, fn.CopySrc) and isinstance(reduce_func, fn.Sum)):
            raise NotImplementedError('gist_amd: update_all supports fn.copy_src + fn.sum only')
        if message_func.out != reduce_func.msg:
            raise ValueError('gist_amd: message field %r != reduce field %r'
                             % (message_func.out, reduce_func.msg))
        from .autograd import spmm_sum
        x = self.ndata[message_func.src]
        self.ndata[reduce_func.out] = spmm_sum(self, x)
        if apply_node_func is not None:
            raise NotImplementedError('gist_amd: apply_node_func is not supported')

    def subgraph(self, nids, with_ndata=True):
        """Node-induced subgraph, node i of the result = nids[i]; ndata rows gathered
        (cluster_gcn/partition_utils.py:23, sampler.py:34)."""
        if not self.rowptr.is_cuda:
            raise RuntimeError('gist_amd: subgraph() runs on the GPU; call g.to(device) first '
                               '(no CPU fallback)')
        dev = self.device
        if torch.is_tensor(nids):
            ids = nids.to(device=dev, dtype=torch.int32).contiguous()
        else:
            nids = np.asarray(nids).reshape(-1)
            ids = torch.from_numpy(nids.astype(np.int32)).to(dev)
        nb = ids.numel()
        if self._remap is None:
            self._remap = torch.empty(self._n, dtype=torch.int32, device=dev)
            hip.fill_i32_(self._remap, -1)
        remap = self._remap
        from . import ops  # noqa: F401  (registers torch.ops.gist.*)
        # the generic path sizes the column arrays exactly: one host sync per structure (the
        # training loop's batches come from gist_extract_batch inside the step instead)
        srp, scl = torch.ops.gist.induced_subgraph(self.rowptr, self.col, ids, remap)
        trp, tcl = torch.ops.gist.induced_subgraph(self.t_rowptr, self.t_col, ids, remap)
        nnz = scl.numel()
        sg = Graph(srp, scl, trp, tcl, nb, self._idtype)
        sg._nnz = nnz
        idl = None
        for k, v in (self.ndata.items() if with_ndata else ()):
            if v.dtype == torch.float32 and v.dim() == 2 and v.is_cuda:
                out = torch.empty(nb, v.shape[1], dtype=torch.float32, device=dev)
                sg.ndata[k] = hip.gather_rows(v, ids, out)
            elif v.dtype == torch.int32 and v.dim() == 1 and v.is_cuda:
                sg.ndata[k] = hip.gather_i32(v, ids, torch.empty(nb, dtype=torch.int32, device=dev))
            else:                              # masks / int64 labels: small, plumbing
                if idl is None:
                    idl = ids.long()
                sg.ndata[k] = v.to(dev)[idl]
        from .dgl_compat import NID
        sg.ndata[NID] = ids.to(self._idtype)
        return sg


class AllRowsMask(torch.Tensor):
    """A boolean row mask that is KNOWN to be all True (the train mask of a batch of the train-induced graph,
    cluster_gcn/sampler.py:34): `x[mask]` with a leading dimension of mask.numel() is x itself -- no nonzero(), no
    device-to-host synchronisation, no gather (cluster_gcn/cluster_gcn.py:100-101 indexes the logits and the labels
    with it in every iteration).  Everything else sees an ordinary bool tensor."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if func is torch.Tensor.__getitem__ and len(args) == 2 and type(args[1]) is AllRowsMask:
            x, m = args
            if isinstance(x, torch.Tensor) and x.dim() >= 1 and m.dim() == 1 and x.shape[0] == m.shape[0]:
                return x
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))


class _BatchNData(dict):
    """ndata of a ClusterBatch: the train graph's columns for the batch's rows, produced on first access -- a slice of
    the column gathered once per epoch in batch order (ClusterIter.epoch_column)."""

    def __init__(self, batch):
        super().__init__()
        self._b = weakref.proxy(batch)      # (no reference cycle: a finished batch is freed by its refcount, not by the GC)

    def _lazy_keys(self):
        from .dgl_compat import NID
        return list(self._b._it.g.ndata.keys()) + [NID]

    def __missing__(self, key):
        from .dgl_compat import NID
        b = self._b
        it = b._it
        a = int(it._offsets[b._j])
        if key == NID:
            v = b._ids.to(b._idtype)
        elif key not in it.g.ndata:
            raise KeyError(key)
        elif key == 'train_mask' and it._all_train:
            v = it._ones[:b._n].as_subclass(AllRowsMask)
        elif key == 'feat':
            src = it.g.ndata['feat']
            v = hip.gather_rows(src, b._ids, torch.empty(b._n, src.shape[1], dtype=torch.float32, device=src.device))
        else:
            v = it.epoch_column(key)[a:a + b._n]
        self[key] = v
        return v

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy_keys()

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        ks = list(dict.keys(self))
        return ks + [k for k in self._lazy_keys() if k not in ks]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def pop(self, key, *default):
        if key in self:
            v = self[key]
            dict.pop(self, key, None)
            return v
        if default:
            return default[0]
        raise KeyError(key)


class ClusterBatch(Graph):
    """What ClusterIter yields in feed mode (cluster_gcn/sampler.py:85-93): batch j of the current epoch, DESCRIBED --
    node ids, the row ranges of its parts -- with nothing launched.  A gist_amd.modules.GCN bound to the iterator
    extracts it on the device as the first launch of its forward (gist_sage_step); any other consumer that touches the
    structure (rowptr / col / update_all / in_degrees / subgraph ...) gets the induced subgraph built on first access by
    the generic path (Graph.subgraph: its own tensors), exactly what the eager iterator yielded."""

    def __init__(self, it, j):
        self._it, self._j = it, int(j)
        ids, n, row_blocks, parts, next_info = it.describe(j)
        self._ids, self._n = ids, n
        self.row_blocks, self.parts, self.next_info = row_blocks, parts, next_info
        self.siblings = it.has_siblings(j)
        self._idtype = it.g._idtype
        self._sg = None
        self._norm = self._remap = self._nnz = None
        self.ndata = _BatchNData(self)

    def _structure(self):
        if self._sg is None:
            it = self._it
            sg = it.g.subgraph(self._ids, with_ndata=False)      # (the batch's columns come from _BatchNData)
            self._sg = sg
            self._nnz = sg._nnz
        return self._sg

    rowptr = property(lambda self: self._structure().rowptr)
    col = property(lambda self: self._structure().col)
    t_rowptr = property(lambda self: self._structure().t_rowptr)
    t_col = property(lambda self: self._structure().t_col)

    @property
    def device(self):
        return self._ids.device

    def local_var(self):
        g = ClusterBatch.__new__(ClusterBatch)
        g.__dict__.update(self.__dict__)
        nd = _BatchNData(g)
        dict.update(nd, self.ndata)
        g.ndata = nd
        return g

    def to(self, device):
        device = torch.device(device) if not isinstance(device, torch.device) else device
        if device.type == 'cuda' and (device.index is None or device == self.device):
            return self                               # cluster_gcn/cluster_gcn.py:97 (already there)
        sg = self._structure().local_var()
        sg.ndata = NData({k: self.ndata[k] for k in self.ndata.keys()})
        return sg.to(device)
