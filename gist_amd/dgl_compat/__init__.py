"""The slice of the `dgl` package namespace that GIST's in-scope scripts import
(SURVEY.md section 8b), backed by gist_amd's HIP kernels.

    import gist_amd.dgl_compat as dgl          # explicit
    gist_amd.dgl_compat.install()              # or: make `import dgl` resolve to this package

Only what the hot path needs is implemented; everything else raises with a clear
message instead of silently degrading.
"""
import sys

import numpy as np

from ..graph import Graph
from . import function  # noqa: F401
from . import backend  # noqa: F401
from . import transform  # noqa: F401
from . import data  # noqa: F401
from . import convert  # noqa: F401

NID = '_ID'


def DGLGraph(data=None, num_nodes=None):
    """`DGLGraph(nx_graph)` (gcn/train.py:69) or `DGLGraph((src, dst))`."""
    if isinstance(data, tuple):
        src, dst = data
        n = num_nodes if num_nodes is not None else int(max(np.max(src), np.max(dst))) + 1
        return Graph.from_edges(src, dst, n)
    if data is None:
        raise ValueError('gist_amd: empty DGLGraph construction is not supported')
    g = data if data.is_directed() else data.to_directed()     # networkx
    nodes = sorted(g.nodes())
    if nodes != list(range(len(nodes))):
        raise ValueError('gist_amd: networkx nodes must be 0..n-1')
    e = np.array(list(g.edges()), np.int64).reshape(-1, 2)
    return Graph.from_edges(e[:, 0], e[:, 1], len(nodes))


def graph(data, num_nodes=None):
    return DGLGraph(data, num_nodes=num_nodes)


def batch(graphs, edge_attrs=None, node_attrs=None):
    raise NotImplementedError('gist_amd: dgl.batch (PPI dataset path) is out of scope')


def install():
    """Register this package as `dgl` so reference-style scripts import unchanged."""
    pkg = sys.modules[__name__]
    sys.modules.setdefault('dgl', pkg)
    for sub in ('function', 'backend', 'transform', 'data', 'convert', 'nn', 'nn.pytorch',
                'data.utils'):
        mod = __import__(__name__ + '.' + sub, fromlist=['_'])
        sys.modules.setdefault('dgl.' + sub, mod)
    return pkg
