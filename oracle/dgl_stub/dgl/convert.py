def from_scipy(adj):
    from . import DGLGraph
    coo = adj.tocoo()
    return DGLGraph((coo.row, coo.col), num_nodes=adj.shape[0])
