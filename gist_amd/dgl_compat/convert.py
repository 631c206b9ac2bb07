from ..graph import Graph


def from_scipy(adj):
    return Graph.from_scipy(adj)
