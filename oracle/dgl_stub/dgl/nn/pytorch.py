"""Stub of dgl.nn.pytorch.GraphConv ([DGL 0.5.3, recalled]: norm='both', degrees
clamped to >= 1, weight [in,out] xavier-uniform, bias zeros, multiply by W first
iff in_feats > out_feats).  Parity for this layer is UNPINNED by the reference."""
import torch
import torch.nn as nn


class GraphConv(nn.Module):
    def __init__(self, in_feats, out_feats, norm='both', weight=True, bias=True,
                 activation=None):
        super().__init__()
        self._in, self._out, self._norm = in_feats, out_feats, norm
        self.weight = nn.Parameter(torch.Tensor(in_feats, out_feats))
        self.bias = nn.Parameter(torch.Tensor(out_feats))
        nn.init.xavier_uniform_(self.weight)
        nn.init.zeros_(self.bias)
        self._activation = activation

    def forward(self, graph, feat):
        graph = graph.local_var()
        import numpy as np
        out_deg = torch.from_numpy(
            np.bincount(graph._src, minlength=graph._n)).float().clamp(min=1)
        feat = feat * torch.pow(out_deg, -0.5).unsqueeze(1)
        if self._in > self._out:
            feat = torch.matmul(feat, self.weight)
            rst = torch.sparse.mm(graph._in_adj(), feat)
        else:
            rst = torch.sparse.mm(graph._in_adj(), feat)
            rst = torch.matmul(rst, self.weight)
        in_deg = graph.in_degrees().float().clamp(min=1)
        rst = rst * torch.pow(in_deg, -0.5).unsqueeze(1)
        rst = rst + self.bias
        if self._activation is not None:
            rst = self._activation(rst)
        return rst
