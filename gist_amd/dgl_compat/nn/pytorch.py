"""dgl.nn.pytorch.GraphConv as gcn/gcn.py:4,30-56 uses it (BASELINE config 1, Cora plumbing).

DGL is not in the reference tree; the semantics follow DGL 0.5.3's documented behaviour
(norm='both': h' = D_in^-1/2 A D_out^-1/2 h W + b, degrees clamped to >= 1, weight [in, out]
xavier-uniform, bias zeros, multiply by W first iff in_feats > out_feats).  Parity for this
layer is UNPINNED by the reference (oracle: graphconv_forward, fixture G5 from the stub).
"""
import torch
import torch.nn as nn

from ... import autograd


class GraphConv(nn.Module):
    def __init__(self, in_feats, out_feats, norm='both', weight=True, bias=True, activation=None):
        super().__init__()
        if norm != 'both' or not weight or not bias:
            raise NotImplementedError('gist_amd: GraphConv supports norm="both" with weight and bias')
        self._in_feats, self._out_feats = in_feats, out_feats
        self.weight = nn.Parameter(torch.Tensor(in_feats, out_feats))
        self.bias = nn.Parameter(torch.Tensor(out_feats))
        nn.init.xavier_uniform_(self.weight)
        nn.init.zeros_(self.bias)
        self._activation = activation

    def forward(self, graph, feat):
        out_deg = (graph.t_rowptr[1:] - graph.t_rowptr[:-1]).float().clamp(min=1)
        in_deg = (graph.rowptr[1:] - graph.rowptr[:-1]).float().clamp(min=1)
        ns = torch.pow(out_deg, -0.5).contiguous()
        nd = torch.pow(in_deg, -0.5).contiguous()
        if self._in_feats > self._out_feats:
            rst = autograd.spmm_sum(graph, autograd.matmul(feat, self.weight), out_scale=nd,
                                    src_scale=ns)
        else:
            rst = autograd.matmul(autograd.spmm_sum(graph, feat, out_scale=nd, src_scale=ns),
                                  self.weight)
        rst = rst + self.bias
        if self._activation is not None:
            rst = self._activation(rst)
        return rst
