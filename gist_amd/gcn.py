"""Small full-graph GCN for BASELINE config 1 (Cora-sized graphs; plumbing).

API of the reference's `gcn/gcn.py:6-67`: `GCN(g, in_feats, n_hidden, n_classes, n_layers,
activation, dropout, use_layernorm, split_input, split_output, num_subnet)`, `forward(features)`.
A stack of GraphConv layers (sym-normalised aggregation on the HIP SpMM + fp32-MFMA GEMM) with
dropout before every layer but the first and -- the reference's quirk, gcn.py:65-66 -- a layer
norm taken over the WHOLE [N, hidden] activation tensor rather than per row.
"""
import torch.nn as nn

from . import autograd
from .dgl_compat.nn.pytorch import GraphConv


def graphconv_dims(in_feats, n_hidden, n_classes, n_layers, split_input, split_output, num_subnet):
    """[(fan_in, fan_out)] of the n_layers + 1 GraphConv layers (gcn.py:26-55)."""
    width = n_hidden // num_subnet
    sizes = [(in_feats // num_subnet if split_input else in_feats,
              width if (n_layers > 1 or split_output) else n_hidden)]
    for k in range(1, n_layers):
        last_hidden = (k == n_layers - 1) and not split_output
        sizes.append((width, n_hidden if last_hidden else width))
    sizes.append((width if split_output else n_hidden, n_classes))
    return sizes


class GCN(nn.Module):
    def __init__(self, g, in_feats, n_hidden, n_classes, n_layers, activation, dropout,
                 use_layernorm=True, split_input=False, split_output=False, num_subnet=1):
        super().__init__()
        self.g, self.use_layernorm = g, use_layernorm
        self.split_input, self.split_output = split_input, split_output
        sizes = graphconv_dims(in_feats, n_hidden, n_classes, n_layers, split_input, split_output,
                               num_subnet)
        convs = [GraphConv(i, o, activation=activation) for (i, o) in sizes[:-1]]
        convs.append(GraphConv(*sizes[-1]))                 # output layer: no activation
        self.layers = nn.ModuleList(convs)
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, features):
        h = features
        n = len(self.layers)
        for k, conv in enumerate(self.layers):
            h = conv(self.g, self.dropout(h) if k else h)
            if self.use_layernorm and k + 1 < n:
                h = autograd.whole_tensor_layer_norm(h)
        return h
