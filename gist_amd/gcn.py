"""Small full-graph GCN with the reference's gcn/gcn.py:6-67 API (BASELINE config 1:
Cora 2-layer GCN hidden=16; plumbing).  GraphConv stack with whole-tensor layer norm
(`F.layer_norm(h, h.shape)`, gcn/gcn.py:65-66 -- over the WHOLE [N, hidden] tensor)."""
import torch.nn as nn

from . import autograd
from .dgl_compat.nn.pytorch import GraphConv


class GCN(nn.Module):
    def __init__(self, g, in_feats, n_hidden, n_classes, n_layers, activation, dropout,
                 use_layernorm=True, split_input=False, split_output=False, num_subnet=1):
        super().__init__()
        self.g = g
        self.layers = nn.ModuleList()
        self.use_layernorm = use_layernorm
        self.split_input, self.split_output = split_input, split_output
        hs = int(n_hidden // num_subnet)
        fin = int(in_feats // num_subnet) if split_input else in_feats
        first_out = n_hidden if (n_layers <= 1 and not split_output) else hs
        self.layers.append(GraphConv(fin, first_out, activation=activation))
        for i in range(n_layers - 1):
            out = n_hidden if (i == n_layers - 2 and not split_output) else hs
            self.layers.append(GraphConv(hs, out, activation=activation))
        self.layers.append(GraphConv(hs if split_output else n_hidden, n_classes))
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, features):
        h = features
        for i, layer in enumerate(self.layers):
            if i != 0:
                h = self.dropout(h)
            h = layer(self.g, h)
            if i < len(self.layers) - 1 and self.use_layernorm:
                h = autograd.whole_tensor_layer_norm(h)
        return h
