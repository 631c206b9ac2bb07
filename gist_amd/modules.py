"""GraphSAGE model with the reference's module API (cluster_gcn/modules.py:191-314).

`ISTSAGELayer` and `GCN` take the same constructor arguments, expose the same
attributes (`layers[i].linear.weight / .bias`, `dropout`, `lynorm`, `activation`)
and initialise their parameters with the same torch RNG calls in the same order
(modules.py:201-203,213-216), so a script written against the reference runs
unchanged and same-seed weights are identical.  forward() runs entirely on the
HIP kernels through gist_amd.autograd.sage_layer.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import autograd, module_engine


def _is_relu(fn):
    return fn is F.relu or fn is torch.relu or isinstance(fn, nn.ReLU)


class ISTSAGELayer(nn.Module):
    """mean-aggregate -> concat -> dropout -> Linear(2*in, out) -> LayerNorm(no affine) -> act."""

    def __init__(self, in_feats, out_feats, dropout, use_lynorm, activation=None):
        super().__init__()
        # the input feature size doubles: [h | mean of in-neighbours]  (modules.py:199-201)
        self.linear = nn.Linear(2 * in_feats, out_feats)
        self.activation = activation
        self.init_layer()
        self.p_drop = float(dropout) if dropout else 0.0
        self.dropout = nn.Dropout(p=dropout) if dropout else 0.
        self.use_lynorm = bool(use_lynorm)
        self.lynorm = (nn.LayerNorm(out_feats, elementwise_affine=False) if use_lynorm
                       else (lambda x: x))
        self.drop_seed = 0

    def init_layer(self):
        stdv = 1. / math.sqrt(self.linear.weight.size(1))          # modules.py:213-216
        self.linear.weight.data.uniform_(-stdv, stdv)
        self.linear.bias.data.uniform_(-stdv, stdv)

    def forward(self, g, h):
        act = self.activation
        fused_relu = act is not None and _is_relu(act)
        p = self.p_drop if (self.training and self.p_drop > 0.0) else 0.0
        out = autograd.sage_layer(g, h, self.linear.weight, self.linear.bias, self.use_lynorm,
                                  fused_relu, p, self.drop_seed)
        if act is not None and not fused_relu:
            out = act(out)
        return out

    def get_norm(self, g):
        return g.norm().unsqueeze(1)


class GraphSAGELayer(nn.Module):
    """The reference's GraphSAGELayer (cluster_gcn/modules.py:100-159) -- the layer --use-pp was
    written for: with use_pp=True and in training mode the input already is [h | A^h] (built once
    by ClusterIter.precalc, sampler.py:58-69) and the layer only projects it; otherwise it
    aggregates like ISTSAGELayer.  LayerNorm here is elementwise_affine=True (modules.py:123)."""

    def __init__(self, in_feats, out_feats, activation, dropout, bias=True, use_pp=False,
                 use_lynorm=True):
        super().__init__()
        self.linear = nn.Linear(2 * in_feats, out_feats, bias=bias)
        self.activation = activation
        self.use_pp = use_pp
        self.dropout = nn.Dropout(p=dropout) if dropout else 0.
        self.lynorm = nn.LayerNorm(out_feats, elementwise_affine=True) if use_lynorm else (lambda x: x)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.linear.weight.size(1))
        self.linear.weight.data.uniform_(-stdv, stdv)
        if self.linear.bias is not None:
            self.linear.bias.data.uniform_(-stdv, stdv)

    def forward(self, g, h):
        if not self.use_pp or not self.training:               # modules.py:133-139
            ah = autograd.spmm_sum(g, h, out_scale=g.norm())
            h = torch.cat((h, ah), dim=1)
        if self.dropout:
            h = self.dropout(h)
        h = autograd.matmul(h, self.linear.weight.t())
        if self.linear.bias is not None:
            h = h + self.linear.bias
        h = self.lynorm(h)
        if self.activation:
            h = self.activation(h)
        return h


class GCN(nn.Module):
    """Layer sizing of the reference's GCN (modules.py:245-308)."""

    def __init__(self, in_feats, n_hidden, n_classes, n_layers, activation, dropout,
                 use_layernorm=True, split_input=False, split_output=False, num_subnet=1,
                 use_aggregation=False):
        super().__init__()
        self.layers = nn.ModuleList()
        self.use_layernorm = use_layernorm
        self.split_input = split_input
        self.split_output = split_output
        if not use_aggregation:
            raise NotImplementedError('You must use graph sage')
        layer_type = ISTSAGELayer
        hs = int(n_hidden // num_subnet)
        first_in = int(in_feats // num_subnet) if split_input else in_feats
        if n_layers <= 1 and not split_output:
            self.layers.append(layer_type(first_in, n_hidden, dropout, use_layernorm,
                                          activation=activation))
        else:
            self.layers.append(layer_type(first_in, hs, dropout, use_layernorm,
                                          activation=activation))
        for i in range(n_layers - 1):
            if i == n_layers - 2 and not split_output:
                self.layers.append(layer_type(hs, n_hidden, dropout, use_layernorm,
                                              activation=activation))
            else:
                self.layers.append(layer_type(hs, hs, dropout, use_layernorm,
                                              activation=activation))
        if split_output:
            self.layers.append(layer_type(hs, n_classes, dropout, False, activation=None))
        else:
            self.layers.append(layer_type(n_hidden, n_classes, dropout, False, activation=None))

    def set_dropout_seed(self, seed):
        self._drop_seed = int(seed)           # (the fused step's generator: one seed, a running element offset)
        for k, layer in enumerate(self.layers):
            layer.drop_seed = int(seed) * 1000003 + k

    def forward(self, g):
        """cluster_gcn/modules.py:310-314.  The returned logits are the caller's own tensor, as in torch: on the fused
        path they come from a small ring of buffers, and a buffer the caller still holds (the tensor or a view of it) is
        never handed out again (gist_amd/module_engine.py); evaluation-mode logits are freshly allocated."""
        # A cluster batch from ClusterIter: the whole forward is ONE dispatcher op on the preallocated step plan
        # (gist_amd/module_engine.py); anything else -- the full graph of an evaluation, a hand-built graph -- runs
        # layer by layer below
        me = module_engine.engine_for(self, g)
        if me is not None:
            return me.forward(g, self.training)
        h = g.ndata['feat']
        for layer in self.layers:
            h = layer(g, h)
        return h
