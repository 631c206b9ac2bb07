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


class FullGraphTrainer(object):
    """Whole-graph training of the small GCN above on one GPU: what `gcn/train.py` does in its body, as an
    object the CLI (`gist_amd/scripts/gcn_train.py`) and tests drive.

    The citation dataset (numpy features / labels / masks + a networkx digraph, the legacy DGL object
    `gcn/train.py:38-45` reads) is moved to the device once; `self_loop` rebuilds every node's loop
    (`gcn/train.py:65-68`).  `step()` is one optimisation step over the training mask (`:103-109`);
    `fit()` runs the epochs with the reference's tenfold learning-rate cuts at 50 % and 75 % of the run when
    asked (`:95-101`), timing steps from the fourth on (`:102,111`) and scoring validation and test
    accuracy after every step (`:113-114`)."""

    def __init__(self, data, n_hidden, n_layers, dropout, use_layernorm, lr, weight_decay,
                 self_loop=True, device=None, activation=None, init_params=None):
        import networkx as nx
        import numpy as np
        import torch
        import torch.nn.functional as F
        from .dgl_compat import DGLGraph
        from .nn import CrossEntropyLoss
        from .optim import Adam
        if not torch.cuda.is_available():
            raise RuntimeError('gist_amd: the GCN trainer runs on the HIP kernels and needs a GPU '
                               '(there is no CPU path)')
        self.device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        dev = self.device
        self.x = torch.as_tensor(np.asarray(data.features, np.float32)).to(dev)
        self.y = torch.as_tensor(np.asarray(data.labels, np.int64)).to(dev)
        self.masks = {k: torch.as_tensor(np.asarray(getattr(data, k + '_mask'), bool)).to(dev)
                      for k in ('train', 'val', 'test')}
        self.n_classes = int(data.num_labels)
        self.n_edges_raw = data.graph.number_of_edges()
        nxg = data.graph.copy()
        if self_loop:
            nxg.remove_edges_from(list(nx.selfloop_edges(nxg)))
            nxg.add_edges_from((v, v) for v in nxg.nodes())
        self.g = DGLGraph(nxg).to(dev)
        self.n_edges = self.g.number_of_edges()
        self.model = GCN(self.g, self.x.shape[1], n_hidden, self.n_classes, n_layers,
                         activation if activation is not None else F.relu, dropout, use_layernorm)
        if init_params is not None:
            with torch.no_grad():
                for conv, (W, b) in zip(self.model.layers, init_params):
                    conv.weight.copy_(torch.as_tensor(W))
                    conv.bias.copy_(torch.as_tensor(b))
        self.model = self.model.to(dev)
        self.criterion = CrossEntropyLoss()
        self.optimizer = Adam(self.model.parameters(), lr=lr, weight_decay=weight_decay)
        self.losses, self.history, self.step_seconds = [], [], []

    def mask_sizes(self):
        return {k: int(m.sum().item()) for k, m in self.masks.items()}

    def accuracy(self, which):
        """Share of the mask's nodes whose arg-max class is their label (`gcn/train.py:14-22`), eval mode."""
        import torch
        m = self.masks[which]
        self.model.eval()
        with torch.no_grad():
            hits = (self.model(self.x)[m].argmax(dim=1) == self.y[m]).sum()
        return float(hits.item()) / float(m.sum().item())

    def step(self):
        self.model.train()
        self.optimizer.zero_grad()
        m = self.masks['train']
        loss = self.criterion(self.model(self.x)[m], self.y[m])
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def fit(self, n_epochs, lr_scheduler=False, untimed=3):
        import time
        import torch
        cuts = {int(0.5 * n_epochs), int(0.75 * n_epochs)} if lr_scheduler else set()
        for epoch in range(n_epochs):
            if epoch in cuts:
                for group in self.optimizer.param_groups:
                    group['lr'] /= 10
            timed = epoch >= untimed
            if timed:
                torch.cuda.synchronize(self.device)
                t0 = time.time()
            loss = self.step()
            if timed:
                torch.cuda.synchronize(self.device)
                self.step_seconds.append(time.time() - t0)
            self.losses.append(loss)
            self.history.append((self.accuracy('val'), self.accuracy('test')))
        return self
