"""Training loops of the reference, driven through the SageEngine fast path.

  ClusterGCNTrainer   cluster_gcn/cluster_gcn.py:19-142      (single GPU, full width)
  (GIST trainer lives in gist_amd/ist.py next to the dispatch/sync code)

Timing follows the reference: `total_time` sums the epoch loops -- batch construction
included, evaluation excluded (cluster_gcn.py:91,106-111).
"""
import time

import numpy as np
import torch

from . import hip
from .engine import SageEngine, Batch, dims_for
from .sampler import EngineClusterIter


def full_graph_batch(g, labels_i32):
    b = Batch()
    b.n = g.number_of_nodes()
    b.rowptr, b.col, b.t_rowptr, b.t_col = g.rowptr, g.col, g.t_rowptr, g.t_col
    b.norm, b.labels, b.ids = g.norm(), labels_i32, None
    return b


class FullGraphEvaluator(object):
    """utils.evaluate (cluster_gcn/utils.py:70-80): eval-mode forward over the WHOLE graph
    with the current parameters, accuracy over a mask.  Shares the training arena."""

    def __init__(self, g, dims, use_layernorm, arena, device):
        self.g = g if g.device == device else g.to(device)
        n = self.g.number_of_nodes()
        self.eng = SageEngine(dims, use_layernorm, 0.0, n_max=n, device=device, arena=arena)
        self.feat = self.g.ndata['feat']
        lab = self.g.ndata['label']
        self.labels = (lab if lab.dtype == torch.int32 else lab.to(torch.int32)).contiguous()
        self.batch = full_graph_batch(self.g, self.labels)
        self.masks = {}

    def accuracy(self, mask_name):
        if mask_name not in self.masks:
            m = self.g.ndata[mask_name].to(torch.uint8).contiguous()
            self.masks[mask_name] = (m, int(m.sum().item()))
        m, total = self.masks[mask_name]
        if total == 0:
            return -1
        n = self.batch.n
        hip.block_gather(self.feat, None, None, self.eng.z0_left(n))
        self.eng.forward(self.batch, training=False)
        self.eng.correct.zero_()
        self.eng.count_correct(self.batch, m)
        return self.eng.correct.item() / total


class ClusterGCNTrainer(object):
    def __init__(self, dataset_name, g, par_li, psize, batch_size, n_hidden, n_layers, n_classes,
                 dropout, use_layernorm, lr, weight_decay, device, seed=0, init_params=None,
                 module=None):
        """`g`: host Graph with ndata feat/label/masks.  Construction order mirrors
        cluster_gcn.py: ClusterIter (consumes one `random.shuffle`) then the model."""
        self.device = device
        train_nid = np.nonzero(g.ndata['train_mask'].numpy())[0].astype(np.int64)
        self.it = EngineClusterIter(dataset_name, g, psize, batch_size, train_nid, par_li=par_li,
                                    device=device)
        in_feats = g.ndata['feat'].shape[1]
        self.dims = dims_for(in_feats, n_hidden, n_classes, n_layers)
        self.engine = SageEngine(self.dims, use_layernorm, dropout, self.it.n_max, device, seed=seed)
        if module is not None:
            self.engine.arena.adopt_module(module)
        if init_params is not None:
            self.engine.arena.load(init_params)
        self.it.bind(self.engine)
        self.lr, self.wd = lr, weight_decay
        self.use_layernorm = use_layernorm
        self.g_host = g
        self.evaluator = None
        self.total_time = 0.0

    def train_epoch(self):
        """One pass of ClusterIter (cluster_gcn.py:92-105).  Returns the last loss tensor."""
        loss = None
        for batch in self.it:
            loss = self.engine.train_step(batch, self.lr, self.wd)
        return loss

    def timed_epoch(self):
        torch.cuda.synchronize(self.device)
        t0 = time.time()
        loss = self.train_epoch()
        torch.cuda.synchronize(self.device)
        self.total_time += time.time() - t0
        return loss

    def evaluate(self, mask_name):
        if self.evaluator is None:
            self.evaluator = FullGraphEvaluator(self.g_host, self.dims, self.use_layernorm,
                                                self.engine.arena, self.device)
        return self.evaluator.accuracy(mask_name)
