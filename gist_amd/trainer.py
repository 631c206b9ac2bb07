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
    with the current parameters, accuracy over a mask.  Shares the training arena.

    Inference-only layout, sized so that the ultra-wide model (H = 32768: cluster_gcn_ist_
    ultra_wide.py moves this evaluation to the CPU, :500-504) runs in HBM: two [N, H] activation
    buffers (ping-pong) and ONE row block of the concatenated operand [rows, 2*in] -- a layer is
    evaluated block of rows by block of rows (aggregate the block's rows over the full graph,
    project, normalise), never materialising [N, 2*in].  The narrowing class layer aggregates
    its C-wide projection instead of the H-wide activations ([h | A^h] W^T = h W1^T + A^(h W2^T)).
    Reddit at H = 32768: 2 x 30.5 GB + 4 GB instead of ~250 GB."""

    def __init__(self, g, dims, use_layernorm, arena, device, row_block=None,
                 block_bytes=4 << 30):
        self.g = g if g.device == device else g.to(device)
        self.dims = [(int(i), int(o)) for i, o in dims]
        self.use_layernorm = bool(use_layernorm)
        self.arena, self.device = arena, device
        n = self.g.number_of_nodes()
        self.n = n
        self.feat = self.g.ndata['feat']
        lab = self.g.ndata['label']
        self.labels = (lab if lab.dtype == torch.int32 else lab.to(torch.int32)).contiguous()
        self.norm = self.g.norm()
        f32 = dict(dtype=torch.float32, device=device)
        L1 = len(self.dims)
        self.n_classes = self.dims[-1][1]
        self.ldc = (self.n_classes + 3) // 4 * 4
        # the class layer is evaluated projection-first when it narrows (always, in practice)
        self.project_first = self.dims[-1][1] < self.dims[-1][0]
        blocked = self.dims[:-1] if self.project_first else self.dims
        max_in = max([i for (i, o) in blocked] + [1])
        max_out = max([o for (i, o) in blocked] + [1])
        if row_block is None:
            row_block = max(4096, int(block_bytes // (8 * max_in)))
        self.row_block = int(min(max(row_block, 1), n))
        hidden = max([o for (i, o) in self.dims[:-1]] + [1])
        self.h = [torch.empty(n, hidden, **f32) for _ in range(2 if L1 > 2 else 1)] if L1 > 1 else []
        self.zb = torch.empty(self.row_block, 2 * max_in, **f32) if blocked else None
        self.yb = torch.empty(self.row_block, max_out, **f32) if blocked else None
        self.logits = torch.empty(n, self.ldc, **f32)
        self.pbuf = torch.empty(n, self.ldc, **f32) if self.project_first else None
        self.correct = torch.zeros(1, dtype=torch.int32, device=device)
        need = 0
        L = hip._lib.load()
        for (i, o) in self.dims:
            need = max(need, L.gist_gemm_workspace_bytes(self.row_block, o, 2 * i),
                       L.gist_gemm_workspace_bytes(n, o, i))
        hip.workspace(need, device)
        self.masks = {}

    def forward(self):
        """GCN.forward (modules.py:310-314) in eval mode over the full graph -> logits [N, C]."""
        g, A, n = self.g, self.arena, self.n
        L1 = len(self.dims)
        cur = self.feat
        for k, (i, o) in enumerate(self.dims):
            last = k == L1 - 1
            W, b = A.W[k], A.b[k]
            if last and self.project_first:
                p = self.pbuf[:, :o]
                out = self.logits[:, :o]
                hip.gemm_nt(cur[:, :i], W[:, i:], None, p)
                hip.gemm_nt(cur[:, :i], W[:, :i], b, out)
                hip.spmm(g.rowptr, g.col, p, out, out_scale=self.norm, accumulate=True)
                break
            dst = self.logits if last else self.h[k % len(self.h)]
            for r0 in range(0, n, self.row_block):
                r1 = min(r0 + self.row_block, n)
                z = self.zb[:r1 - r0, :2 * i]
                hip.block_gather(cur[r0:r1, :i], None, None, z[:, :i])
                hip.spmm(g.rowptr[r0:r1 + 1], g.col, cur[:, :i], z[:, i:], out_scale=self.norm[r0:r1])
                if last:
                    hip.gemm_nt(z, W, b, dst[r0:r1, :o])
                else:
                    y = self.yb[:r1 - r0, :o]
                    hip.gemm_nt(z, W, b, y)
                    hip.ln_relu_fwd(y, dst[r0:r1, :o], None, self.use_layernorm, True)
            cur = dst
        return self.logits[:, :self.n_classes]

    def accuracy(self, mask_name):
        if mask_name not in self.masks:
            m = self.g.ndata[mask_name].to(torch.uint8).contiguous()
            self.masks[mask_name] = (m, int(m.sum().item()))
        m, total = self.masks[mask_name]
        if total == 0:
            return -1
        logits = self.forward()
        self.correct.zero_()
        hip.argmax_correct(logits, self.labels, m, self.correct)
        return self.correct.item() / total


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
