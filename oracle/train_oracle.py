"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's two training loops.

  run_cluster_gcn  cluster_gcn/cluster_gcn.py:19-142      (single device Cluster-GCN)
  run_gist         cluster_gcn/cluster_gcn_ist_distrib.py:370-479 + :71-367, all S
                   ranks simulated in ONE process (rank r's python-`random` stream
                   is identical on every rank, :570-572, so one stream suffices)

Built from oracle/gist_oracle.py; pinned against tests/golden/G6_e2e_*.npz.
dropout is fixed at 0 (torch's Philox stream is not restated).
"""
import copy
import random as _pyrandom
import time

import numpy as np

from . import gist_oracle as O


class TrainGraph(object):
    """Train-induced graph + features resident on the host (sampler.py:34)."""

    def __init__(self, rowptr, col, feat, label):
        self.rowptr, self.col = rowptr, col
        self.t_rowptr, self.t_col = O.transpose_csr(rowptr, col)
        self.feat, self.label = feat, label
        self.remap = np.full(rowptr.shape[0] - 1, -1, np.int64)

    def batch(self, ids):
        rp, cl = O.induced_subgraph(self.rowptr, self.col, ids, remap=self.remap)
        trp, tcl = O.induced_subgraph(self.t_rowptr, self.t_col, ids, remap=self.remap)
        return rp, cl, trp, tcl, self.feat[ids], self.label[ids]


def full_forward(rowptr, col, feat, params, use_layernorm):
    """utils.evaluate's model(g) in eval mode, cluster_gcn/utils.py:70-73."""
    logits, _ = O.gcn_forward(rowptr, col, feat, params, use_layernorm)
    return logits


def evaluate(rowptr, col, feat, label, mask, params, use_layernorm):
    """cluster_gcn/utils.py:70-80 (acc; micro-F1 is identical for argmax)."""
    logits = full_forward(rowptr, col, feat, params, use_layernorm)
    return O.calc_acc(label[mask], logits[mask])


def run_cluster_gcn(full, masks, par_li, psize, batch_size, params, use_layernorm,
                    lr, n_epochs, weight_decay=0.0, rng=_pyrandom, eval_fn=None,
                    on_epoch=None):
    """cluster_gcn/cluster_gcn.py:46-136.  `full` = (rowptr, col, feat, label) of the
    whole graph; masks = (train, val, test).  params are updated in place."""
    f_rowptr, f_col, f_feat, f_label = full
    train_mask, val_mask, test_mask = masks
    train_nid = np.nonzero(train_mask)[0].astype(np.int64)
    tr_rowptr, tr_col = O.induced_subgraph(f_rowptr, f_col, train_nid)      # sampler.py:34
    tg = TrainGraph(tr_rowptr, tr_col, f_feat[train_nid], f_label[train_nid])
    it = O.ClusterIterOracle(par_li, psize, batch_size, rng)                 # :50-52
    opt = O.new_opt_state(params)                                            # :78-80
    total_time, val_accs, test_accs = 0.0, [], []
    for epoch in range(n_epochs):                                            # :89
        t0 = time.time()
        for ids in it:                                                       # :92-105
            rp, cl, trp, tcl, x, y = tg.batch(ids)
            O.train_step(rp, cl, trp, tcl, x, y, params, opt, use_layernorm, lr,
                         weight_decay=weight_decay)
        total_time += time.time() - t0                                       # :106-108
        if on_epoch is not None:
            on_epoch(epoch, params)
        if eval_fn is not None:
            val_accs.append(eval_fn(params, val_mask))                       # :121-127
            test_accs.append(eval_fn(params, test_mask))
    return dict(total_time=total_time, val_accs=val_accs, test_accs=test_accs)


def run_gist(full, masks, par_li, psize, batch_size, base, S, n_hidden, n_layers,
             use_layernorm, lr, n_epochs, iter_per_site, weight_decay=0.0,
             rng=_pyrandom, eval_fn=None, on_sync=None):
    """GIST training with all S sites simulated sequentially.

    cluster_gcn/cluster_gcn_ist_distrib.py: ClusterIter first (:507-509, one
    shuffle), then ini_sync_dispatch_model (:595, L shuffles), then train()
    (:370-479).  base = full-width [(W,b)] (rank 0's base_model), updated in place.
    Returns per-site loss lists and the event log.
    """
    f_rowptr, f_col, f_feat, f_label = full
    train_mask, val_mask, test_mask = masks
    train_nid = np.nonzero(train_mask)[0].astype(np.int64)
    tr_rowptr, tr_col = O.induced_subgraph(f_rowptr, f_col, train_nid)
    tg = TrainGraph(tr_rowptr, tr_col, f_feat[train_nid], f_label[train_nid])
    it = O.ClusterIterOracle(par_li, psize, batch_size, rng)
    part = O.sample_partitions(n_layers, S, n_hidden, rng)                   # :199
    subs = [O.dispatch_site(base, part, s) for s in range(S)]               # :203-283
    local_epochs = n_epochs // S                                             # :385
    losses = [[] for _ in range(S)]
    events, val_accs, test_accs = [], [], []
    total_iter = 0
    opts = [None] * S
    n_iters = len(it)
    for e in range(local_epochs):
        run_eval = True
        for j, ids in enumerate(it):
            if total_iter % iter_per_site == 0:                              # :400
                if e > 0:
                    events.append('dispatch')
                    part = O.sample_partitions(n_layers, S, n_hidden, rng)   # :287
                    subs = [O.dispatch_site(base, part, s) for s in range(S)]
                opts = [O.new_opt_state(subs[s]) for s in range(S)]         # :405-407
            rp, cl, trp, tcl, x, y = tg.batch(ids)
            for s in range(S):                                               # :408-417
                loss, _, _ = O.train_step(rp, cl, trp, tcl, x, y, subs[s], opts[s],
                                          use_layernorm, lr, weight_decay=weight_decay)
                losses[s].append(loss)
            events.append('step')
            total_iter += 1
            last = (j == n_iters - 1) and (e == local_epochs - 1)
            if total_iter % iter_per_site == 0 or last:                      # :422-427
                events.append('sync')
                O.sync_sites(base, subs, part)
                # every site's copy of the shared last bias becomes the mean (:103)
                for s in range(S):
                    subs[s][n_layers] = (subs[s][n_layers][0], base[n_layers][1].copy())
                if on_sync is not None:
                    on_sync(base)
                if run_eval or last:                                         # :431-450
                    run_eval = False
                    events.append('eval')
                    if eval_fn is not None:
                        val_accs.append(eval_fn(base, val_mask))
                        test_accs.append(eval_fn(base, test_mask))
    return dict(losses=losses, events=events, val_accs=val_accs, test_accs=test_accs)
