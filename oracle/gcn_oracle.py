"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the small full-graph GCN path
(BASELINE config 1, "Cora plumbing"): gcn/gcn.py GCN + the training loop of gcn/train.py.

Never imported by the product package.  Every function cites the reference lines it
restates (paths relative to /root/reference).

Pinning status: the LOOP (what is trained on which rows, dropout placement, whole-tensor
layer norm, CE over the train mask, Adam with coupled weight decay, the lr schedule, the
accuracies) is PINNED by tests/golden/G5_train_*.npz, recorded from gcn/train.py main() run
unchanged (oracle/gen_golden.py G5t).  The GraphConv layer itself is DGL 0.5.3 code that is
not in the reference tree: its arithmetic (norm='both', degrees clamped >= 1, W [in,out],
W first iff in > out) is restated from DGL's documented behaviour -- the recorded run used the
same restatement (oracle/dgl_stub), so that layer's parity stays UNPINNED.
"""
import numpy as np

from . import gist_oracle as O

LN_EPS = np.float32(1e-5)


class CitationGraph(object):
    """In-edge CSR + reversed CSR + the two GraphConv scales of a small graph."""

    def __init__(self, src, dst, n):
        self.n = int(n)
        self.rowptr, self.col = O.csr_from_edges(src, dst, n)
        self.t_rowptr, self.t_col = O.transpose_csr(self.rowptr, self.col)
        in_deg = np.maximum(np.diff(self.rowptr), 1).astype(np.float32)
        out_deg = np.maximum(np.diff(self.t_rowptr), 1).astype(np.float32)
        self.nd = in_deg ** np.float32(-0.5)          # destination side, D_in^-1/2
        self.ns = out_deg ** np.float32(-0.5)         # source side, D_out^-1/2


def with_self_loops(src, dst, n):
    """gcn/train.py:66-68: drop existing self loops, add one per node."""
    keep = src != dst
    loops = np.arange(n, dtype=np.int64)
    return np.concatenate([src[keep], loops]), np.concatenate([dst[keep], loops])


def gcn_forward(g, x, params, use_layernorm, drop_masks=None, drop_p=0.0):
    """gcn/gcn.py:58-67 (GCN.forward) over dgl GraphConv layers (gcn/gcn.py:30-56).
    params = [(W [in,out], b [out])]; relu on all but the last layer; dropout on the INPUT of
    every layer but the first; whole-tensor layer norm after all but the last."""
    h = x.astype(np.float32)
    caches = []
    L = len(params)
    for k, (W, b) in enumerate(params):
        c = dict(k=k)
        if k > 0 and drop_masks is not None:                            # :62-63
            h = (h * drop_masks[k] * np.float32(1.0 / (1.0 - drop_p))).astype(np.float32)
            c['mask'] = drop_masks[k]
        u = (h * g.ns[:, None]).astype(np.float32)
        w_first = W.shape[0] > W.shape[1]
        if w_first:
            p = (u @ W).astype(np.float32)
            q = O.spmm_sum(g.rowptr, g.col, p)
            c.update(u=u)
        else:
            q0 = O.spmm_sum(g.rowptr, g.col, u)
            q = (q0 @ W).astype(np.float32)
            c.update(q0=q0)
        r = (q * g.nd[:, None] + b).astype(np.float32)
        relu = k < L - 1
        a = np.maximum(r, 0) if relu else r
        c.update(W=W, w_first=w_first, r=r, relu=relu)
        if use_layernorm and k < L - 1:                                  # :65-66
            mu = a.mean(dtype=np.float32)
            var = ((a - mu) ** 2).mean(dtype=np.float32)
            rstd = np.float32(1.0) / np.sqrt(var + LN_EPS)
            y = ((a - mu) * rstd).astype(np.float32)
            c.update(ln=True, y=y, rstd=rstd)
            h = y
        else:
            c.update(ln=False)
            h = a.astype(np.float32)
        caches.append(c)
    return h, caches


def gcn_backward(g, caches, d_logits, drop_p=0.0):
    """Autograd of gcn_forward: [(dW, db)] per layer."""
    grads = [None] * len(caches)
    d = d_logits.astype(np.float32)
    for c in reversed(caches):
        if c['ln']:
            y, rstd = c['y'], c['rstd']
            m1 = d.mean(dtype=np.float32)
            m2 = (d * y).mean(dtype=np.float32)
            d = ((d - m1 - y * m2) * rstd).astype(np.float32)
        if c['relu']:
            d = d * (c['r'] > 0)
        db = d.sum(axis=0, dtype=np.float32)
        dq = (d * g.nd[:, None]).astype(np.float32)
        W = c['W']
        if c['w_first']:
            dp = O.spmm_sum(g.t_rowptr, g.t_col, dq)
            dW = (c['u'].T @ dp).astype(np.float32)
            du = (dp @ W.T).astype(np.float32)
        else:
            dW = (c['q0'].T @ dq).astype(np.float32)
            du = O.spmm_sum(g.t_rowptr, g.t_col, (dq @ W.T).astype(np.float32))
        grads[c['k']] = (dW, db)
        d = (du * g.ns[:, None]).astype(np.float32)
        if 'mask' in c:
            d = (d * c['mask'] * np.float32(1.0 / (1.0 - drop_p))).astype(np.float32)
    return grads


def accuracy(logits, labels, mask):
    """gcn/train.py:14-22 evaluate(): argmax accuracy over the masked rows."""
    idx = np.nonzero(mask)[0]
    return float((logits[idx].argmax(1) == labels[idx]).sum()) / len(idx)


def train(g, feats, labels, train_mask, val_mask, test_mask, params, use_layernorm, lr,
          weight_decay, n_epochs, lr_scheduler=False):
    """gcn/train.py:93-123: full-graph Adam steps on CE over the train rows (dropout off:
    parity runs use p = 0), optional /10 lr drops at 50 % and 75 % (:95-101), val/test accuracy
    after every epoch.  Returns per-epoch losses, [(val, test)] records, final params."""
    params = [(W.copy(), b.copy()) for W, b in params]
    opt = O.new_opt_state(params)
    losses, record = [], []
    tmask = np.asarray(train_mask).astype(bool)
    for epoch in range(n_epochs):
        if lr_scheduler and epoch in (int(0.5 * n_epochs), int(0.75 * n_epochs)):
            lr = lr / 10
        logits, caches = gcn_forward(g, feats, params, use_layernorm)
        loss, dlog = O.cross_entropy(logits, labels, tmask)              # :106
        grads = gcn_backward(g, caches, dlog)
        opt['step'] += 1
        for k, ((W, b), (dW, db)) in enumerate(zip(params, grads)):
            O.adam_step(W, dW, opt['m'][k][0], opt['v'][k][0], opt['step'], lr,
                        weight_decay=weight_decay)
            O.adam_step(b, db, opt['m'][k][1], opt['v'][k][1], opt['step'], lr,
                        weight_decay=weight_decay)
        losses.append(float(loss))
        ev, _ = gcn_forward(g, feats, params, use_layernorm)            # :113-115 (eval mode)
        record.append((accuracy(ev, labels, val_mask), accuracy(ev, labels, test_mask)))
    return np.array(losses, np.float32), record, params
