"""TEST INFRASTRUCTURE ONLY -- numpy (+ C/OpenMP helper) restatement of GIST's hot path.

This file is the ORACLE: the parity checker for the HIP kernels in gist_amd/csrc
and the CPU baseline bench.py reports.  It is never imported by the product
package.  Every function cites the reference lines it restates (paths relative
to /root/reference).

Pinning status: PINNED -- every function below is checked in
tests/test_oracle_golden.py against vectors recorded from the reference's own
code (oracle/gen_golden.py -> tests/golden/*.npz).  Third-party arithmetic the
reference borrows (DGL 0.5.3 gSpMM / in_degrees / subgraph, torch 1.6 Linear /
LayerNorm / CrossEntropyLoss / Adam) is restated from its published semantics;
see SURVEY.md section 8c.

All floating point is float32 (the reference runs fp32 end to end); indices are
int64 on the host like the reference's `g.long()` graph.
"""
import ctypes
import os
import random as _pyrandom

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    """C helper (oracle/csrc/gist_oracle.c); built on demand with gcc."""
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'libgist_oracle.so')
        if not os.path.exists(path):
            from . import build as _b
            _b.build()
        L = ctypes.CDLL(path)
        p, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        L.oracle_spmm_csr_f32.argtypes = [p, p, p, i64, p, i64, i64, i64, p, p, ci]
        L.oracle_induced_count.argtypes = [p, p, p, i64, p, p]
        L.oracle_induced_fill.argtypes = [p, p, p, i64, p, p, p]
        L.oracle_transpose_csr.argtypes = [p, p, i64, p, p]
        for f in (L.oracle_spmm_csr_f32, L.oracle_induced_count,
                  L.oracle_induced_fill, L.oracle_transpose_csr):
            f.restype = None
        _LIB = L
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ----------------------------------------------------------------------------
# graph primitives
# ----------------------------------------------------------------------------

def in_degree_norm(rowptr):
    """1/in_degree with inf -> 0.  cluster_gcn/modules.py:239-243 (get_norm)."""
    deg = np.diff(_i64(rowptr)).astype(np.float32)
    with np.errstate(divide='ignore'):
        r = np.float32(1.0) / deg
    r[np.isinf(r)] = 0
    return r.astype(np.float32)


def spmm_sum(rowptr, col, x, out_scale=None, src_scale=None, out=None,
             accumulate=False):
    """y[v] = out_scale[v] * sum_{e in row v} src_scale[col[e]] * x[col[e]].

    Forward use: update_all(copy_src, sum) then `* norm`
    (cluster_gcn/modules.py:223-226; cluster_gcn/sampler.py:64-67).
    Backward use (autograd of the same op): the same sum over the transposed
    CSR with src_scale = norm (dH line).
    """
    rowptr, col = _i64(rowptr), _i64(col)
    assert x.dtype == np.float32 and x.ndim == 2 and x.strides[1] == 4
    n, d = rowptr.shape[0] - 1, x.shape[1]
    if out is None:
        out = np.zeros((n, d), np.float32)
        accumulate = False
    assert out.dtype == np.float32 and out.strides[1] == 4
    os_ = None if out_scale is None else _f32(out_scale)
    ss_ = None if src_scale is None else _f32(src_scale)
    _lib().oracle_spmm_csr_f32(_ptr(rowptr), _ptr(col), _ptr(x), x.strides[0] // 4,
                               _ptr(out), out.strides[0] // 4, n, d,
                               _ptr(os_), _ptr(ss_), int(bool(accumulate)))
    return out


def spmm_sum_py(rowptr, col, x):
    """Pure-numpy twin of spmm_sum (no scales) used to cross-check the C helper."""
    n = len(rowptr) - 1
    y = np.zeros((n, x.shape[1]), np.float32)
    for v in range(n):
        acc = np.zeros(x.shape[1], np.float32)
        for e in range(rowptr[v], rowptr[v + 1]):
            acc = acc + x[col[e]]
        y[v] = acc
    return y


def transpose_csr(rowptr, col):
    """CSR of the reversed graph (what autograd of update_all walks)."""
    rowptr, col = _i64(rowptr), _i64(col)
    n = rowptr.shape[0] - 1
    t_rowptr = np.zeros(n + 1, np.int64)
    t_col = np.zeros(col.shape[0], np.int64)
    _lib().oracle_transpose_csr(_ptr(rowptr), _ptr(col), n, _ptr(t_rowptr), _ptr(t_col))
    return t_rowptr, t_col


def csr_from_edges(src, dst, n):
    """In-edge CSR (row = destination) keeping the original edge order in a row."""
    src, dst = _i64(src), _i64(dst)
    order = np.argsort(dst, kind='stable')
    rowptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(dst, minlength=n), out=rowptr[1:])
    return rowptr, src[order].copy()


def induced_subgraph(rowptr, col, ids, remap=None):
    """Node-induced subgraph with node i of the result = ids[i].

    cluster_gcn/partition_utils.py:20-25 (`g.subgraph(concat(parts))`) and
    cluster_gcn/sampler.py:34 (`g.subgraph(seed_nid)`): edges whose two ends are
    both selected are kept (inter-part edges included), relabelled by position
    in `ids`, original order kept.
    """
    rowptr, col, ids = _i64(rowptr), _i64(col), _i64(ids).reshape(-1)
    n_full, nb = rowptr.shape[0] - 1, ids.shape[0]
    own = remap is None
    if own:
        remap = np.full(n_full, -1, np.int64)
    deg = np.zeros(nb, np.int64)
    L = _lib()
    L.oracle_induced_count(_ptr(rowptr), _ptr(col), _ptr(ids), nb, _ptr(remap), _ptr(deg))
    sub_rowptr = np.zeros(nb + 1, np.int64)
    np.cumsum(deg, out=sub_rowptr[1:])
    sub_col = np.zeros(int(sub_rowptr[-1]), np.int64)
    L.oracle_induced_fill(_ptr(rowptr), _ptr(col), _ptr(ids), nb, _ptr(remap),
                          _ptr(sub_rowptr), _ptr(sub_col))
    if not own:
        remap[ids] = -1
    return sub_rowptr, sub_col


# ----------------------------------------------------------------------------
# one ISTSAGELayer  (cluster_gcn/modules.py:191-243)
# ----------------------------------------------------------------------------

LN_EPS = np.float32(1e-5)   # nn.LayerNorm default, modules.py:209


def sage_layer_forward(rowptr, col, h, W, b, use_lynorm, relu, drop_mask=None,
                       drop_p=0.0):
    """ISTSAGELayer.forward, cluster_gcn/modules.py:218-237.

    h [N,in]; W [out, 2*in]; b [out].  drop_mask (optional, {0,1} of shape
    [N, 2*in]) restates nn.Dropout on the CONCATENATED tensor (:227-231).
    Returns (out, cache).
    """
    h = _f32(h)
    norm = in_degree_norm(rowptr)                             # :222,239-243
    ah = spmm_sum(rowptr, col, h, out_scale=norm)             # :223-226
    z = np.concatenate([h, ah], axis=1)                       # :227
    if drop_mask is not None:                                 # :230-231
        z = (z * drop_mask * np.float32(1.0 / (1.0 - drop_p))).astype(np.float32)
    y = z @ W.T + b                                           # :233
    cache = dict(z=z, W=W, norm=norm, use_lynorm=use_lynorm, relu=relu,
                 drop_mask=drop_mask, drop_p=drop_p, n_in=h.shape[1])
    if use_lynorm:                                            # :209,234
        mu = y.mean(axis=1, keepdims=True, dtype=np.float32)
        var = ((y - mu) ** 2).mean(axis=1, keepdims=True, dtype=np.float32)
        rstd = (np.float32(1.0) / np.sqrt(var + LN_EPS)).astype(np.float32)
        yhat = ((y - mu) * rstd).astype(np.float32)
        cache.update(yhat=yhat, rstd=rstd)
    else:
        yhat = y.astype(np.float32)
    out = np.maximum(yhat, 0) if relu else yhat               # :235-236
    cache['out'] = out
    return out.astype(np.float32), cache


def sage_layer_backward(cache, d_out, t_rowptr, t_col, need_dh=True):
    """Autograd of ISTSAGELayer.forward (backward block).

    t_rowptr/t_col = CSR of the reversed graph.  Returns (dh, dW, db).
    """
    g = _f32(d_out)
    if cache['relu']:
        g = g * (cache['out'] > 0)
    if cache['use_lynorm']:
        yhat, rstd = cache['yhat'], cache['rstd']
        m1 = g.mean(axis=1, keepdims=True, dtype=np.float32)
        m2 = (g * yhat).mean(axis=1, keepdims=True, dtype=np.float32)
        g = (rstd * (g - m1 - yhat * m2)).astype(np.float32)
    g = _f32(g)
    dW = g.T @ cache['z']
    db = g.sum(axis=0, dtype=np.float32)
    if not need_dh:
        return None, dW.astype(np.float32), db
    dz = g @ cache['W']
    if cache['drop_mask'] is not None:
        dz = dz * cache['drop_mask'] * np.float32(1.0 / (1.0 - cache['drop_p']))
    dz = _f32(dz)
    n_in = cache['n_in']
    dh = np.ascontiguousarray(dz[:, :n_in])
    spmm_sum(t_rowptr, t_col, dz[:, n_in:], src_scale=cache['norm'], out=dh,
             accumulate=True)
    return dh, dW.astype(np.float32), db


# ----------------------------------------------------------------------------
# GCN model (cluster_gcn/modules.py:245-314)
# ----------------------------------------------------------------------------

def gcn_layer_dims(in_feats, n_hidden, n_classes, n_layers, split_output=False,
                   num_subnet=1):
    """(in, out, use_lynorm_allowed, relu) per layer for split_input=False.

    cluster_gcn/modules.py:245-308.  Full model = (False, False, 1); GIST
    sub-model = (False, True, S).
    """
    hs = n_hidden // num_subnet
    dims = []
    if n_layers <= 1 and not split_output:
        dims.append((in_feats, n_hidden))                     # :274-278
    else:
        dims.append((in_feats, hs))                           # :279-283
    for i in range(n_layers - 1):
        if i == n_layers - 2 and not split_output:
            dims.append((hs, n_hidden))                       # :288-292
        else:
            dims.append((hs, hs))                             # :293-297
    if split_output:
        dims.append((hs, n_classes))                          # :300-304
    else:
        dims.append((n_hidden, n_classes))                    # :305-308
    out = []
    for k, (i, o) in enumerate(dims):
        last = k == len(dims) - 1
        out.append((i, o, not last, not last))
    return out


def gcn_forward(rowptr, col, feat, params, use_layernorm, drop_masks=None,
                drop_p=0.0):
    """GCN.forward, cluster_gcn/modules.py:310-314.  params = [(W,b), ...]."""
    h = _f32(feat)
    caches = []
    L = len(params)
    for k, (W, b) in enumerate(params):
        last = k == L - 1
        dm = None if drop_masks is None else drop_masks[k]
        h, c = sage_layer_forward(rowptr, col, h, W, b,
                                  use_lynorm=(use_layernorm and not last),
                                  relu=not last, drop_mask=dm, drop_p=drop_p)
        caches.append(c)
    return h, caches


def gcn_backward(caches, d_logits, t_rowptr, t_col):
    grads = [None] * len(caches)
    g = d_logits
    for k in range(len(caches) - 1, -1, -1):
        g, dW, db = sage_layer_backward(caches[k], g, t_rowptr, t_col, need_dh=(k > 0))
        grads[k] = (dW, db)
    return grads


# ----------------------------------------------------------------------------
# loss / optimiser (cluster_gcn_ist_distrib.py:384,405-417; cluster_gcn.py:76-105)
# ----------------------------------------------------------------------------

def cross_entropy(logits, labels, mask=None):
    """nn.CrossEntropyLoss() (mean) over rows with mask; returns (loss, dlogits).

    cluster_gcn/cluster_gcn_ist_distrib.py:384,411-414.
    """
    logits = _f32(logits)
    n = logits.shape[0]
    if mask is None:
        mask = np.ones(n, bool)
    mask = np.asarray(mask, bool)
    cnt = int(mask.sum())
    mx = logits.max(axis=1, keepdims=True)
    ex = np.exp(logits - mx, dtype=np.float32)
    se = ex.sum(axis=1, keepdims=True, dtype=np.float32)
    logp = (logits - mx) - np.log(se, dtype=np.float32)
    rows = np.arange(n)
    nll = -logp[rows, labels]
    loss = np.float32(nll[mask].sum(dtype=np.float32) / np.float32(cnt))
    d = (ex / se).astype(np.float32)
    d[rows, labels] -= 1
    d = d * (mask[:, None] / np.float32(cnt))
    return loss, d.astype(np.float32)


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8,
              weight_decay=0.0):
    """torch.optim.Adam (coupled L2), one tensor, in place; `step` is 1-based.

    cluster_gcn/cluster_gcn_ist_distrib.py:405-407,417.
    """
    f = np.float32
    if weight_decay != 0:
        g = g + f(weight_decay) * p
    m *= f(beta1)
    m += f(1 - beta1) * g
    v *= f(beta2)
    v += f(1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = np.sqrt(v) / f(np.sqrt(bc2)) + f(eps)
    p -= f(lr / bc1) * (m / denom)
    return p


def train_step(rowptr, col, t_rowptr, t_col, feat, labels, params, opt_state,
               use_layernorm, lr, weight_decay=0.0, mask=None, drop_masks=None,
               drop_p=0.0):
    """One iteration of the training loop: forward, CE, backward, Adam.

    cluster_gcn/cluster_gcn_ist_distrib.py:408-417 / cluster_gcn/cluster_gcn.py:98-105.
    opt_state = dict(step=int, m=[(mW,mb)...], v=[(vW,vb)...]); updated in place.
    """
    logits, caches = gcn_forward(rowptr, col, feat, params, use_layernorm,
                                 drop_masks=drop_masks, drop_p=drop_p)
    loss, dlog = cross_entropy(logits, labels, mask)
    grads = gcn_backward(caches, dlog, t_rowptr, t_col)
    opt_state['step'] += 1
    t = opt_state['step']
    for k, ((W, b), (dW, db)) in enumerate(zip(params, grads)):
        adam_step(W, dW, opt_state['m'][k][0], opt_state['v'][k][0], t, lr,
                  weight_decay=weight_decay)
        adam_step(b, db, opt_state['m'][k][1], opt_state['v'][k][1], t, lr,
                  weight_decay=weight_decay)
    return loss, logits, grads


def new_opt_state(params):
    return dict(step=0,
                m=[(np.zeros_like(W), np.zeros_like(b)) for W, b in params],
                v=[(np.zeros_like(W), np.zeros_like(b)) for W, b in params])


# ----------------------------------------------------------------------------
# cluster batch source (cluster_gcn/sampler.py, cluster_gcn/partition_utils.py)
# ----------------------------------------------------------------------------

class ClusterIterOracle(object):
    """ClusterIter, cluster_gcn/sampler.py:15-56,85-93, with partition lists given.

    Consumes python's global `random` exactly as the reference does: one
    shuffle at construction (:55) and one at every epoch end (:92).
    Yields the concatenated node ids of a batch (partition_utils.py:21-24).
    """

    def __init__(self, par_li, psize, batch_size, rng=_pyrandom):
        self.par_li = par_li
        self.psize = psize
        self.batch_size = batch_size
        self.max = int(psize // batch_size)                   # :54
        self.rng = rng
        rng.shuffle(self.par_li)                              # :55

    def __len__(self):
        return self.max

    def __iter__(self):
        self.n = 0
        return self

    def batch_ids(self, i):
        parts = [self.par_li[s] for s in range(i * self.batch_size,
                                               (i + 1) * self.batch_size)
                 if s < self.psize]                           # partition_utils.py:21-22
        return np.concatenate(parts).reshape(-1).astype(np.int64)

    def __next__(self):
        if self.n < self.max:                                 # :86-90
            ids = self.batch_ids(self.n)
            self.n += 1
            return ids
        self.rng.shuffle(self.par_li)                         # :92
        raise StopIteration


# ----------------------------------------------------------------------------
# IST partition / dispatch / sync (cluster_gcn/cluster_gcn_ist_distrib.py:51-367)
# ----------------------------------------------------------------------------

def create_partition(num_subnet, size, rng=_pyrandom):
    """cluster_gcn/cluster_gcn_ist_distrib.py:51-65.  Returns [(idx, full_idx)] per site."""
    possible = [x for x in range(size)]
    rng.shuffle(possible)                                     # :53
    lists = [[] for _ in range(num_subnet)]
    for i in range(size):                                     # :55-58
        lists[i % num_subnet].append(possible[i])
    out = []
    for idx in lists:                                         # :60-64
        idx = np.asarray(idx, np.int64)
        out.append((idx, np.concatenate([idx, idx + size])))
    return out


def sample_partitions(n_layers, num_subnet, n_hidden, rng=_pyrandom):
    """DistributedGNNWrapper.sample_partitions, :93-98 (one per hidden layer)."""
    return [create_partition(num_subnet, n_hidden, rng) for _ in range(n_layers)]


def dispatch_site(base, part, site):
    """Slice the base model into site `site`'s sub-model.

    ini_sync_dispatch_model / dispatch_model, :203-226 and :291-313 (rank 0's own
    copy) and the equivalent broadcast payloads :231-283 / :315-365.
    base = [(W,b)] full model; returns [(W,b)] copies.
    """
    L = len(part)
    sub = []
    for k in range(L + 1):
        W, b = base[k]
        if k == 0:
            idx, _ = part[0][site]
            sub.append((W[idx, :].copy(), b[idx].copy()))                 # :205-209
        elif k == L:
            _, full = part[-1][site]
            sub.append((W[:, full].copy(), b.copy()))                     # :210-217
        else:
            _, full_prev = part[k - 1][site]
            nxt, _ = part[k][site]
            sub.append((W[:, full_prev][nxt, :].copy(), b[nxt].copy()))   # :218-226
    return sub


def sync_sites(base, subs, part):
    """sync_model, :100-195: write every site's blocks back; last bias = mean.

    The all-reduce (:103, :38-41) is SUM then divide by S; numpy sums in site
    order.  base is updated in place.
    """
    S, L = len(subs), len(part)
    bl = subs[0][L][1].copy()
    for s in range(1, S):
        bl = bl + subs[s][L][1]
    bl = (bl / np.float32(S)).astype(np.float32)
    for s in range(S):
        for k in range(L + 1):
            W, b = base[k]
            sW, sb = subs[s][k]
            if k == 0:
                idx, _ = part[0][s]
                W[idx, :] = sW                                            # :111-114
                b[idx] = sb
            elif k == L:
                _, full = part[-1][s]
                W[:, full] = sW                                           # :118-119
            else:
                _, full_prev = part[k - 1][s]
                nxt, _ = part[k][s]
                W[np.ix_(nxt, full_prev)] = sW                            # :127-133
                b[nxt] = sb
    base[L][1][:] = bl                                                    # :122-123
    return base


def ist_schedule(n_epochs, num_subnet, iters_per_epoch, iter_per_site):
    """The (total_iter, action) sequence of train(), :385-427.

    Restates the control flow quirks: local_epochs = n_epochs // S (:385);
    fresh Adam at every total_iter % iter_per_site == 0 (:400-407); re-dispatch
    only when e > 0 (:401-403); sync at multiples of iter_per_site and at the
    very last iteration (:422-427); eval at the first sync of each epoch and at
    the end (:431-450).
    """
    ev = []
    local_epochs = n_epochs // num_subnet
    total = 0
    for e in range(local_epochs):
        run_eval = True
        for j in range(iters_per_epoch):
            if total % iter_per_site == 0:
                if e > 0:
                    ev.append((total, 'dispatch'))
                ev.append((total, 'new_adam'))
            ev.append((total, 'step'))
            total += 1
            last = (j == iters_per_epoch - 1) and (e == local_epochs - 1)
            if total % iter_per_site == 0 or last:
                ev.append((total, 'sync'))
                if run_eval or last:
                    run_eval = False
                    ev.append((total, 'eval'))
    return ev


# ----------------------------------------------------------------------------
# evaluation (cluster_gcn/utils.py:47-80)
# ----------------------------------------------------------------------------

def calc_acc(y_true, logits):
    """utils.py:58-67 (== micro-F1 for single-label argmax, utils.py:47-56)."""
    pred = np.argmax(logits, axis=1)
    if pred.shape[0] == 0:
        return -1
    return float((pred == y_true).sum() / pred.shape[0])


# ----------------------------------------------------------------------------
# data preparation (cluster_gcn_ist_distrib.py:492-499, sampler.py:58-69)
# ----------------------------------------------------------------------------

def standard_scaler(feats, fit_mask):
    """sklearn.preprocessing.StandardScaler().fit(feats[fit_mask]).transform(feats) as the
    reference applies it (cluster_gcn/cluster_gcn_ist_distrib.py:492-499): float64 mean and
    POPULATION variance per column over the train rows, scale = sqrt(var) with 0 -> 1, and
    sklearn's in-place float32 transform `X -= mean_; X /= scale_` (two roundings).  sklearn is
    a third-party dependency: tests/test_oracle_golden.py checks this restatement against the
    installed sklearn itself.  Returns (scaled float32 [N,F], mean, var)."""
    x = _f32(feats)
    fit = x[np.asarray(fit_mask, bool)].astype(np.float64)
    mean = fit.mean(axis=0)
    var = ((fit - mean) ** 2).mean(axis=0)
    scale = np.sqrt(var)
    scale[scale == 0.0] = 1.0
    out = (x.astype(np.float64) - mean).astype(np.float32)
    out = (out.astype(np.float64) / scale).astype(np.float32)
    return out, mean, var


def preaggregate(rowptr, col, feats):
    """ClusterIter.precalc (cluster_gcn/sampler.py:58-69, --use-pp): [X | A^ X] with
    A^ = diag(1/in_deg) A on the TRAIN graph (zero-degree rows -> 0)."""
    x = _f32(feats)
    return np.concatenate([x, spmm_sum(rowptr, col, x, out_scale=in_degree_norm(rowptr))], axis=1)


# ----------------------------------------------------------------------------
# small-graph GCN layer (gcn/gcn.py; DGL GraphConv, [DGL recalled] -- UNPINNED)
# ----------------------------------------------------------------------------

def graphconv_forward(rowptr, col, out_deg, x, W, b, relu):
    """dgl.nn.pytorch.GraphConv(norm='both') as used by gcn/gcn.py:30-56.

    parity unpinned: DGL is not in /root/reference; restated from its
    documented behaviour (degrees clamped >= 1, W is [in,out], multiply by W
    first iff in > out).
    """
    x = _f32(x)
    in_deg = np.maximum(np.diff(_i64(rowptr)), 1).astype(np.float32)
    ns = (np.maximum(out_deg, 1).astype(np.float32)) ** np.float32(-0.5)
    nd = in_deg ** np.float32(-0.5)
    if W.shape[0] > W.shape[1]:
        y = spmm_sum(rowptr, col, _f32(x @ W), out_scale=nd, src_scale=ns)
    else:
        y = spmm_sum(rowptr, col, x, out_scale=nd, src_scale=ns) @ W
    y = _f32(y + b)
    return np.maximum(y, 0) if relu else y


def whole_tensor_layer_norm(h):
    """F.layer_norm(h, h.shape), gcn/gcn.py:65-66: normalises over the WHOLE tensor."""
    mu = h.mean(dtype=np.float32)
    var = ((h - mu) ** 2).mean(dtype=np.float32)
    return ((h - mu) / np.sqrt(var + LN_EPS)).astype(np.float32)
