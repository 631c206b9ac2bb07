"""The drop-in module path on the fused step.

A script that keeps the reference's loop body (cluster_gcn/cluster_gcn.py:96-105,
cluster_gcn_ist_distrib.py:408-417)

    pred = model(cluster)
    loss = loss_f(pred[batch_train_mask], batch_labels[batch_train_mask])
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()

runs the SAME preallocated plan `gist_sage_step` runs for the engine path, as three C-ABI calls instead of one
(GIST_STEP_PHASE_FORWARD / _BACKWARD / _OPTIMIZER, include/gist_hip.h):

  * `GCN.forward` on a ClusterBatch (what ClusterIter yields) is ONE dispatcher op, `gist::gcn_forward`: extraction of the
    batch, the whole forward, and -- because the fused class layer produces them in the same launch -- the mean CE over
    the batch rows and its gradient w.r.t. the logits;
  * `gist_amd.nn.CrossEntropyLoss` recognises those logits and the batch's own labels and returns that loss (any other
    loss on `pred` is an ordinary autograd graph ending in `gist::gcn_backward` with GIST_STEP_DLOGITS_GIVEN);
  * `loss.backward()` is ONE dispatcher op, `gist::gcn_backward`: the backward pass into the gradient arena (complete on
    return: `p.grad` are views of it);
  * `gist_amd.optim.Adam.step()` is one launch over the flat arena (with the next batch's extraction in its grid).

The module's parameters are re-homed into the engine's arena (values preserved), so `model.parameters()`,
`state_dict()`, the IST block movers and the evaluator keep working on the same tensors.  Parameters after a step are
bitwise those of the engine path (tests/test_module_engine_gpu.py).  GIST_MODULE_ENGINE=0 turns the binding off (the
op-by-op module path of gist_amd/autograd.py, one dispatcher op per layer: the parity twin).
"""
import os
import weakref

import torch

from . import _lib, hip
from .engine import SageEngine

_REGISTRY = weakref.WeakValueDictionary()      # handle -> ModuleEngine (owned by its model)
_NEXT_HANDLE = [1]

_IDLE, _FWD_DONE, _BWD_DONE = 0, 1, 2


def _storage_uses(t):
    """Tensors (of any Python lifetime: views count) that share t's storage, + the probe itself."""
    return torch._C._storage_Use_Count(t.untyped_storage()._cdata)

_lib_def = torch.library.Library('gist', 'FRAGMENT')
_lib_def.define('gcn_forward(Tensor[] params, int handle, int token, int n, int ldc, bool train) -> Tensor')
_lib_def.define('gcn_backward(Tensor d_logits, int handle, int token, bool given) -> Tensor[]')


def _gcn_forward_cuda(params, handle, token, n, ldc, train):
    """GCN.forward (cluster_gcn/modules.py:310-314) on the pending cluster batch of module engine `handle`: returns the
    padded logits [n, ldc] (columns >= n_classes are padding)."""
    return _REGISTRY[handle]._run_forward(token, n, ldc, train)


def _gcn_backward_cuda(d_logits, handle, token, given):
    """The backward pass of the forward `token` into the gradient arena; returns its per-parameter views
    [dW_0, db_0, dW_1, ...].  given: d_logits [n, ldc] is the caller's gradient w.r.t. the padded logits (else the mean
    CE's, already formed by the forward)."""
    return _REGISTRY[handle]._run_backward(token, d_logits, given)


_lib_def.impl('gcn_forward', _gcn_forward_cuda, 'CUDA')
_lib_def.impl('gcn_backward', _gcn_backward_cuda, 'CUDA')


@torch.library.register_fake('gist::gcn_forward')
def _(params, handle, token, n, ldc, train):
    return params[0].new_empty(n, ldc)


@torch.library.register_fake('gist::gcn_backward')
def _(d_logits, handle, token, given):
    return [d_logits.new_empty(v.shape) for v in _REGISTRY[handle].grad_views]


class _GCNForward(torch.autograd.Function):
    """The tape entry of gist::gcn_forward: its backward is gist::gcn_backward."""

    # (the node knows its engine weakly: autograd hangs it on the output -- a view of the engine's own ring buffer --
    # and a strong reference would close engine -> buffer -> node -> engine through C++, where no collector looks)
    @staticmethod
    def forward(ctx, me, token, n, ldc, *params):
        ctx.me, ctx.token, ctx.n_in = weakref.ref(me), token, 4 + len(params)
        return torch.ops.gist.gcn_forward(list(params), me.handle, token, n, ldc, True)

    @staticmethod
    def backward(ctx, d_y):
        me = ctx.me()
        if me is None:
            raise RuntimeError('gist_amd: backward through the forward of a model that no longer exists')
        me.autograd_backward(ctx.token, d_y)
        return (None,) * ctx.n_in                      # (the gradients were delivered to p.grad: arena views)


class _FusedLoss(torch.autograd.Function):
    """mean CE over the batch rows, already computed by the forward's class-layer launch."""

    @staticmethod
    def forward(ctx, logits, me, token):
        ctx.me, ctx.token = weakref.ref(me), token
        return me._loss0.detach()                      # (an alias: the node hangs on it, not on the ring's own tensor)

    @staticmethod
    def backward(ctx, g):
        me = ctx.me()
        if me is None or me.token != ctx.token or me.state != _FWD_DONE:
            raise RuntimeError('gist_amd: backward through a loss whose forward is no longer the model\'s latest '
                               '(the engine reuses its buffers; GIST_MODULE_ENGINE=0 for the op-by-op path)')
        n = me._pending[0].n
        return me.engine.dlogits[:n, :me.n_classes] * g, None, None


class StepLoss(torch.Tensor):
    """The loss tensor of a fused step.  An ordinary scalar tensor whose .backward() with no arguments -- what the
    reference's loop calls -- runs the backward phase directly (one dispatcher op, no tape walk).  It lives in a slot of
    the engine's loss ring for as long as the caller holds it (a held slot is not reused); .detach() copies out."""
    __torch_function__ = torch._C._disabled_torch_function_impl

    def detach(self):
        return torch.Tensor.detach(self).clone()

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        st = self.__dict__.get('_gist_step')
        if (st is not None and gradient is None and not create_graph and inputs is None and not retain_graph
                and st[0].fast_backward(st[1])):
            return None
        return torch.Tensor.backward(self, gradient, retain_graph, create_graph, inputs)


def _is_relu(fn):
    import torch.nn as nn
    import torch.nn.functional as F
    return fn is F.relu or fn is torch.relu or isinstance(fn, nn.ReLU)


def eligible(model):
    """Is `model` (gist_amd.modules.GCN) the network the step plan implements?  ISTSAGELayer everywhere, ReLU + the
    model's LayerNorm flag on all but the last layer, none on the last, one dropout probability."""
    from .modules import ISTSAGELayer
    layers = list(model.layers)
    if not layers or len(layers) > _lib.GIST_MAX_LAYERS or not all(type(l) is ISTSAGELayer for l in layers):
        return False
    p = layers[0].p_drop
    ln = layers[0].use_lynorm if len(layers) > 1 else False
    for k, l in enumerate(layers):
        last = k == len(layers) - 1
        if l.p_drop != p or l.linear.bias is None:
            return False
        if last:
            if l.use_lynorm or l.activation is not None:
                return False
        elif l.use_lynorm != ln or l.activation is None or not _is_relu(l.activation):
            return False
        if k > 0 and l.linear.in_features != 2 * layers[k - 1].linear.out_features:
            return False
    return True


class ModuleEngine(object):
    """One nn.Module GCN bound to one ClusterIter: the SageEngine behind `model(cluster)`."""

    def __init__(self, model, it):
        layers = list(model.layers)
        self._model, self.it = weakref.ref(model), it
        dims = [(l.linear.in_features // 2, l.linear.out_features) for l in layers]
        dev = it.g.device
        ln = layers[0].use_lynorm if len(layers) > 1 else False
        self.engine = eng = SageEngine(dims, ln, layers[0].p_drop, it.n_max, dev, seed=getattr(model, '_drop_seed', 0))
        eng.arena.adopt_module(model)
        eng.prefetch = os.environ.get('GIST_MODULE_PREFETCH', '1') != '0'
        it.bind(eng)
        if eng.plan is None:
            raise RuntimeError('gist_amd: no native step plan for this model')
        A = eng.arena
        self.params = [p for l in layers for p in (l.linear.weight, l.linear.bias)]
        self.grad_views = [v for k in range(len(layers)) for v in (A.dW[k], A.db[k])]
        self._home_ptrs = [v.data_ptr() for k in range(len(layers)) for v in (A.W[k], A.b[k])]
        # (a Parameter knows its engine by HANDLE: a strong reference here would close a Parameter <-> ModuleEngine cycle
        # that only the cyclic collector could free -- and never does once the objects are in its permanent generation)
        self.n_classes, self.ldc = eng.n_classes, eng.ldc
        self.handle = _NEXT_HANDLE[0]
        _NEXT_HANDLE[0] += 1
        _REGISTRY[self.handle] = self
        for p in self.params:
            p._gist_me = self.handle
        self.token = 0
        self.state = _IDLE
        self._pending = None        # (engine Batch, cluster, logits tensor)
        self._loss0 = None
        self._f32 = dict(dtype=torch.float32, device=dev)
        self._last = len(layers) - 1
        self._first_step = True
        # logits and loss of a step live in small rings (no allocator call in the loop).  A slot whose tensor the caller
        # still holds (the tensor, a view of it, `preds.append(model(c))`) is NOT reused: the ring gets a fresh buffer
        # for that slot, and the held one stays the caller's -- every call's result is its own tensor, as in torch.
        self._logit_ring = [torch.empty(it.n_max, self.ldc, **self._f32) for _ in range(4)]
        self._loss_ring = [torch.zeros((), **self._f32) for _ in range(64)]
        self._idle_uses = _storage_uses(self._logit_ring[0])

    def __deepcopy__(self, memo):
        return None                   # (a copied model binds its own engine on first use)

    def homed(self):
        """Is EVERY parameter's storage still its arena view?  (model.to(...), an assigned .data, load_state_dict(assign=True),
        an IST mover that swaps tensors: any of them on any parameter sends forward / Adam.step back through adoption)"""
        for p, ptr in zip(self.params, self._home_ptrs):
            if p.data_ptr() != ptr:
                return False
        return True

    def _ring_slot(self, ring, i, make):
        """ring[i] if nobody outside the engine holds it (or a view of it), else a fresh buffer put in its place."""
        t = ring[i]
        if _storage_uses(t) > self._idle_uses:
            t = ring[i] = make()
        return t

    # ---- forward -------------------------------------------------------------------------------------------------
    def forward(self, g, training):
        eng = self.engine
        if not self.homed():          # model.to(...) / .data replaced: bring the values back into the arena
            eng.arena.adopt_module(self._model())
        n = g._n
        b = self.it.batcher.lazy(g._ids)
        b.row_blocks, b.parts, b.next_info = g.row_blocks, g.parts, g.next_info
        b.siblings = g.siblings
        P = eng.plan
        self.token += 1
        self._pending = None          # (the previous step's view of its ring slot)
        if not training:
            y = torch.empty(n, self.ldc, **self._f32)      # evaluation: every call's logits are their own tensor
        else:
            y = self._ring_slot(self._logit_ring, self.token & 3,
                                lambda: torch.empty(self.it.n_max, self.ldc, **self._f32))[:n]
        P.layer[self._last].Y = y.data_ptr()
        self._loss0 = self._ring_slot(self._loss_ring, self.token & 63, lambda: torch.zeros((), **self._f32))
        P.loss = self._loss0.data_ptr()
        self._pending = (b, g, y)
        if not training:
            eng._native_step(b, 0.0, 0.0, train=False)
            self.state = _IDLE
            return y if self.ldc == self.n_classes else y[:, :self.n_classes]
        if torch.is_grad_enabled():
            out = _GCNForward.apply(self, self.token, n, self.ldc, *self.params)
        else:
            out = torch.ops.gist.gcn_forward(self.params, self.handle, self.token, n, self.ldc, True)
        pred = out if self.ldc == self.n_classes else out[:, :self.n_classes]
        pred._gist_step = (self, self.token)
        return pred

    def _run_forward(self, token, n, ldc, train):
        b, g, y = self._pending
        if token != self.token or n != b.n:
            raise RuntimeError('gist_amd: gist::gcn_forward called with a stale token')
        self.engine._native_step(b, 0.0, 0.0, train=True, phase=_lib.GIST_STEP_PHASE_FORWARD)
        self.state = _FWD_DONE
        return y

    # ---- loss ----------------------------------------------------------------------------------------------------
    def fused_loss(self, logits, labels, token):
        """The mean CE of the forward `token` if (logits, labels) are its logits and the batch's own labels."""
        if token != self.token or self.state != _FWD_DONE:
            return None
        g = self._pending[1]
        if labels is not dict.get(g.ndata, 'label') or logits.shape[0] != g._n:
            return None
        if torch.is_grad_enabled() and logits.requires_grad:
            loss = _FusedLoss.apply(logits, self, token)
        else:
            loss = self._loss0
        loss = loss.as_subclass(StepLoss)
        loss._gist_step = (self, token)
        return loss

    # ---- backward ------------------------------------------------------------------------------------------------
    def fast_backward(self, token):
        """loss.backward() of the standard loop: True if the backward phase ran (gradients in p.grad)."""
        if token != self.token or self.state != _FWD_DONE:
            return False
        for p in self.params:
            if p.grad is not None or not p.requires_grad:
                return False
        views = torch.ops.gist.gcn_backward(self.engine.dlogits, self.handle, token, False)
        for p, v in zip(self.params, views):
            p.grad = v
        return True

    def _run_backward(self, token, d_logits, given):
        if token != self.token or self.state != _FWD_DONE:
            raise RuntimeError('gist_amd: backward through a GCN forward that is no longer the model\'s latest (the '
                               'engine reuses its buffers; GIST_MODULE_ENGINE=0 for the op-by-op path)')
        eng = self.engine
        b = self._pending[0]
        if given:
            hip.block_gather(d_logits if d_logits.stride(-1) == 1 else d_logits.contiguous(), None, None,
                             eng.dlogits[:b.n, :d_logits.shape[1]])
        eng._native_step(b, 0.0, 0.0, train=True, phase=_lib.GIST_STEP_PHASE_BACKWARD, given=bool(given))
        self.state = _BWD_DONE
        return self.grad_views

    def autograd_backward(self, token, d_y):
        """The tape's way in (any loss on the logits): d_y is the gradient w.r.t. the padded logits.  Gradients are
        delivered like torch's AccumulateGrad would: p.grad = the arena view, or added to what is there."""
        olds = [(v, v.clone()) for p, v in zip(self.params, self.grad_views) if p.grad is v]
        views = torch.ops.gist.gcn_backward(d_y, self.handle, token, True)
        for v, o in olds:
            v.add_(o)
        for p, v in zip(self.params, views):
            if not p.requires_grad:
                continue
            if p.grad is None:
                p.grad = v
            elif p.grad is not v:
                p.grad.add_(v)

    # ---- optimiser -----------------------------------------------------------------------------------------------
    def owns(self, params):
        return len(params) == len(self.params) and all(a is b for a, b in zip(params, self.params))

    def grads_in_arena(self):
        return all(p.grad is v for p, v in zip(self.params, self.grad_views))

    def flat_state(self, opt):
        """Adam's moments for this model as two flat arrays in the arena's layout (per-tensor views in opt.state)."""
        fs = getattr(opt, '_flat_state', None)
        if fs is None or fs[0] is not self:
            A = self.engine.arena
            m = torch.zeros(A.numel, **self._f32)
            v = torch.zeros(A.numel, **self._f32)
            for i, (p, gv) in enumerate(zip(self.params, self.grad_views)):
                off = (gv.data_ptr() - A.grads.data_ptr()) // 4
                mv, vv = m[off:off + p.numel()].view_as(p), v[off:off + p.numel()].view_as(p)
                if opt.state[i] is not None:          # moments of earlier per-tensor steps
                    mv.copy_(opt.state[i][0])
                    vv.copy_(opt.state[i][1])
                opt.state[i] = (mv, vv)
            fs = opt._flat_state = (self, m, v)
        return fs[1], fs[2]

    def optimizer_step(self, opt):
        """optimizer.step() over the flat arena: the optimiser phase of the pending step (+ the next batch's
        extraction), or -- gradients that did not come from this engine's backward -- one plain Adam launch."""
        eng = self.engine
        A = eng.arena
        m, v = self.flat_state(opt)
        lr = opt.param_groups[0]['lr']
        if self.state == _BWD_DONE:
            P = eng.plan
            P.exp_avg, P.exp_avg_sq = m.data_ptr(), v.data_ptr()
            eng._native_step(self._pending[0], lr, opt.weight_decay, train=True, betas=opt.betas, eps=opt.eps,
                             phase=_lib.GIST_STEP_PHASE_OPTIMIZER, adam_step=opt.step_count)
            self.state = _IDLE
            if self._first_step:
                # the first complete iteration has pulled in what torch imports lazily (dispatcher caches, autograd,
                # pinned-memory allocator: ~10^5 objects created AFTER bind()'s freeze): freeze those too, or the first
                # full garbage collection of the loop (~100 iterations in) stalls the host for 20-40 ms
                self._first_step = False
                from .sampler import freeze_setup_objects
                freeze_setup_objects()
        else:
            hip.adam_(A.params, A.grads, m, v, opt.step_count, lr, opt.betas[0], opt.betas[1], opt.eps,
                      opt.weight_decay)


def engine_for(model, g):
    """The ModuleEngine behind model(g), or None: g is not a described cluster batch, the model is not the step plan's
    network, or GIST_MODULE_ENGINE=0."""
    from .graph import ClusterBatch
    if type(g) is not ClusterBatch:
        return None
    mes = model.__dict__.get('_module_engines')
    if mes is None:
        mes = model.__dict__['_module_engines'] = {}
    it = g._it
    me = mes.get(id(it))
    if me is None:
        ok = (os.environ.get('GIST_MODULE_ENGINE', '1') != '0' and hip._prof is None and eligible(model) and it.feed()
              and all(p.is_cuda and p.device == it.g.device and p.dtype == torch.float32 for p in model.parameters())
              and model.layers[0].linear.in_features == 2 * it.batcher.feat.shape[1])
        me = mes[id(it)] = ModuleEngine(model, it) if ok else False
        # (me.it keeps id(it) unique for the model's lifetime)
    return me or None
