"""SageEngine: the training step of GIST's hot path laid out for MI355X.

One engine = one (sub-)GCN on one GPU.  Everything lives in HBM for the whole run:

  * parameters, gradients and Adam moments are four flat fp32 arenas; layer k's
    W_k [out, 2*in] and b_k [out] are views.  Adam is ONE kernel over the arena and
    the IST sync all-gathers the arena as it stands (no packing copies).
  * activations are preallocated for the largest batch: Z_k = [h | ah]  [n, 2*in_k]
    (left half written by the previous layer's LN+ReLU epilogue or the feature
    gather, right half by the SpMM: torch.cat of modules.py:227 never materialises),
    Y_k [n, out_k] (pre-norm, overwritten by yhat, then by dY in the backward).
  * the cluster batch (induced CSR + reversed CSR, norm, labels, features) is
    extracted on the device from the resident training graph; the host only sends
    the epoch's part order once per epoch.  No per-iteration H2D, no host sync.

The step follows SURVEY.md appendix A / cluster_gcn_ist_distrib.py:408-417 exactly:
forward, mean CE over the batch rows, backward, Adam (coupled L2).
"""
import os

import numpy as np
import torch

from . import hip


_FORCE_PAIRS = os.environ.get('GIST_SPMM_PAIRS') == '1'      # dev: every batch is prepared with pairs / fine blocks


def _round_up(x, m):
    return (x + m - 1) // m * m


class ParamArena(object):
    """Flat parameter / gradient / Adam-moment storage with per-layer views."""

    def __init__(self, dims, device, with_grads=True):
        self.dims = list(dims)
        self.offsets = []
        off = 0
        for (i, o) in self.dims:
            self.offsets.append((off, off + o * 2 * i))
            off += o * 2 * i + o
        self.numel = off
        self.device = device
        self.params = torch.zeros(off, dtype=torch.float32, device=device)
        self.W, self.b, self.dW, self.db = [], [], [], []
        for (i, o), (w0, b0) in zip(self.dims, self.offsets):
            self.W.append(self.params[w0:b0].view(o, 2 * i))
            self.b.append(self.params[b0:b0 + o])
        self.grads = self.exp_avg = self.exp_avg_sq = None
        if with_grads:          # a base-model replica (IST) holds parameters only
            self.grads = torch.zeros(off, dtype=torch.float32, device=device)
            self.exp_avg = torch.zeros(off, dtype=torch.float32, device=device)
            self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=device)
            for (i, o), (w0, b0) in zip(self.dims, self.offsets):
                self.dW.append(self.grads[w0:b0].view(o, 2 * i))
                self.db.append(self.grads[b0:b0 + o])
        self.step = 0

    def reset_optimizer(self):
        """Fresh Adam state (cluster_gcn_ist_distrib.py:405-407 builds a new optimizer)."""
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.step = 0

    def load(self, params):
        """params = [(W, b)] numpy arrays or tensors."""
        for k, (W, b) in enumerate(params):
            self.W[k].copy_(torch.as_tensor(W).to(self.device))
            self.b[k].copy_(torch.as_tensor(b).to(self.device))

    def export(self):
        return [(W.detach().cpu().numpy().copy(), b.detach().cpu().numpy().copy())
                for W, b in zip(self.W, self.b)]

    def adopt_module(self, gcn):
        """Re-home an nn.Module GCN's parameters into the arena (values preserved)."""
        for k, layer in enumerate(gcn.layers):
            self.W[k].copy_(layer.linear.weight.data.to(self.device))
            self.b[k].copy_(layer.linear.bias.data.to(self.device))
            layer.linear.weight.data = self.W[k]
            layer.linear.bias.data = self.b[k]


class Batch(object):
    """Views into the batcher's buffers describing the current cluster batch."""
    __slots__ = ('n', 'rowptr', 'col', 't_rowptr', 't_col', 'norm', 'labels', 'ids', 'ready',
                 'row_blocks', 'batcher', 'parts', 'z0_dropped', 'next_info', 'ah_owner', 'siblings')

    def __init__(self):
        self.ready = True
        # CONTRACT: one training step per extraction.  When layer 0's dropout was folded into the extraction's
        # feature gather, Z[0]'s left half holds dropout(features) afterwards, not the features: a second
        # training step on the same Batch object would aggregate and drop the dropped values again.
        # z0_dropped records that state and train_step refuses such a batch (re-extract it instead).
        self.z0_dropped = False
        self.row_blocks = None      # int32 [n_blocks + 1]: row ranges of the batch's METIS parts
        self.batcher = None         # set on lazy batches: who extracts them
        self.parts = None           # (node_part [N, 2], part_slot [parts, 2], batch index) -- sampler
        self.next_info = None       # (ids, batch index) of the batch that follows in the same epoch -- sampler
        self.siblings = True        # may two parts of the batch share hundreds of edges? (gist_step_plan.sibling_parts)
        # the engine whose Z[0] right half already holds layer 0's aggregation of this batch (an eager extraction through
        # gist_extract_parts_desc_batch with feat_intra); cleared by the first forward that drops Z[0] in place
        self.ah_owner = None


class ClusterBatcher(object):
    """Device-resident training graph + on-device induced-subgraph extraction.

    Replaces `get_subgraph` + `cluster.to(device)` (cluster_gcn/partition_utils.py:20-25,
    cluster_gcn_ist_distrib.py:409).  `graph` is a gist_amd.graph.Graph already on the
    GPU; feat [N, F] fp32 and labels [N] int32 live beside it.
    """

    def __init__(self, graph, feat, labels, n_max, nnz_max):
        dev = graph.device
        assert dev.type == 'cuda', 'ClusterBatcher needs the training graph on the GPU'
        self.g, self.feat, self.labels = graph, feat, labels
        self.n_max, self.nnz_max = int(n_max), int(max(nnz_max, 1))
        i32 = dict(dtype=torch.int32, device=dev)
        self.remap = torch.empty(graph.number_of_nodes(), **i32)
        hip.fill_i32_(self.remap, -1)
        self.rowptr = torch.zeros(self.n_max + 1, **i32)
        self.t_rowptr = torch.zeros(self.n_max + 1, **i32)
        self.col = torch.zeros(self.nnz_max, **i32)
        self.t_col = torch.zeros(self.nnz_max, **i32)
        self.norm = torch.zeros(self.n_max, dtype=torch.float32, device=dev)
        self.lab = torch.zeros(self.n_max, **i32)
        # per node the sum of its in-neighbours' features INSIDE its part (set by the iterator when the batches are
        # unions of locality parts): the one-launch extraction then forms layer 0's aggregation itself
        self.feat_intra = None
        # what a step's optimiser launch extracted into these buffers for the NEXT step (SageEngine.prefetch):
        # (part_slot table, batch index, rows, ids pointer, dropout offset) or None
        self.prefetched = None

    def lazy(self, ids):
        """Describe the batch WITHOUT launching anything: the native step driver
        (gist_sage_step) performs the extraction itself as the first part of the step."""
        n = ids.numel()
        if n > self.n_max:
            raise ValueError('gist_amd: batch of %d rows exceeds n_max=%d' % (n, self.n_max))
        b = Batch()
        b.n, b.rowptr, b.col = n, self.rowptr[:n + 1], self.col
        b.t_rowptr, b.t_col = self.t_rowptr[:n + 1], self.t_col
        b.norm, b.labels, b.ids = self.norm[:n], self.lab[:n], ids
        b.ready = False
        b.batcher = self
        return b

    def extract(self, ids, z0_left, drop=None):
        """ids: int32 device tensor (node ids in the training graph); z0_left: the [n, F]
        left half of layer 0's [h | ah] buffer, filled with the gathered features.
        drop = (x0, p, seed, offset, mask_ld): layer 0's dropout folded into the gather
        (gist_extract_batch_drop): z0_left receives dropout(features), x0 the features."""
        n = ids.numel()
        if n > self.n_max:
            raise ValueError('gist_amd: batch of %d rows exceeds n_max=%d' % (n, self.n_max))
        self.prefetched = None      # (the buffers are overwritten)
        g = self.g
        rp, trp = self.rowptr[:n + 1], self.t_rowptr[:n + 1]
        if drop is not None:
            hip.extract_batch_drop(g, ids, self.remap, rp, self.col, trp, self.t_col, self.norm, self.feat,
                                   z0_left, self.labels, self.lab, *drop)
        else:
            hip.extract_batch(g, ids, self.remap, rp, self.col, trp, self.t_col, self.norm, self.feat,
                              z0_left, self.labels, self.lab)
        b = Batch()
        b.n, b.rowptr, b.col, b.t_rowptr, b.t_col = n, rp, self.col, trp, self.t_col
        b.norm, b.labels, b.ids = self.norm[:n], self.lab[:n], ids
        return b


class SageEngine(object):
    def __init__(self, dims, use_layernorm, dropout, n_max, device, seed=0, arena=None):
        """dims = [(in_k, out_k)] for the L+1 SAGE layers (modules.py:245-308)."""
        self.dims = [(int(i), int(o)) for i, o in dims]
        self.L1 = len(self.dims)
        self.use_layernorm = bool(use_layernorm)
        self.p_drop = float(dropout) if dropout else 0.0
        self.n_max = int(n_max)
        self.device = device
        self.seed = int(seed)
        self.drop_calls = 0
        self.arena = arena if arena is not None else ParamArena(self.dims, device)
        f32 = dict(dtype=torch.float32, device=device)
        self.Z = [torch.zeros(self.n_max, 2 * i, **f32) for (i, o) in self.dims]
        self.n_classes = self.dims[-1][1]
        self.ldc = _round_up(self.n_classes, 4)
        self.Y = [torch.zeros(self.n_max, o, **f32) for (i, o) in self.dims[:-1]]
        self.Y.append(torch.zeros(self.n_max, self.ldc, **f32))      # logits, padded ld
        self.dlogits = torch.zeros(self.n_max, self.ldc, **f32)
        self.rstd = [torch.zeros(self.n_max, **f32) for _ in self.dims[:-1]]
        wide = max([2 * i for (i, o) in self.dims[1:]] + [4])
        self.dZ = torch.zeros(self.n_max * wide, **f32)
        max_out = max(o for (i, o) in self.dims)
        self.partials = torch.zeros(
            max(1, hip._lib.load().gist_colsum_partials(self.n_max)) * max_out, **f32)
        self.row_loss = torch.zeros(self.n_max, **f32)
        self.loss = torch.zeros(1, **f32)
        self.correct = torch.zeros(1, dtype=torch.int32, device=device)
        # split-K workspace sized once for the largest request of any GEMM of the step
        need = 0
        L = hip._lib.load()
        for (i, o) in self.dims:
            for (m, n, k) in ((self.n_max, o, 2 * i), (self.n_max, 2 * i, o), (o, 2 * i, self.n_max)):
                need = max(need, L.gist_gemm_workspace_bytes(m, n, k))
        self._ws = hip.workspace(need, device)
        self._ws2 = torch.empty(max(int(need), 1 << 20), dtype=torch.uint8, device=device)
        self._drop_offsets = []
        # The fused sequence (include/gist_hip.h, gist_step_plan.fuse; GIST_STEP_FUSE=0 = the un-fused
        # one): H[k] = undropped input of layer k where its dropout is folded into the producers,
        # bias-gradient chunk sums and the slabs of deferred split-K projections.
        self.fuse = os.environ.get('GIST_STEP_FUSE', '1') != '0'
        self.H = [None] * self.L1
        if self.fuse and self.p_drop > 0.0:
            for k, (i, o) in enumerate(self.dims):
                ld = i if i % 4 == 0 else _round_up(i + 2, 4)
                self.H[k] = torch.zeros(self.n_max, ld, **f32)
        self._fused = None          # op-by-op path's own slabs / chunk sums (lazy)
        self._extract_scratch = None    # barrier ticket + counts of the one-launch extraction
        # True: a training step's optimiser launch also extracts the NEXT batch of the epoch into the batch buffers
        # (gist_adam_segments_extract_f32).  For loops that only use the loss: labels, CSR and Z[0] of the batch just
        # stepped are gone when train_step returns.  The trainers and bench.py set it; off by default
        self.prefetch = False
        self._prefetch_refused = None
        self._segments = []
        self._logit_slabs_n = 1
        self.plan = None
        self._spmm_prep = None      # prepared block structure of the current batch (native step)
        self._plan_keep = None
        # inference-only reassociation of the last layer (see forward); GIST_PROJECT_FIRST=0
        # keeps the reference's aggregate-then-project order
        self.project_first = os.environ.get('GIST_PROJECT_FIRST', '1') != '0'

    # ------------------------------------------------------------------
    def attach_batcher(self, batcher):
        """Build the native step plan (struct gist_step_plan): after this, train_step is
        ONE call into libgist_hip.so per iteration (gist_sage_step) instead of ~45."""
        from . import _lib
        A = self.arena
        if A.grads is None:
            raise ValueError('gist_amd: the native step needs an arena with gradients')
        if self.L1 > _lib.GIST_MAX_LAYERS:
            return None
        P = _lib.StepPlan()
        P.n_layers = self.L1
        P.use_layernorm = int(self.use_layernorm)
        P.p_drop = self.p_drop
        P.seed = self.seed
        for k, (i, o) in enumerate(self.dims):
            l = P.layer[k]
            l.n_in, l.n_out = i, o
            l.W, l.b = A.W[k].data_ptr(), A.b[k].data_ptr()
            l.dW, l.db = A.dW[k].data_ptr(), A.db[k].data_ptr()
            l.Z, l.ldz = self.Z[k].data_ptr(), 2 * i
            l.Y, l.ldy = self.Y[k].data_ptr(), self.Y[k].shape[1]
            l.rstd = self.rstd[k].data_ptr() if k < self.L1 - 1 else None
        P.dlogits, P.ldc = self.dlogits.data_ptr(), self.ldc
        P.dZ, P.partials = self.dZ.data_ptr(), self.partials.data_ptr()
        P.row_loss, P.loss = self.row_loss.data_ptr(), self.loss.data_ptr()
        P.workspace, P.workspace_bytes = self._ws.data_ptr(), self._ws.numel()
        P.workspace2, P.workspace2_bytes = self._ws2.data_ptr(), self._ws2.numel()
        P.params, P.grads = A.params.data_ptr(), A.grads.data_ptr()
        P.exp_avg, P.exp_avg_sq = A.exp_avg.data_ptr(), A.exp_avg_sq.data_ptr()
        P.n_params = A.numel
        g = batcher.g
        P.g_rowptr, P.g_col = g.rowptr.data_ptr(), g.col.data_ptr()
        P.g_t_rowptr, P.g_t_col = g.t_rowptr.data_ptr(), g.t_col.data_ptr()
        P.feat, P.ld_feat = batcher.feat.data_ptr(), batcher.feat.stride(0)
        fi = getattr(batcher, 'feat_intra', None)
        if fi is not None and self.fuse:
            P.feat_intra, P.ld_feat_intra = fi.data_ptr(), fi.stride(0)
        else:
            P.feat_intra, P.ld_feat_intra = None, 0
        P.labels_all, P.remap = batcher.labels.data_ptr(), batcher.remap.data_ptr()
        P.rowptr, P.col = batcher.rowptr.data_ptr(), batcher.col.data_ptr()
        P.t_rowptr, P.t_col = batcher.t_rowptr.data_ptr(), batcher.t_col.data_ptr()
        P.col_capacity = batcher.col.numel()
        P.norm, P.labels = batcher.norm.data_ptr(), batcher.lab.data_ptr()
        # split projection operands kept by the step (include/gist_hip.h, h3_workspace): sized by
        # the library for these shapes; 0 bytes = mode 'f32' or no layer large enough
        P.n_max = self.n_max
        # upper bound of |feat| for the f16x3 mode's layer-0 split scale (ignored in the other
        # modes).  `batcher.feat` must not grow in place after this: call attach_batcher again
        # (it re-reads the bound) if the features are re-normalised or replaced.
        P.feat_absmax = float(batcher.feat.abs().max().item()) if batcher.feat.numel() else 0.0
        import ctypes
        # (sized for the larger of the two split modes, so the GEMM mode may be switched after bind)
        L_ = _lib.load()
        cur = L_.gist_gemm_get_mode()
        need = 0
        for m_ in ((1, 2) if cur != 0 else ()):      # (a query per mode: the process-wide mode is not touched)
            need = max(need, L_.gist_step_h3_workspace_bytes_mode(ctypes.byref(P), m_))
        self._h3_ws = None
        if need > 0 and os.environ.get('GIST_STEP_H3', '1') != '0':
            self._h3_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            P.h3_workspace, P.h3_workspace_bytes = self._h3_ws.data_ptr(), need
        if self._spmm_prep is not None:
            P.spmm_prepared, P.spmm_prepared_bytes = self._spmm_prep.data_ptr(), self._spmm_prep.numel()
        P.fuse = int(self.fuse)
        self._fused_ws = self._col_partials = None
        if self.fuse:
            for k in range(self.L1):
                if self.H[k] is not None:
                    P.hsrc[k], P.ld_hsrc[k] = self.H[k].data_ptr(), self.H[k].stride(0)
            nf = L_.gist_step_col_partials_floats(ctypes.byref(P))
            self._col_partials = torch.zeros(max(int(nf), 4), dtype=torch.float32, device=self.device)
            P.col_partials = self._col_partials.data_ptr()
            nb = L_.gist_step_fused_workspace_bytes(ctypes.byref(P))
            if nb > 0:
                self._fused_ws = torch.empty(int(nb), dtype=torch.uint8, device=self.device)
                P.fused_workspace, P.fused_workspace_bytes = self._fused_ws.data_ptr(), int(nb)
        self.plan = P
        self._plan_keep = (batcher, g, self._ws, self._ws2, self._h3_ws, self._fused_ws,
                           self._col_partials)     # keep every buffer alive
        return P

    def check_extract(self):
        """Raises if a workgroup of the one-launch extraction ever gave up at its grid barrier (the
        error word of gist_extract_parts_batch's scratch); one small D2H read, call it off the hot path."""
        if self._extract_scratch is not None and int(self._extract_scratch[1].item()) != 0:
            raise RuntimeError('gist_amd: gist_extract_parts_batch timed out at its grid barrier; '
                               'the batches extracted since the last check are invalid')

    def check_extract_deferred(self):
        """The same check without draining the queue: a one-thread kernel writes the error word and a running tag into
        pinned host memory (gist_publish_i64: no copy command, no event object, no busy-waiting runtime call); this call
        first makes sure the mark of the PREVIOUS call has arrived -- it normally has, long ago; a host more than one call
        ahead of the GPU sleeps here in 50-us naps, which is what bounds its run-ahead -- and raises for that mark's
        error word."""
        if self._extract_scratch is None:
            return
        import time
        if getattr(self, '_mark', None) is None:
            self._mark = torch.zeros(2, dtype=torch.int64, pin_memory=True)
            self._mark_np = self._mark.numpy()
            self._mark_tag = 0
        prev = self._mark_tag
        if prev > 0:
            deadline = None
            while int(self._mark_np[1]) < prev:
                if deadline is None:
                    deadline = time.time() + 600.0
                elif time.time() > deadline:
                    raise RuntimeError('gist_amd: the GPU never reached the progress mark of the previous epoch')
                time.sleep(5e-5)
            if int(self._mark_np[0]) != 0:
                raise RuntimeError('gist_amd: gist_extract_parts_batch timed out at its grid barrier; '
                                   'the batches extracted since the last check are invalid')
        self._mark_tag = prev + 1
        hip.publish_i64_raw(self._extract_scratch[1:2].data_ptr(), self._mark_tag, self._mark.data_ptr())

    def enable_timer(self, capacity):
        """HIP-event timing of every SpMM/GEMM issued by the native step (gist_timer_*)."""
        from . import _lib
        L = _lib.load()
        if self.plan is None:
            raise RuntimeError('gist_amd: enable_timer needs attach_batcher first')
        self.disable_timer()
        self._timer = L.gist_timer_create(int(capacity))
        self.plan.timer = self._timer
        return self._timer

    def disable_timer(self):
        from . import _lib
        if getattr(self, '_timer', None):
            _lib.load().gist_timer_destroy(self._timer)
        self._timer = None
        if self.plan is not None:
            self.plan.timer = None

    def read_timer(self):
        """[(ms, kind, m, n, k)] -- synchronises the device first."""
        import ctypes
        from . import _lib
        L = _lib.load()
        torch.cuda.synchronize(self.device)
        out = []
        ms, kind = ctypes.c_float(), ctypes.c_int32()
        m, n, k = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        for i in range(L.gist_timer_count(self._timer)):
            _lib.check(L.gist_timer_read(self._timer, i, ctypes.byref(ms), ctypes.byref(kind),
                                         ctypes.byref(m), ctypes.byref(n), ctypes.byref(k)),
                       'gist_timer_read')
            out.append((ms.value, kind.value, m.value, n.value, k.value))
        return out

    def _native_step(self, b, lr, weight_decay, train, betas=(0.9, 0.999), eps=1e-8, phase=0, given=False,
                     adam_step=None):
        """One gist_sage_step call.  phase = 0: the whole iteration; GIST_STEP_PHASE_FORWARD / _BACKWARD / _OPTIMIZER: one
        third of it (the module path: GCN.forward, loss.backward(), optimizer.step()); the backward and optimiser calls
        reuse the forward call's batch, dropout offset and flags.  given: the caller wrote its own dlogits."""
        import ctypes
        from . import _lib
        L = _lib.load()
        P = self.plan
        if phase in (_lib.GIST_STEP_PHASE_BACKWARD, _lib.GIST_STEP_PHASE_OPTIMIZER):
            cb, off, flags, ids_ptr = self._phase_ctx
            if cb is not b:
                raise RuntimeError('gist_amd: backward / optimiser phase of a batch that is not the last one forwarded')
            batcher = b.batcher
            nxt = None
            if phase == _lib.GIST_STEP_PHASE_OPTIMIZER:
                nxt, flags = self._plan_next(b, batcher, flags)
                self.arena.step += 1
            elif given:
                flags |= _lib.GIST_STEP_DLOGITS_GIVEN
            rc = L.gist_sage_step(ctypes.byref(P), ids_ptr, b.n, off, lr, betas[0], betas[1], eps, weight_decay,
                                  max(adam_step if adam_step is not None else self.arena.step, 1), flags | phase,
                                  hip._stream())
            _lib.check(rc, 'gist_sage_step')
            if phase == _lib.GIST_STEP_PHASE_OPTIMIZER and batcher is not None:
                batcher.prefetched = nxt
            return self.loss
        off = self.drop_calls
        ids_ptr = b.ids.data_ptr() if b.ids is not None else None
        # was this batch extracted beside the previous step's optimiser launch (self.prefetch)?  The GEMM mode is part of
        # the key: it decides whether layer 0's mask was folded into that extraction's feature gather (advisor, round 4)
        batcher = b.batcher
        pre = batcher.prefetched if batcher is not None else None
        if batcher is not None:
            batcher.prefetched = None
        pre_ok = (pre is not None and train and not b.ready and b.parts is not None and
                  pre == (b.parts[1].data_ptr(), int(b.parts[2]), b.n, ids_ptr, off, L.gist_gemm_get_mode(), id(self)))
        flags = (_lib.GIST_STEP_TRAIN if train else 0)
        if pre_ok:
            flags |= _lib.GIST_STEP_PREEXTRACTED
        elif not b.ready:
            flags |= _lib.GIST_STEP_EXTRACT
        if train and self.p_drop > 0.0:
            for (i, o) in self.dims:
                numel = b.n * 2 * i
                self.drop_calls += numel + (numel & 1)
        if train and phase == 0:
            self.arena.step += 1
        rb = b.row_blocks
        P.sibling_parts = 1 if (b.siblings or _FORCE_PAIRS) else 0
        if rb is not None and rb.numel() > 1:
            P.row_blocks, P.n_row_blocks = rb.data_ptr(), rb.numel() - 1
            # room for the batch's prepared block structure (include/gist_hip.h, spmm_prepared):
            # both orientations, grown to the largest block count seen
            need = 2 * L.gist_spmm_blocks_bytes(rb.numel() - 1)
            if self._spmm_prep is None or self._spmm_prep.numel() < need:
                self._spmm_prep = torch.empty(need + need // 4, dtype=torch.uint8, device=self.device)
                P.spmm_prepared = self._spmm_prep.data_ptr()
                P.spmm_prepared_bytes = self._spmm_prep.numel()
        else:
            P.row_blocks, P.n_row_blocks = None, 0
        # one-launch extraction when the batch comes with its part tables (gist_extract_parts_batch)
        if b.parts is not None and not b.ready and self.fuse and L.gist_extract_parts_supported(self.n_max):
            node_part, tab, j = b.parts
            if self._extract_scratch is None:
                self._extract_scratch = torch.zeros(int(L.gist_extract_parts_scratch_bytes(self.n_max)) // 8 + 1,
                                                    dtype=torch.int64, device=self.device)
            P.node_part, P.part_slot = node_part.data_ptr(), tab.data_ptr()
            P.batch_index, P.extract_scratch = int(j), self._extract_scratch.data_ptr()
        else:
            P.node_part = P.part_slot = P.extract_scratch = None
            P.batch_index = -1
        P.next_ids, P.next_n, P.next_batch_index, P.next_drop_offset = None, 0, -1, 0
        nxt = None
        if phase == 0:
            nxt, flags = self._plan_next(b, batcher, flags)
        else:
            self._phase_ctx = (b, off, flags, ids_ptr)
        rc = L.gist_sage_step(ctypes.byref(P), ids_ptr, b.n, off, lr, betas[0], betas[1],
                              eps, weight_decay, max(adam_step if adam_step is not None else self.arena.step, 1),
                              flags | phase, hip._stream())
        _lib.check(rc, 'gist_sage_step')
        if batcher is not None:
            batcher.prefetched = nxt
        if not b.ready and train and self.fuse and self.p_drop > 0.0 and self.H[0] is not None:
            b.z0_dropped = True      # (one training step per extraction: Batch contract)
        b.ready = True
        return self.loss

    def _plan_next(self, b, batcher, flags):
        """The NEXT batch of the epoch, extracted beside this step's optimiser launch (GIST_STEP_EXTRACT_NEXT): only for
        callers that promise not to look at the batch buffers (labels, CSR, Z[0]) after a training step.  Returns
        (the key the next step must match, flags)."""
        import ctypes
        from . import _lib
        L = _lib.load()
        P = self.plan
        nxt = None
        P.next_ids, P.next_n, P.next_batch_index, P.next_drop_offset = None, 0, -1, 0
        if (self.prefetch and self._prefetch_refused is not True and (flags & _lib.GIST_STEP_TRAIN) and batcher is not None
                and b.next_info is not None and P.node_part is not None and b.row_blocks is not None):
            nids, nj = b.next_info
            P.next_ids, P.next_n, P.next_batch_index = nids.data_ptr(), nids.numel(), int(nj)
            P.next_drop_offset = self.drop_calls          # (= `off` of the next training step)
            if L.gist_sage_step_extracts_next(ctypes.byref(P), b.n, flags):
                flags |= _lib.GIST_STEP_EXTRACT_NEXT
                nxt = (P.part_slot, int(nj), nids.numel(), nids.data_ptr(), self.drop_calls, L.gist_gemm_get_mode(),
                       id(self))
            else:
                self._prefetch_refused = True      # (a property of the plan: un-fused, or an arena too large to gain)
        return nxt, flags

    # ------------------------------------------------------------------
    def z0_left(self, n):
        return self.Z[0][:n, :self.dims[0][0]]

    def logits(self, n):
        return self.Y[-1][:n, :self.n_classes]

    def _drop_offset(self, numel):
        off = self.drop_calls
        self.drop_calls += numel + (numel & 1)
        return off

    # -- op-by-op path: the SAME sequence gist_sage_step issues, one C-ABI call per kernel ----------
    def _fused_buffers(self):
        """Slabs / chunk sums of the op-by-op path, sized like the native step's (the split counts of
        the deferred projections must agree: gist_step_fused_slab_bytes)."""
        if self._fused is not None:
            return self._fused
        import ctypes
        from . import _lib
        L = _lib.load()
        P = _lib.StepPlan()
        P.n_layers, P.n_max = self.L1, self.n_max
        for k, (i, o) in enumerate(self.dims):
            P.layer[k].n_in, P.layer[k].n_out = i, o
        u8 = dict(dtype=torch.uint8, device=self.device)
        f32 = dict(dtype=torch.float32, device=self.device)
        chunks = int(L.gist_row_chunks16(self.n_max))
        fb = {'dw': [], 'partials': []}
        for k, (i, o) in enumerate(self.dims):
            nb = int(L.gist_step_fused_slab_bytes(ctypes.byref(P), k))
            fb['dw'].append(torch.empty(nb, **u8) if nb > 0 else None)
            fb['partials'].append(torch.zeros(max(chunks * o, 4), **f32))
        nb = int(L.gist_step_fused_slab_bytes(ctypes.byref(P), self.L1))
        fb['logits'] = torch.empty(nb, **u8) if nb > 0 else None
        nb = int(L.gist_step_fused_slab_bytes(ctypes.byref(P), self.L1 + 1))
        fb['y'] = torch.empty(nb, **u8) if nb > 0 else None          # a hidden layer's forward projection
        self._fused = fb
        return fb

    def _offsets(self, n, training):
        """Dropout counter base of every layer for this step (gist_sage_step's layer k uses
        drop_offset + sum_{j<k} round_up(n * 2 * n_in_j, 2))."""
        offs = []
        if training and self.p_drop > 0.0:
            for (i, o) in self.dims:
                offs.append(self._drop_offset(n * 2 * i))
        return offs

    def forward(self, b, training, _step=False):
        """GCN.forward (modules.py:310-314) on the batch whose features already sit in
        Z[0][:, :F] (a lazy batch is extracted first).  Returns the logits view [n, C].
        _step: called by train_step -- the class layer's logits may stay split-K slabs for the loss
        kernel and a lazy batch's feature gather carries layer 0's dropout, like gist_sage_step."""
        n = b.n
        A = self.arena
        drop = training and self.p_drop > 0.0
        blocked = b.row_blocks is not None and b.row_blocks.numel() > 1
        rb = b.row_blocks if blocked else None
        self._drop_offsets = offs = self._offsets(n, training)
        fold = [False] * self.L1
        if self.fuse and drop:
            for k, (i, o) in enumerate(self.dims):
                fold[k] = (self.H[k] is not None and (k > 0 or (_step and not b.ready)) and
                           hip.spmm_drop_takes(1, i, self.H[k][:n, :i], self.Z[k][:n, i:], blocked))
        self._fwd_fold = fold
        self._pre_ah = False
        if not b.ready:
            if b.batcher is None:
                raise RuntimeError('gist_amd: lazy batch without a batcher')
            i0 = self.dims[0][0]
            dr = (self.H[0][:n, :i0], self.p_drop, self.seed, offs[0], 2 * i0) if fold[0] else None
            fi = getattr(b.batcher, 'feat_intra', None)
            if (fi is not None and b.parts is not None and self.fuse and self.plan is not None and self.plan.feat_intra
                    and hip._lib.load().gist_extract_parts_supported(self.n_max)):
                # the native step's extraction: one launch that also forms layer 0's aggregation
                # (gist_extract_parts_desc.feat_intra), which sums in its own order -- so the twin runs the same launch
                bt = b.batcher
                if self._extract_scratch is None:
                    self._extract_scratch = torch.zeros(
                        int(hip._lib.load().gist_extract_parts_scratch_bytes(self.n_max)) // 8 + 1,
                        dtype=torch.int64, device=self.device)
                bt.prefetched = None
                hip.extract_parts(bt.g, b.ids, self.n_max, b.parts[0], b.parts[1], b.parts[2], bt.rowptr[:n + 1], bt.col,
                                  bt.t_rowptr[:n + 1], bt.t_col, bt.norm, bt.feat, self.z0_left(n), bt.labels, bt.lab,
                                  self._extract_scratch, drop=dr, feat_intra=fi, ah=self.Z[0][:n, i0:])
                self._pre_ah = True
            else:
                b.batcher.extract(b.ids, self.z0_left(n), drop=dr)
            b.ready = True
            b.z0_dropped = dr is not None
        self._logit_slabs_n = 1
        # block structure of the batch, once for all its aggregations (gist_sage_step does the same): every
        # blocked aggregation then runs on a kernel that reads it (fp32 block-dense below 1536 columns, bf16x3 above)
        self._prep_fwd = self._prep_bwd = None
        if blocked and any(hip.spmm_prepared_useful(self.Z[k][:n, :i], self.Z[k][:n, i:])
                           for k, (i, o) in enumerate(self.dims)):
            self._prep_fwd = hip.spmm_prepare(b.rowptr, b.col, rb)
            if training:
                self._prep_bwd = hip.spmm_prepare(b.t_rowptr, b.t_col, rb)
        pf = self._prep_fwd
        for k, (i, o) in enumerate(self.dims):
            z = self.Z[k][:n]
            if not training and o < i and k == self.L1 - 1 and self.project_first:
                # Inference, narrowing layer (H -> C): [h | A^h] W^T = h W1^T + A^(h W2^T), so
                # aggregate the C-wide projection instead of the H-wide activations (full-graph
                # evaluation: one D=4096 pass over 115 M edges becomes a D=41 pass).  Same
                # value up to fp32 summation order; training keeps the reference order because
                # dropout acts on [h | A^h] and dW needs A^h.
                W = A.W[k]
                p_buf = self.dlogits[:n, :o]                      # free in inference
                hip.gemm_nt(z[:, :i], W[:, i:], None, p_buf)
                hip.gemm_nt(z[:, :i], W[:, :i], A.b[k], self.Y[k][:n, :o])
                hip.spmm(b.rowptr, b.col, p_buf, self.Y[k][:n, :o], out_scale=b.norm,
                         accumulate=True)
                continue
            if k == 0 and (self._pre_ah or b.ah_owner is self):      # ah (and its mask when folded) came with the extraction
                if drop and not fold[0]:
                    hip.dropout_(z, self.p_drop, self.seed, offs[k])
                    b.ah_owner = None
            elif fold[k]:      # source = the undropped input, store = dropout(ah)
                hip.spmm_drop(b.rowptr, b.col, self.H[k][:n, :i], z[:, i:], 1, self.p_drop, self.seed,
                              offs[k] + i, 0, 2 * i, out_scale=b.norm, row_blocks=rb, prepared=pf)
            else:
                hip.spmm(b.rowptr, b.col, z[:, :i], z[:, i:], out_scale=b.norm, row_blocks=b.row_blocks,
                         prepared=pf if blocked else None)
                if drop:
                    hip.dropout_(z, self.p_drop, self.seed, offs[k])
            last = k == self.L1 - 1
            if last:
                self._cls_fused = False
                if self.fuse and training and _step:      # the loss kernel sums the split-K slabs
                    fb = self._fused_buffers()
                    # gist_sage_step's class layer: projection, CE, dZ and bias chunks in ONE launch (loss_and_backward)
                    L_ = hip._lib.load()
                    if (fb['dw'][k] is not None and self.ldc <= 64 and (not offs or offs[k] % 2 == 0) and
                            int(hip.tuning('class_fused')) != 1 and hip.class_layer_takes(z, A.W[k], o) and
                            fb['dw'][k].numel() >= L_.gist_class_dw_slab_bytes(n, o, 2 * i)):
                        self._cls_fused = True
                        continue
                    if fb['logits'] is not None:
                        self._logit_slabs_n = hip.gemm_slabs('nt', z, A.W[k], A.b[k], self.Y[k][:n, :o],
                                                             fb['logits'])
                        continue
                hip.gemm_nt(z, A.W[k], A.b[k], self.Y[k][:n, :o])
            else:
                y = self.Y[k][:n]
                i_next = self.dims[k + 1][0]
                rstd = self.rstd[k][:n] if self.use_layernorm else None
                if (self.fuse and training and _step and self._fused_buffers()['y'] is not None and
                        not hip.gemm_splits_own_operands(n, o, 2 * i)):
                    # the projection's k slices stay slabs; the LayerNorm sums them as it reads (gist_sage_step)
                    ys = self._fused_buffers()['y']
                    ns = hip.gemm_slabs('nt', z, A.W[k], A.b[k], y, ys)
                    if ns > 1:
                        hip.ln_relu_fwd_slabs(y, ys, ns, A.b[k], self.Z[k + 1][:n, :i_next],
                                              self.H[k + 1][:n, :i_next] if fold[k + 1] else None, rstd,
                                              self.use_layernorm, True, self.p_drop if fold[k + 1] else 0.0,
                                              self.seed, offs[k + 1] if fold[k + 1] else 0,
                                              2 * i_next if fold[k + 1] else o)
                        continue
                else:
                    hip.gemm_nt(z, A.W[k], A.b[k], y)
                if fold[k + 1]:
                    hip.ln_relu_fwd_drop(y, self.Z[k + 1][:n, :i_next], self.H[k + 1][:n, :i_next], rstd,
                                         self.use_layernorm, True, self.p_drop, self.seed, offs[k + 1],
                                         2 * i_next)
                else:
                    hip.ln_relu_fwd(y, self.Z[k + 1][:n, :i_next], rstd, self.use_layernorm, True)
        return self.logits(n)

    def loss_and_backward(self, b, mask=None, count=None, _step=False):
        """CE (mean over masked rows) + full backward into the gradient arena.  With the fused
        sequence inside train_step (_step, self.fuse, no mask) the bias gradients and split weight
        gradients are left in chunks / slabs for adam_step, like gist_sage_step does; called on its
        own, the gradient arena is complete on return."""
        n = b.n
        A = self.arena
        defer = self.fuse and mask is None and _step
        drop = bool(self._drop_offsets)
        blocked = b.row_blocks is not None and b.row_blocks.numel() > 1
        fb = self._fused_buffers() if defer else None
        pb = getattr(self, '_prep_bwd', None)
        self._segments = []
        self._loss_rows = n if defer else 0
        cls_fused = defer and getattr(self, '_cls_fused', False)
        if cls_fused:
            kL = self.L1 - 1
            iL, oL = self.dims[kL]
            dzL = self.dZ[:n * 2 * iL].view(n, 2 * iL) if kL > 0 else None
            hip.class_layer(self.Z[kL][:n], A.W[kL], A.b[kL], b.labels, n, self.Y[kL][:n, :oL], self.dlogits[:n],
                            self.row_loss[:n], dzL, self.p_drop if drop else 0.0, self.seed,
                            self._drop_offsets[kL] if drop else 0, fb['partials'][kL])
        elif defer:
            hip.softmax_xent_slabs(self.logits(n), fb['logits'], self._logit_slabs_n, A.b[-1], b.labels, None, n,
                                   self.row_loss[:n], None, self.dlogits[:n])
        else:
            if self._logit_slabs_n > 1:
                raise RuntimeError('gist_amd: masked loss after a slab forward')
            hip.softmax_xent(self.logits(n), b.labels, mask, n if count is None else count,
                             self.row_loss[:n], self.loss, self.dlogits[:n])
        L = hip._lib.load()
        chunks = int(L.gist_row_chunks16(n))
        goff = lambda t: (t.data_ptr() - A.grads.data_ptr()) // 4
        for k in range(self.L1 - 1, -1, -1):
            i, o = self.dims[k]
            z = self.Z[k][:n]
            db_done = False
            if k == self.L1 - 1:
                dy = self.dlogits[:n, :o]
            else:
                # dO arrives in the left half of dZ (ld = 2*in_{k+1}); dY overwrites yhat
                i_next = self.dims[k + 1][0]
                d_out = self.dZ[:n * 2 * i_next].view(n, 2 * i_next)[:, :i_next]
                dy = self.Y[k][:n]
                rstd = self.rstd[k][:n] if self.use_layernorm else None
                if defer:
                    hip.ln_relu_bwd_colsum(d_out, dy, rstd, dy, self.use_layernorm, True, fb['partials'][k])
                    db_done = True
                else:
                    hip.ln_relu_bwd(d_out, dy, rstd, dy, self.use_layernorm, True)
            bwd_fold = False
            if cls_fused and k == self.L1 - 1:
                db_done = True                   # (dZ and the bias chunks came with the loss)
                ns = hip.class_dw_slabs(dy, z, fb['dw'][k])
                self._segments.append((goff(A.dW[k]), goff(A.dW[k]) + o * 2 * i, fb['dw'][k], o * 2 * i, ns))
                self._segments.append((goff(A.db[k]), goff(A.db[k]) + o, fb['partials'][k], o, chunks))
                if k > 0:
                    dz = self.dZ[:n * 2 * i].view(n, 2 * i)
                    hip.spmm(b.t_rowptr, b.t_col, dz[:, i:], dz[:, :i], src_scale=b.norm,
                             accumulate=True, row_blocks=b.row_blocks, prepared=pb if blocked else None)
                continue
            dual_done = False
            if k > 0:
                dz = self.dZ[:n * 2 * i].view(n, 2 * i)
                bwd_fold = (self.fuse and drop and k < self.L1 - 1 and
                            hip.spmm_drop_takes(2, i, dz[:, i:], dz[:, :i], blocked))
                p_here = self.p_drop if (drop and not bwd_fold) else 0.0
                off = self._drop_offsets[k] if drop else 0
                # gist_sage_step: dZ_k and dW_k of a narrow hidden layer in one launch (gist_gemm_nn_tn_dual_f32)
                if (defer and k < self.L1 - 1 and db_done and p_here == 0.0 and
                        hip.gemm_dual_takes(dy, A.W[k], z, dz)):
                    ns = hip.gemm_nn_tn_dual(dy, A.W[k], dz, z, A.dW[k], fb['dw'][k])
                    if ns > 1:
                        self._segments.append((goff(A.dW[k]), goff(A.dW[k]) + o * 2 * i, fb['dw'][k], o * 2 * i, ns))
                    dual_done = True
                elif defer and k == self.L1 - 1:
                    hip.gemm_nn_dropout_colsum_(dy, A.W[k], dz, p_here, self.seed, off, fb['partials'][k])
                    db_done = True
                else:
                    hip.gemm_nn_dropout_(dy, A.W[k], dz, p_here, self.seed, off)
            if dual_done:
                pass
            elif defer and fb['dw'][k] is not None:
                ns = hip.gemm_slabs('tn', dy, z, None, A.dW[k], fb['dw'][k])
                if ns > 1:
                    self._segments.append((goff(A.dW[k]), goff(A.dW[k]) + o * 2 * i, fb['dw'][k], o * 2 * i, ns))
            else:
                hip.gemm_tn(dy, z, A.dW[k])
            if defer:
                if not db_done:
                    raise RuntimeError('gist_amd: one-layer models take the native step')
                self._segments.append((goff(A.db[k]), goff(A.db[k]) + o, fb['partials'][k], o, chunks))
            else:
                hip.colsum(dy, A.db[k], self.partials)
            if k > 0:
                if bwd_fold:
                    hip.spmm_drop(b.t_rowptr, b.t_col, dz[:, i:], dz[:, :i], 2, self.p_drop, self.seed,
                                  self._drop_offsets[k], self._drop_offsets[k] + i, 2 * i, src_scale=b.norm,
                                  accumulate=True, row_blocks=b.row_blocks, prepared=pb)
                else:
                    hip.spmm(b.t_rowptr, b.t_col, dz[:, i:], dz[:, :i], src_scale=b.norm,
                             accumulate=True, row_blocks=b.row_blocks, prepared=pb if blocked else None)
        return self.loss

    def adam_step(self, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        A = self.arena
        A.step += 1
        if self._segments or getattr(self, '_loss_rows', 0):
            n = self._loss_rows
            hip.adam_segments_(A.params, A.grads, A.exp_avg, A.exp_avg_sq, A.step, lr, self._segments,
                               row_loss=self.row_loss[:n] if n else None, n_loss_rows=n, loss_count=n,
                               loss=self.loss if n else None, beta1=betas[0], beta2=betas[1], eps=eps,
                               weight_decay=weight_decay)
            self._segments, self._loss_rows = [], 0
            return
        hip.adam_(A.params, A.grads, A.exp_avg, A.exp_avg_sq, A.step, lr, betas[0], betas[1], eps,
                  weight_decay)

    def train_step(self, b, lr, weight_decay=0.0, mask=None, count=None):
        """One iteration of the reference loop (cluster_gcn_ist_distrib.py:408-417).
        Returns the device loss tensor; nothing synchronises with the host.  With a native
        plan attached (attach_batcher) and no mask this is a single gist_sage_step call."""
        if b.ready and getattr(b, 'z0_dropped', False):
            raise RuntimeError('gist_amd: one training step per extraction -- this Batch was already stepped once (with '
                               'dropout fused into the step its buffers may hold dropped values); take a fresh batch '
                               'from the iterator')
        if self.plan is not None and mask is None and hip._prof is None:
            return self._native_step(b, lr, weight_decay, train=True)
        self.forward(b, training=True, _step=mask is None)
        loss = self.loss_and_backward(b, mask, count, _step=True)
        self.adam_step(lr, weight_decay)
        return loss

    def count_correct(self, b, mask=None):
        """Adds #correct argmax predictions of the current logits to self.correct."""
        hip.argmax_correct(self.logits(b.n), b.labels, mask, self.correct)
        return self.correct


def dims_for(in_feats, n_hidden, n_classes, n_layers, split_output=False, num_subnet=1):
    """Layer (in, out) sizes of the reference GCN for split_input=False (modules.py:245-308)."""
    hs = n_hidden // num_subnet
    dims = [(in_feats, n_hidden if (n_layers <= 1 and not split_output) else hs)]
    for i in range(n_layers - 1):
        dims.append((hs, n_hidden if (i == n_layers - 2 and not split_output) else hs))
    dims.append((hs if split_output else n_hidden, n_classes))
    return dims


def batch_capacity(rowptr_host, par_li, batch_size):
    """Host-side sizing: (max rows, max induced nnz bound) over any union of
    `batch_size` parts.  The nnz bound is the sum of FULL degrees of the rows."""
    deg = np.diff(np.asarray(rowptr_host, np.int64))
    sizes = np.array([len(p) for p in par_li], np.int64)
    degs = np.array([int(deg[np.asarray(p, np.int64)].sum()) for p in par_li], np.int64)
    k = min(batch_size, len(par_li))
    n_max = int(np.sort(sizes)[-k:].sum())
    nnz_max = int(np.sort(degs)[-k:].sum())
    return n_max, nnz_max
