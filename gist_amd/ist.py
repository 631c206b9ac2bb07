"""IST / GIST orchestration: feature-dimension partition of a GraphSAGE model into S
independent sub-GCNs, local training, periodic weight sync.

Reference: cluster_gcn/cluster_gcn_ist_distrib.py
  create_partition            :51-65
  DistributedGNNWrapper       :68-367   (sample_partitions, ini_sync_dispatch_model,
                                         dispatch_model, sync_model)
  train                       :370-479

What is kept identical (parity surface): the partition sampler (python `random`,
same call order on every rank), which block of which base tensor each site owns
(SURVEY.md appendix B), the shared last-layer bias averaging, the schedule quirks
(no re-dispatch in epoch 0, fresh Adam at every dispatch point, sync at multiples
of iter_per_site and at the very last iteration), rank 0's base model after every
sync, the five-line stdout contract.

What is re-designed for one MI355X node (8 GPUs, xGMI full mesh, 288 GB each):
  * the reference is a parameter server: rank 0 owns the base model and moves every
    block with its own 2-rank broadcast inside a freshly created/destroyed process
    group -- (S-1)(2L+1) serial round trips per sync and again per dispatch.
    Here every rank keeps a REPLICA of the base model in HBM (8.8 GB at H=32768);
    a sync is ONE RCCL all-gather of the sub-models' flat parameter arenas (equal
    size on every rank, each block crosses each link once) followed by on-device
    index scatters; a dispatch needs NO communication at all (local index gather).
  * the shared bias mean is computed from the gathered copies in site order, so it
    is bitwise identical on every rank.
"""
import random as _pyrandom
import time

import torch
import torch.distributed as dist

from .engine import ParamArena, SageEngine, dims_for


def create_partition(num_subnet, size, rng=_pyrandom):
    """cluster_gcn_ist_distrib.py:51-65: shuffle range(size) with python's `random`, deal
    round-robin to the sites; returns [(idx, full_idx)] as LongTensors.  The consumption of
    `random` (one shuffle of a `size`-long list) and the deal order are the parity surface."""
    order = list(range(size))
    rng.shuffle(order)
    # site s takes order[s], order[s + S], order[s + 2S], ... (round-robin deal)
    out = []
    for s in range(num_subnet):
        own = torch.LongTensor(order[s::num_subnet])
        out.append((own, torch.cat((own, own + size))))
    return out


class HipBlocks(object):
    """Block movers on the HIP kernels (the product path)."""

    def __init__(self):
        from . import hip
        self.hip = hip

    def gather(self, src, row_idx, col_idx, dst):
        self.hip.block_gather(src, row_idx, col_idx, dst)

    def scatter(self, src, row_idx, col_idx, dst):
        self.hip.block_scatter(src, row_idx, col_idx, dst)

    def mean_rows(self, src_flat, stride, n_src, n, out):
        self.hip.mean_rows(src_flat, stride, n_src, n, out)


class TorchDistComm(object):
    """Collectives over torch.distributed (backend 'nccl' == RCCL on ROCm)."""

    def __init__(self, group=None):
        self.group = group

    def world_size(self):
        return dist.get_world_size(self.group)

    def rank(self):
        return dist.get_rank(self.group)

    def all_gather_flat(self, out, inp):
        # one collective; a failure (RCCL error, wrong sizes) propagates to the caller
        dist.all_gather_into_tensor(out, inp, group=self.group)

    def broadcast(self, t, src=0):
        dist.broadcast(t, src=src, group=self.group)

    def barrier(self):
        dist.barrier(group=self.group)


class LocalCommGroup(object):
    """All S sites inside ONE process on one GPU (the reference's own launcher puts every
    rank on `--cuda-id 0`, script/reddit/run_ist_distrib.sh:16-18).  The 'all-gather' is a
    device copy of each registered sub arena; the caller runs the sites' steps in turn."""

    def __init__(self, n_sites):
        self.n = n_sites
        self.subs = [None] * n_sites
        self.bases = [None] * n_sites

    def handle(self, rank):
        return LocalComm(self, rank)


class LocalComm(object):
    def __init__(self, group, rank):
        self.group, self._rank = group, rank

    def register(self, base, sub):
        self.group.bases[self._rank] = base
        self.group.subs[self._rank] = sub

    def world_size(self):
        return self.group.n

    def rank(self):
        return self._rank

    def all_gather_flat(self, out, inp):
        n = inp.numel()
        for s, a in enumerate(self.group.subs):
            out[s * n:(s + 1) * n].copy_(a.params)

    def broadcast(self, t, src=0):
        if self._rank != src:
            t.copy_(self.group.bases[src].params)

    def barrier(self):
        pass


class DistributedGNNWrapper(object):
    """One rank's view of GIST: a replica of the base model + its sub-model.

    Constructor mirrors the reference (`args` needs num_subnet, n_hidden, n_layers, rank,
    dropout, use_layernorm).  `base_init` = [(W,b)] full-width parameters on rank 0
    (others pass None and receive them in ini_sync_dispatch_model)."""

    def __init__(self, args, g, in_feats, n_classes, device, base_init=None, blocks=None,
                 comm=None, n_max=None, seed=0):
        self.args = args
        self.g = g
        self.in_feats, self.n_classes = in_feats, n_classes
        self.device = device
        self.S, self.H, self.L = args.num_subnet, args.n_hidden, args.n_layers
        assert self.H % self.S == 0
        self.h = self.H // self.S
        self.rank = args.rank
        self.blocks = blocks if blocks is not None else HipBlocks()
        self.comm = comm if comm is not None else TorchDistComm()
        self.base_dims = dims_for(in_feats, self.H, n_classes, self.L)
        self.sub_dims = dims_for(in_feats, self.H, n_classes, self.L, split_output=True,
                                 num_subnet=self.S)
        self.base = ParamArena(self.base_dims, device, with_grads=False)
        if base_init is not None:
            self.base.load(base_init)
        self.sub = ParamArena(self.sub_dims, device)
        self.gathered = torch.zeros(self.S * self.sub.numel, dtype=torch.float32, device=device)
        if hasattr(self.comm, 'register'):
            self.comm.register(self.base, self.sub)
        self.current_partition = None
        self._idx = None
        self.engine = None
        if n_max is not None:
            self.engine = SageEngine(self.sub_dims, args.use_layernorm, args.dropout, n_max,
                                     device, seed=seed * 131 + self.rank, arena=self.sub)

    # -- partitions ------------------------------------------------------------------
    def sample_partitions(self):
        """:93-98 -- one create_partition per hidden layer, python `random` stream."""
        return [create_partition(self.S, self.H) for _ in range(self.L)]

    def _set_partition(self, part):
        self.current_partition = part
        dev = self.device
        self._idx = [[(idx.to(torch.int32).to(dev), full.to(torch.int32).to(dev))
                      for (idx, full) in layer] for layer in part]

    def _block_index(self, k, site):
        """(row_idx, col_idx) of site's block in base W_k; bias index for b_k (None = shared)."""
        L = self.L
        if k == 0:
            idx, _ = self._idx[0][site]
            return idx, None, idx
        if k == L:
            _, full = self._idx[L - 1][site]
            return None, full, None
        _, full_prev = self._idx[k - 1][site]
        nxt, _ = self._idx[k][site]
        return nxt, full_prev, nxt

    # -- dispatch ----------------------------------------------------------------------
    def _gather_own(self):
        """Slice the (local replica of the) base model into this rank's sub-model
        (:203-226 / :291-313 and the broadcast payloads :231-283 / :315-365)."""
        for k in range(self.L + 1):
            rows, cols, bidx = self._block_index(k, self.rank)
            self.blocks.gather(self.base.W[k], rows, cols, self.sub.W[k])
            self.blocks.gather(self.base.b[k].view(1, -1), None, bidx, self.sub.b[k].view(1, -1))

    def ini_sync_dispatch_model(self, part=None):
        """:197-283.  The base model leaves rank 0 once (replication), then every rank
        slices its own sub-model locally.  `part` lets a single-process multi-site driver
        sample the partition ONCE for all its sites (one `random` stream per process)."""
        part = part if part is not None else self.sample_partitions()
        if self.comm.world_size() > 1:
            self.comm.broadcast(self.base.params, src=0)
        self._set_partition(part)
        self._gather_own()

    def dispatch_model(self, part=None):
        """:285-367 -- new partition, local gather, no communication."""
        self._set_partition(part if part is not None else self.sample_partitions())
        self._gather_own()

    # -- sync --------------------------------------------------------------------------
    def sync_gather(self):
        """Phase 1 of sync_model: collect every site's flat sub arena (the one collective)."""
        P = self.sub.numel
        if self.comm.world_size() > 1:
            self.comm.all_gather_flat(self.gathered, self.sub.params)
        else:
            self.gathered[:P].copy_(self.sub.params)

    def sync_apply(self):
        """Phase 2: index-scatter all S sites' blocks into the local base replica; the
        shared last bias becomes the mean over sites (:103) -- also in the sub-model,
        as the reference's in-place all-reduce does."""
        P = self.sub.numel
        L = self.L
        for s in range(self.S):
            site = self.gathered[s * P:(s + 1) * P]
            for k in range(L + 1):
                (i, o), (w0, b0) = self.sub_dims[k], self.sub.offsets[k]
                rows, cols, bidx = self._block_index(k, s)
                self.blocks.scatter(site[w0:b0].view(o, 2 * i), rows, cols, self.base.W[k])
                if k < L:
                    self.blocks.scatter(site[b0:b0 + o].view(1, o), None, bidx,
                                        self.base.b[k].view(1, -1))
        # shared output bias: mean of the S copies, in site order (bitwise equal on all ranks)
        w0, b0 = self.sub.offsets[L]
        C = self.n_classes
        self.blocks.mean_rows(self.gathered[b0:], P, self.S, C, self.base.b[L])
        self.sub.b[L].copy_(self.base.b[L])

    def sync_model(self):
        """:100-195.  One all-gather of the flat sub arenas, then on-device scatters."""
        self.sync_gather()
        self.sync_apply()


def train(ist_model, args, cluster_iterator, evaluator=None, log=print):
    """The GIST loop, cluster_gcn_ist_distrib.py:370-479, on the engine fast path.

    `ist_model` is this rank's DistributedGNNWrapper -- or a LIST of S wrappers sharing a
    LocalCommGroup, in which case all sites run in this one process on one GPU (the
    reference's own launcher puts every rank on `--cuda-id 0`); the partition is then
    sampled once per dispatch, exactly one `random` stream per process as in the reference.
    `cluster_iterator` is an EngineClusterIter bound to the first wrapper's engine;
    `evaluator` (rank 0) exposes accuracy(mask_name) on the base replica.
    Returns total_time, per-site per-iteration device losses, accuracies, event log."""
    models = list(ist_model) if isinstance(ist_model, (list, tuple)) else [ist_model]
    local = len(models) > 1
    # one sub-GCN per process (the distributed run): a step's optimiser launch may extract the next batch of the epoch
    # beside it -- the loop only reads the loss.  Several sub-GCNs in one process share the extracted batch: not then
    for m in models:
        if m.engine is not None:
            m.engine.prefetch = not local
    comm = models[0].comm
    multi = (not local) and comm.world_size() > 1
    is_rank0 = models[0].rank == 0
    local_epochs = args.n_epochs // args.num_subnet                      # :385
    losses = [[] for _ in models]
    events, val_accs, test_accs, trn_losses = [], [], [], []
    loss_mark = 0
    total_iter, total_time = 0, 0.0
    n_iters = len(cluster_iterator)
    dev = models[0].device
    sync_dev = (lambda: torch.cuda.synchronize(dev)) if dev.type == 'cuda' else (lambda: None)
    sync_dev()
    start_time = time.time()
    for e in range(local_epochs):
        log('%d: running epoch %d / %d' % (models[0].rank, e, local_epochs))
        run_eval = True
        for j, batch in enumerate(cluster_iterator):
            if total_iter % args.iter_per_site == 0:                     # :400
                if e > 0:
                    if multi:
                        comm.barrier()
                    part = models[0].sample_partitions() if local else None
                    for m in models:
                        m.dispatch_model(part)                           # :401-403
                    events.append('dispatch')
                for m in models:
                    m.sub.reset_optimizer()                              # :404-407
            for si, m in enumerate(models):                              # :408-417
                if si > 0:
                    cluster_iterator.fill_features(batch, m.engine)
                loss = m.engine.train_step(batch, args.lr, args.weight_decay)
                losses[si].append(loss.clone())
            events.append('step')
            total_iter += 1
            last = (j == n_iters - 1) and (e == local_epochs - 1)
            if total_iter % args.iter_per_site == 0 or last:             # :422-427
                if multi:
                    comm.barrier()
                for m in models:
                    m.sync_gather()
                for m in models:
                    m.sync_apply()
                events.append('sync')
                if run_eval or last:                                     # :431-450
                    sync_dev()
                    total_time += time.time() - start_time
                    for m in models:                 # (the device is idle: every extraction so far was complete)
                        if m.engine is not None:
                            m.engine.check_extract()
                    run_eval = False
                    events.append('eval')
                    if is_rank0 and evaluator is not None:
                        val_accs.append(evaluator.accuracy('val_mask'))
                        test_accs.append(evaluator.accuracy('test_mask'))
                        # :432-433,446 -- mean training loss of rank 0 since the last evaluation
                        seg = losses[0][loss_mark:]
                        trn_losses.append(float(torch.stack(seg).mean().item()) if seg else 0.0)
                        loss_mark = len(losses[0])
                    sync_dev()
                    start_time = time.time()
    if multi:
        comm.barrier()
    return dict(total_time=total_time, losses=losses, events=events, val_accs=val_accs,
                test_accs=test_accs, trn_losses=trn_losses)


def print_results(res, log=print):
    """The five lines sweeps scrape (cluster_gcn_ist_distrib.py:475-479)."""
    log('Training Time: %.4f' % res['total_time'])
    log('Last Val: %.4f' % res['val_accs'][-1])
    log('Best Val: %.4f' % max(res['val_accs']))
    log('Last Test: %.4f' % res['test_accs'][-1])
    log('Best Test: %.4f' % max(res['test_accs']))
