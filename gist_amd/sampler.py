"""Cluster batch source with the reference's ClusterIter API (cluster_gcn/sampler.py:11-93,
cluster_gcn/partition_utils.py:11-25), device resident.

Same constructor arguments, same iteration protocol, and -- crucially -- the same
consumption of python's global `random` stream (one shuffle at construction,
sampler.py:55, one at the end of every epoch, :92), so batch order is identical to
the reference for the same seed and partition list.

Differences, all on the MI355X side of the boundary:
  * the train-induced graph (sampler.py:34) is extracted ON the GPU and stays there
    together with its features and labels;
  * a batch is the node-induced subgraph of the union of `batch_size` parts
    (partition_utils.py:20-25), built by HIP kernels from the resident graph -- the
    reference does this on the CPU inside the timed loop and uploads the result;
  * METIS is not bundled: partition lists come from the reference's own cache file
    `../data/{dn}_{psize}.npy` (sampler.py:44-51 format), from `par_li=`, or -- on a cache
    miss, like the reference -- from the library's own partitioner (gist_partition_graph),
    whose result is then cached in the same format.
"""
import os
import random

import numpy as np
import torch

from . import hip
from .graph import ClusterBatch


def freeze_setup_objects():
    """Everything a run has built by the time its iterator is bound -- the dataset, the partition lists, torch itself:
    ~10^6 Python objects -- lives until the process ends.  A full (generation 2) pass of the cyclic garbage collector
    over them takes ~70 ms on the host and lands in the middle of the training loop every few thousand iterations
    (measured: one such pass in a 300-step window, profiles/r05_module_path.md).  gc.freeze() moves them to the
    permanent generation: later collections only walk what the loop itself allocates.  A process-wide choice, so it is
    the APPLICATION's: GIST_GC_FREEZE=1 turns it on (bench.py and the gist_amd/scripts entry points set that default for
    their own process); a library user's collector is left alone."""
    if os.environ.get('GIST_GC_FREEZE', '0') == '1':
        import gc
        gc.collect()
        gc.freeze()


def load_partition_cache(path):
    """Reader for the reference's cache: object array of int64 node-id arrays."""
    arr = np.load(path, allow_pickle=True)
    return [np.asarray(p, np.int64).reshape(-1) for p in arr]


def save_partition_cache(path, par_li):
    """Writer producing the same on-disk format (wraps the ragged list as dtype=object,
    which numpy >= 1.24 requires)."""
    arr = np.empty(len(par_li), dtype=object)
    for i, p in enumerate(par_li):
        arr[i] = np.asarray(p, np.int64)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.save(path, arr, allow_pickle=True)


def get_partition_list(g, psize):
    """partition_utils.py:11-18: list of node-id arrays, one per part.  The parts come from the
    library's own partitioner (dgl_compat.transform.metis_partition), not from METIS."""
    from .dgl_compat import NID
    from .dgl_compat.backend import asnumpy
    from .dgl_compat.transform import metis_partition
    p_gs = metis_partition(g, psize)
    return [asnumpy(val.ndata[NID]) for _, val in p_gs.items()]


def get_subgraph(g, par_arr, i, psize, batch_size):
    """partition_utils.py:20-25, on the device graph."""
    parts = [par_arr[s] for s in range(i * batch_size, (i + 1) * batch_size) if s < psize]
    return g.subgraph(np.concatenate(parts).reshape(-1).astype(np.int64))


class ClusterIter(object):
    """The partition sampler given a graph and a partition number (sampler.py:11-56)."""

    def __init__(self, dn, g, psize, batch_size, seed_nid, use_pp=False, par_li=None,
                 device=None):
        self.use_pp = use_pp
        if device is None:
            device = g.device if g.device.type == 'cuda' else torch.device(
                'cuda', torch.cuda.current_device())
        gd = g if g.device == device else g.to(device)
        if torch.is_tensor(seed_nid):
            seed_nid = seed_nid.cpu().numpy()
        self.g = gd.subgraph(np.asarray(seed_nid, np.int64))         # sampler.py:34
        if use_pp:                                                   # sampler.py:37-40
            self.precalc(self.g)
        self.psize = psize
        self.batch_size = batch_size
        if par_li is not None:
            self.par_li = [np.asarray(p, np.int64).reshape(-1) for p in par_li]
        elif dn:
            fn = os.path.join('../data/', dn + '_{}.npy'.format(psize))   # sampler.py:45
            if os.path.exists(fn):
                self.par_li = load_partition_cache(fn)
            else:
                self.par_li = get_partition_list(self.g, psize)
                save_partition_cache(fn, self.par_li)
        else:
            self.par_li = get_partition_list(self.g, psize)
        self.max = int((psize) // batch_size)                        # sampler.py:54
        random.shuffle(self.par_li)                                  # sampler.py:55
        self.get_fn = get_subgraph
        self.n = 0
        # device feed (None = not tried yet): the epoch's part order uploaded once per epoch and every batch DESCRIBED
        # to the consumer, which extracts it on the device (gist_extract_parts_batch inside gist_sage_step) -- what
        # EngineClusterIter always does and what this class does for `model(cluster)` loops when the parts partition the
        # train graph (ClusterBatch below).  GIST_MODULE_ENGINE=0: every cluster built eagerly by g.subgraph
        self._feed = None
        self.batcher = None
        self.engine = None
        self.native = False
        self._node_part = None
        self.locality = None
        self.locality_stats = None

    def precalc(self, g):
        """sampler.py:58-69: the TRAIN graph's features become [X | A^ X] (A^ = mean over
        in-neighbours), computed once on the device by the aggregation kernel, so that a layer built
        with use_pp=True (gist_amd.modules.GraphSAGELayer, the reference's modules.py:100-159) skips
        its own aggregation while training.  Not usable with GCN / ISTSAGELayer, whose first layer
        aggregates itself (in the reference too: SURVEY.md appendix C.8)."""
        x = g.ndata['feat']
        n, f = x.shape
        z = torch.empty(n, 2 * f, dtype=torch.float32, device=x.device)
        hip.block_gather(x, None, None, z[:, :f])
        xs = x if f % 4 == 0 else z[:, :f]                     # any 2-D view with unit inner stride
        hip.spmm(g.rowptr, g.col, xs, z[:, f:], out_scale=g.norm())
        g.ndata['norm'] = g.norm().unsqueeze(1)
        g.ndata['feat'] = z

    def __len__(self):
        return self.max

    def feed(self):
        """True when this iterator describes its batches for on-device extraction (lazily set up on first use)."""
        if self._feed is None:
            self._feed = False
            tg = self.g
            ok = (os.environ.get('GIST_MODULE_ENGINE', '1') != '0' and not self.use_pp and tg.device.type == 'cuda'
                  and 'feat' in tg.ndata and 'label' in tg.ndata and tg.ndata['feat'].dim() == 2
                  and tg.ndata['feat'].dtype == torch.float32 and self.max > 0)
            if ok:
                self._init_feed()
                self._feed = self._node_part is not None and hip._lib.load().gist_extract_parts_supported(self.n_max) == 1
                if self._feed:
                    tm = tg.ndata.get('train_mask')
                    self._all_train = bool(tm.all().item()) if tm is not None else False
                    self._ones = torch.ones(self.n_max, dtype=torch.bool, device=tg.device)
        return self._feed

    def _bound_run_ahead(self):
        """An epoch that was left early (a `break`, a step cap) never reached the deferred check at its end, which is also
        what keeps the host at most one epoch ahead of the GPU: before the staging buffer of two epochs ago is rewritten,
        wait for the progress mark here instead."""
        if not getattr(self, '_exhausted', True) and self.engine is not None:
            self.engine.check_extract_deferred()
        self._exhausted = False

    def __iter__(self):
        self.n = 0
        if self.feed():
            self._bound_run_ahead()
            self._upload_epoch()
            self._epoch_cols = {}
        return self

    def batch_ids(self, i):
        parts = [self.par_li[s] for s in range(i * self.batch_size, (i + 1) * self.batch_size)
                 if s < self.psize]
        return np.concatenate(parts).reshape(-1).astype(np.int64)

    def __next__(self):
        if self.n < self.max:
            if self._feed:
                result = ClusterBatch(self, self.n)
            else:
                result = self.get_fn(self.g, self.par_li, self.n, self.psize, self.batch_size)
            self.n += 1
            return result
        if self._feed and self.engine is not None:
            self.engine.check_extract_deferred()
        self._exhausted = True
        random.shuffle(self.par_li)                                  # sampler.py:92
        raise StopIteration

    def epoch_column(self, key):
        """ndata[key] of the train graph in THIS epoch's batch order (one gather per epoch and key: a batch's rows are a
        slice of it, valid for the whole epoch whatever happens to the batch buffers)."""
        col = self._epoch_cols.get(key)
        if col is None:
            v = self.g.ndata[key]
            idx = self._epoch_cols.get('__ids64')
            if idx is None:
                idx = self._epoch_cols['__ids64'] = self._epoch_ids.to(torch.int64)
            col = self._epoch_cols[key] = v.index_select(0, idx)
        return col

    # ---- device feed: what the extraction kernels need, per run / per epoch ----------------------------------------
    def _init_feed(self):
        from .engine import ClusterBatcher, batch_capacity
        tg = self.g
        rowptr_host = tg.rowptr.cpu().numpy()
        self.n_max, self.nnz_max = batch_capacity(rowptr_host, self.par_li, self.batch_size)
        feat = tg.ndata['feat']
        lab = tg.ndata['label']
        if lab.dtype != torch.int32:
            lab = lab.to(torch.int32)
        self.batcher = ClusterBatcher(tg, feat.contiguous(), lab.contiguous(), self.n_max,
                                      self.nnz_max)
        self._epoch_ids = None
        self._offsets = None
        # One-launch extraction (gist_extract_parts_batch): every node's part and position in it.  Only
        # when the parts really partition the train graph (disjoint; the reference's METIS output is).
        self._node_part = None
        n_nodes = tg.number_of_nodes()
        total = sum(len(p) for p in self.par_li)
        if 0 < total <= n_nodes:
            part_of = np.full(n_nodes, -1, np.int32)
            pos = np.zeros(n_nodes, np.int32)
            ok = True
            for k, p in enumerate(self.par_li):
                if len(p) == 0:
                    continue
                if (part_of[p] != -1).any() or len(np.unique(p)) != len(p):
                    ok = False
                    break
                part_of[p] = k
                pos[p] = np.arange(len(p), dtype=np.int32)
            if ok:
                # nodes outside every part can never be in a batch: park them on a part id that no
                # epoch table maps to a batch
                part_of[part_of < 0] = len(self.par_li)
                self._part_of_host = part_of
                self._node_part = torch.from_numpy(np.stack([part_of, pos], 1).copy()).to(tg.device)   # [N, 2]
        self._extract_scratch = None
        self._sibling_keys = None
        self._epoch_siblings = None
        # Are the parts LOCALITY blocks?  The blocked aggregation kernels (a part's feature tile staged in
        # LDS, or its diagonal block as a dense counts x features product) pay only when most neighbours of a
        # batch row lie in the row's own part: on a batch without locality the matrix-core kernel gathers
        # every row in full and runs 1.6x SLOWER than the row-split kernel (profiles/r02_spmm_kernels_bench.txt).
        # Measured once on the device: the share of the train graph's edges inside a part, and the expected
        # number of a row's neighbours that fall in the OTHER parts of its batch (the dense kernel keeps at
        # most 8 of those per row).  Without locality the batches carry no row blocks and every aggregation
        # runs on gist_spmm_csr_f32.
        self.locality = None
        self.locality_stats = None
        if self._node_part is not None and tg.number_of_edges() > 0:
            rp = tg.rowptr.to(torch.int64)
            rows = torch.repeat_interleave(torch.arange(n_nodes, device=tg.device), rp[1:] - rp[:-1])
            po = self._node_part[:, 0]
            pr, pc = po[rows], po[tg.col.to(torch.int64)]
            intra = int((pr == pc).sum().item())
            # Sibling parts: pairs of parts joined by hundreds of edges (one community cut in two).  A batch that holds
            # both is prepared with their off-diagonal blocks as dense pairs (gist_step_plan.sibling_parts); which batches
            # those are is known from this table alone, on the host, once per epoch.  (Threshold = the prepare kernel's,
            # gist_spmm_pair_min_edges; a pair listed here that the kernel does not take costs a few empty workgroups.)
            cross = pr != pc
            K = np.int64(len(self.par_li) + 1)
            keys, cnt = torch.unique(pr[cross].to(torch.int64) * int(K) + pc[cross].to(torch.int64), return_counts=True)
            keys = keys[cnt >= int(hip._lib.load().gist_spmm_pair_min_edges())].cpu().numpy()
            self._sibling_keys = np.unique(np.concatenate([keys, (keys % K) * K + keys // K]))      # either direction
            del rows, pr, pc, cross
            nnz = tg.number_of_edges()
            inside = intra / float(nnz)
            n_parts = max(len(self.par_li), 2)
            share = (min(self.batch_size, n_parts) - 1) / float(n_parts - 1)      # of the OTHER parts, those in a row's batch
            outside_in_batch = (nnz - intra) / float(n_nodes) * share
            # of a batch row's neighbours INSIDE THE BATCH, the expected share in the row's own part: what the blocked kernels
            # live on.  (Round 5: the gate used to ask for half of ALL the train graph's edges inside a part; a partition
            # of a graph with mixing 0.3 and communities larger than a part keeps 0.48 of them, yet 0.99 of the IN-BATCH
            # edges -- a batch holds 20 of 1500 parts -- and ran the ungrouped kernel at 2.1x the time: profiles/r05_unplanted_graph.json)
            in_batch_inside = intra / max(intra + (nnz - intra) * share, 1.0)
            self.locality_stats = dict(edges_inside_parts=round(inside, 4),
                                       in_batch_edges_inside_parts=round(in_batch_inside, 4),
                                       outside_neighbours_per_batch_row=round(outside_in_batch, 3))
            self.locality = in_batch_inside >= 0.8 and outside_in_batch <= 4.0
            if os.environ.get('GIST_SPMM_LOCALITY') in ('0', '1'):      # dev override
                self.locality = os.environ['GIST_SPMM_LOCALITY'] == '1'

    def bind(self, engine, native=True):
        """Feed `engine`.  native=True attaches the C++ step driver: batches are then only
        DESCRIBED here (ids slice) and extracted inside gist_sage_step."""
        self.engine = engine
        self.native = bool(native) and engine.arena.grads is not None
        if getattr(self.batcher, 'feat_intra', None) is None and self._node_part is not None and self.locality \
                and engine.fuse and os.environ.get('GIST_STEP_PREAGG', '1') != '0':
            import time
            t0 = time.time()
            self.batcher.feat_intra = self._intra_part_sums()
            if self.batcher.feat_intra.is_cuda:
                torch.cuda.synchronize(self.batcher.feat_intra.device)
            self.intra_part_sums_seconds = time.time() - t0      # (once per run; bench.py reports it with its set-up times)
        if self.native:
            self.native = engine.attach_batcher(self.batcher) is not None
        freeze_setup_objects()
        return self

    def refresh_input_aggregation(self):
        """Recompute the part-internal neighbour sums after the input features were changed IN PLACE (they are summed once,
        at bind(): a run normally fixes its features -- scaler, --use-pp -- before it builds its iterator).  Same buffer, so
        an attached engine's plan stays valid."""
        fi = getattr(self.batcher, 'feat_intra', None)
        if fi is not None:
            fi.copy_(self._intra_part_sums())

    def _intra_part_sums(self):
        """feat_intra[v] = sum of feat[u] over the in-neighbours u of v INSIDE v's part (train graph), once per run.
        A batch is a union of whole parts, so this part of layer 0's aggregation is the same in every batch: the
        one-launch extraction adds the few neighbours in the batch's other parts and the norm
        (gist_extract_parts_desc.feat_intra, include/gist_hip.h)."""
        tg = self.batcher.g
        n = tg.number_of_nodes()
        rp = tg.rowptr.to(torch.int64)
        deg = rp[1:] - rp[:-1]
        rows = torch.repeat_interleave(torch.arange(n, device=tg.device), deg)
        po = self._node_part[:, 0]
        keep = po[rows] == po[tg.col.to(torch.int64)]
        cnt = torch.zeros(n, dtype=torch.int64, device=tg.device)
        cnt.index_add_(0, rows[keep], torch.ones_like(rows[keep]))
        rp_f = torch.zeros(n + 1, dtype=torch.int64, device=tg.device)
        rp_f[1:] = torch.cumsum(cnt, 0)
        col_f = tg.col[keep].contiguous()                    # (CSR order kept: a boolean mask preserves it)
        del rows, keep, cnt
        out = torch.zeros_like(self.batcher.feat)
        hip.spmm(rp_f.to(torch.int32), col_f, self.batcher.feat, out)
        return out

    def _host_epoch_tables(self):
        """Everything the device needs for one epoch, computed on the HOST from the (already shuffled) part order,
        vectorised: the epoch's node ids, per batch its row offsets and the row ranges of its parts (cut at 128 rows:
        the LDS-staged / block-dense aggregations take one block at a time), and -- for the one-launch extraction --
        per part the batch it belongs to this epoch and its first row there."""
        mx, bs = self.max, self.batch_size
        used = self.par_li[:mx * bs]
        sizes = np.fromiter((len(p) for p in used), np.int64, len(used))
        ids = np.concatenate(used).astype(np.int32) if len(used) else np.zeros(0, np.int32)
        per_batch = sizes.reshape(mx, bs).sum(1) if mx else sizes[:0]
        off = np.zeros(mx + 1, np.int64)
        np.cumsum(per_batch, out=off[1:])
        # first row of every used part inside its batch
        first = np.cumsum(sizes) - sizes - np.repeat(off[:-1], bs)
        # blocks: a part of sz rows gives ceil(sz / 128) block ends first + min(128 (j + 1), sz); every batch's list starts with 0
        nblk = (sizes + 127) // 128
        part_rep = np.repeat(np.arange(len(used)), nblk)
        j_in = np.arange(int(nblk.sum())) - np.repeat(np.cumsum(nblk) - nblk, nblk)
        ends = first[part_rep] + np.minimum(128 * (j_in + 1), sizes[part_rep])
        per_batch_blocks = nblk.reshape(mx, bs).sum(1) if mx else nblk[:0]
        boff = np.zeros(mx + 1, np.int64)
        np.cumsum(per_batch_blocks + 1, out=boff[1:])                   # + the leading 0 of every batch
        blocks = np.zeros(int(boff[-1]), np.int32)
        dst = np.arange(ends.shape[0]) + np.repeat(np.arange(mx), per_batch_blocks) + 1
        blocks[dst] = ends
        tab = None
        if self._node_part is not None:
            tab = np.full((len(self.par_li) + 1, 2), -1, np.int32)
            nz = np.flatnonzero(sizes > 0)
            firsts = np.fromiter((used[i][0] for i in nz), np.int64, len(nz))
            pid = self._part_of_host[firsts]
            tab[pid, 0] = (nz // bs).astype(np.int32)
            tab[pid, 1] = first[nz].astype(np.int32)
        # which batches of this epoch hold two sibling parts (see _init_feed)
        self._epoch_siblings = None
        sk = self._sibling_keys
        if sk is not None and self._node_part is not None and mx and all(len(p) for p in used):
            pid_all = self._part_of_host[np.fromiter((p[0] for p in used), np.int64, len(used))].astype(np.int64).reshape(mx, bs)
            if sk.size == 0:
                self._epoch_siblings = np.zeros(mx, bool)
            else:
                K = np.int64(len(self.par_li) + 1)
                pk = pid_all[:, :, None] * K + pid_all[:, None, :]
                self._epoch_siblings = np.isin(pk.reshape(mx, -1), sk).any(1)
        return ids, off, blocks, boff, tab

    def _upload_epoch(self):
        """ONE upload per epoch: the tables are built in a pinned staging buffer and a copy KERNEL (gist_copy_i32) moves
        them into the epoch's own set of device buffers (two sets, alternating: the previous epoch's last batches may
        still be executing).  No synchronisation: the pageable-memory copy of round 3 waited for the whole queue (30-65 us
        per step of an epoch of 75).  Both staging buffers are allocated at the FIRST call, directly as pinned memory
        (round 5): `torch.empty(n).pin_memory()` at the first epoch boundary ran a multi-threaded CPU copy whose OpenMP
        workers then spin-waited on every core -- inside a container with a CPU quota that exhausted the quota and the
        kernel froze the whole process for the rest of the 100-ms period (profiles/r05_module_path.md: 30-65 ms of idle GPU
        in ~half of all 150-step windows, on either host path)."""
        ids, off, blocks, boff, tab = self._host_epoch_tables()
        dev = self.g.device
        n_ids, n_blk = ids.shape[0], blocks.shape[0]
        n_tab = 0 if tab is None else tab.size
        # (every piece starts on a 16-byte boundary: the kernels read the part table as 8-byte pairs)
        o_blk = (n_ids + 3) // 4 * 4
        o_tab = (o_blk + n_blk + 3) // 4 * 4
        total = o_tab + n_tab
        par = getattr(self, '_epoch_parity', 0) ^ 1
        self._epoch_parity = par
        bufs = getattr(self, '_epoch_bufs', None)
        if bufs is None:
            bufs = self._epoch_bufs = [None, None]
        for q in ((0, 1) if bufs[par] is None else (par,)):
            if bufs[q] is None or bufs[q][0].numel() < total:
                cap = total + total // 8 + 64
                bufs[q] = (torch.empty(cap, dtype=torch.int32, pin_memory=True),
                           torch.empty(cap, dtype=torch.int32, device=dev))
        host, devbuf = bufs[par]
        # The staging buffer is rewritten here: the copy issued from it two epochs ago must be complete.  With a bound
        # engine that runs the one-launch extraction it is -- check_extract_deferred() at the end of the previous epoch
        # waited for an event recorded after it.  Otherwise nothing bounds the host's run-ahead: wait for the stream.
        # (No torch event of its own: the progress mark below it is the loop's only host <-> device handshake.)
        if self.engine is None or self.engine._extract_scratch is None:
            torch.cuda.current_stream(dev).synchronize()
        hv = host.numpy()
        hv[:n_ids] = ids
        hv[o_blk:o_blk + n_blk] = blocks
        if n_tab:
            hv[o_tab:total] = tab.ravel()
        hip.copy_i32_raw(host.data_ptr(), devbuf.data_ptr(), total)      # (a kernel reading the pinned buffer: no copy command)
        self._epoch_ids = devbuf[:n_ids]
        self._offsets = off
        self._epoch_blocks = devbuf[o_blk:o_blk + n_blk]
        self._block_offsets = boff
        self._part_tables = devbuf[o_tab:total].view(-1, 2) if n_tab else None

    def describe(self, j):
        """(ids, n, row_blocks, parts, next_info) of batch j of the current epoch: what Batch / ClusterBatch carry."""
        a, b = int(self._offsets[j]), int(self._offsets[j + 1])
        ids = self._epoch_ids[a:b]
        row_blocks = parts = next_info = None
        if self.locality is not False:       # (None: parts of unknown quality, e.g. overlapping: as before)
            row_blocks = self._epoch_blocks[int(self._block_offsets[j]):int(self._block_offsets[j + 1])]
        if self._part_tables is not None:
            parts = (self._node_part, self._part_tables, j)
            if j + 1 < self.max:      # the batch that follows in this epoch (SageEngine.prefetch)
                a2, b2 = int(self._offsets[j + 1]), int(self._offsets[j + 2])
                next_info = (self._epoch_ids[a2:b2], j + 1)
        return ids, b - a, row_blocks, parts, next_info

    def has_siblings(self, j):
        """May batch j of the current epoch hold two parts joined by hundreds of edges?  (True when unknown.)"""
        es = self._epoch_siblings
        return True if es is None else bool(es[j])


class EngineClusterIter(ClusterIter):
    """ClusterIter that feeds a SageEngine: yields engine Batches built in preallocated
    device buffers.  The epoch's part order is uploaded ONCE per epoch (one H2D of the
    permuted node ids); every batch is then a slice of that device array, so the
    training loop performs no per-iteration host<->device traffic."""

    def __init__(self, dn, g, psize, batch_size, seed_nid, engine_in_feats=None, **kw):
        super().__init__(dn, g, psize, batch_size, seed_nid, **kw)
        self._init_feed()
        self._feed = True

    def _extract_with_aggregation(self, ids, n):
        from . import hip as _hip
        from .engine import Batch
        bt, eng = self.batcher, self.engine
        L = _hip._lib.load()
        if self._extract_scratch is None:
            self._extract_scratch = torch.zeros(int(L.gist_extract_parts_scratch_bytes(self.n_max)) // 8 + 1,
                                                dtype=torch.int64, device=bt.feat.device)
        bt.prefetched = None
        f = bt.feat.shape[1]
        _hip.extract_parts(bt.g, ids, self.n_max, self._node_part, self._part_tables, self.n, bt.rowptr[:n + 1], bt.col,
                           bt.t_rowptr[:n + 1], bt.t_col, bt.norm, bt.feat, eng.z0_left(n), bt.labels, bt.lab,
                           self._extract_scratch, feat_intra=bt.feat_intra, ah=eng.Z[0][:n, f:2 * f])
        b = Batch()
        b.n, b.rowptr, b.col, b.t_rowptr, b.t_col = n, bt.rowptr[:n + 1], bt.col, bt.t_rowptr[:n + 1], bt.t_col
        b.norm, b.labels, b.ids = bt.norm[:n], bt.lab[:n], ids
        b.ah_owner = eng
        return b

    def fill_features(self, batch, engine):
        """Gather the current batch's features into ANOTHER engine's layer-0 buffer (several
        sub-GCNs trained in one process share one extracted batch)."""
        hip.gather_rows(self.batcher.feat, batch.ids, engine.z0_left(batch.n))

    def __iter__(self):
        self.n = 0
        self._bound_run_ahead()
        self._upload_epoch()
        return self

    def __next__(self):
        if self.n < self.max:
            ids, n, row_blocks, parts, next_info = self.describe(self.n)
            from . import hip as _hip
            if self.native and _hip._prof is None:
                batch = self.batcher.lazy(ids)
            elif (getattr(self.batcher, 'feat_intra', None) is not None and self._part_tables is not None
                  and self.engine.fuse and self.engine.dims[0][0] == self.batcher.feat.shape[1]):
                # the extraction the native step runs (one launch, layer 0's aggregation formed with it: it sums in its own
                # order, so the op-by-op path takes the same launch)
                batch = self._extract_with_aggregation(ids, n)
            else:
                batch = self.batcher.extract(ids, self.engine.z0_left(n))
            batch.row_blocks, batch.parts, batch.next_info = row_blocks, parts, next_info
            batch.siblings = self.has_siblings(self.n)
            self.n += 1
            return batch
        if self.engine is not None:
            # every batch of an epoch must have been extracted completely (gist_extract_parts_batch's error word).  The
            # word of the epoch that just ended is copied to the host asynchronously and inspected at the NEXT epoch end
            # (no queue drain per epoch); engine.check_extract() -- a synchronising read -- closes a run: the trainers
            # and bench.py call it before they report anything
            self.engine.check_extract_deferred()
        self._exhausted = True
        random.shuffle(self.par_li)
        raise StopIteration
