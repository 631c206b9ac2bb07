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

    def __iter__(self):
        self.n = 0
        return self

    def batch_ids(self, i):
        parts = [self.par_li[s] for s in range(i * self.batch_size, (i + 1) * self.batch_size)
                 if s < self.psize]
        return np.concatenate(parts).reshape(-1).astype(np.int64)

    def __next__(self):
        if self.n < self.max:
            result = self.get_fn(self.g, self.par_li, self.n, self.psize, self.batch_size)
            self.n += 1
            return result
        random.shuffle(self.par_li)                                  # sampler.py:92
        raise StopIteration


class EngineClusterIter(ClusterIter):
    """ClusterIter that feeds a SageEngine: yields engine Batches built in preallocated
    device buffers.  The epoch's part order is uploaded ONCE per epoch (one H2D of the
    permuted node ids); every batch is then a slice of that device array, so the
    training loop performs no per-iteration host<->device traffic."""

    def __init__(self, dn, g, psize, batch_size, seed_nid, engine_in_feats=None, **kw):
        super().__init__(dn, g, psize, batch_size, seed_nid, **kw)
        from .engine import ClusterBatcher, batch_capacity
        tg = self.g
        rowptr_host = tg.rowptr.cpu().numpy()
        self.n_max, self.nnz_max = batch_capacity(rowptr_host, self.par_li, batch_size)
        feat = tg.ndata['feat']
        lab = tg.ndata['label']
        if lab.dtype != torch.int32:
            lab = lab.to(torch.int32)
        self.batcher = ClusterBatcher(tg, feat.contiguous(), lab.contiguous(), self.n_max,
                                      self.nnz_max)
        self.engine = None
        self.native = False
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
            intra = int((po[rows] == po[tg.col.to(torch.int64)]).sum().item())
            del rows
            nnz = tg.number_of_edges()
            inside = intra / float(nnz)
            n_parts = max(len(self.par_li), 2)
            outside_in_batch = (nnz - intra) / float(n_nodes) * (min(batch_size, n_parts) - 1) / (n_parts - 1)
            self.locality_stats = dict(edges_inside_parts=round(inside, 4),
                                       outside_neighbours_per_batch_row=round(outside_in_batch, 3))
            self.locality = inside >= 0.5 and outside_in_batch <= 4.0
            if os.environ.get('GIST_SPMM_LOCALITY') in ('0', '1'):      # dev override
                self.locality = os.environ['GIST_SPMM_LOCALITY'] == '1'

    def bind(self, engine, native=True):
        """Feed `engine`.  native=True attaches the C++ step driver: batches are then only
        DESCRIBED here (ids slice) and extracted inside gist_sage_step."""
        self.engine = engine
        self.native = bool(native) and engine.arena.grads is not None
        if self.native:
            self.native = engine.attach_batcher(self.batcher) is not None
        return self

    def fill_features(self, batch, engine):
        """Gather the current batch's features into ANOTHER engine's layer-0 buffer (several
        sub-GCNs trained in one process share one extracted batch)."""
        hip.gather_rows(self.batcher.feat, batch.ids, engine.z0_left(batch.n))

    def _upload_epoch(self):
        used = self.par_li[:self.max * self.batch_size]
        sizes = np.array([len(p) for p in used], np.int64)
        ids = np.concatenate(used).astype(np.int32) if len(used) else np.zeros(0, np.int32)
        off = np.zeros(self.max + 1, np.int64)
        per_batch = sizes.reshape(self.max, self.batch_size).sum(1) if self.max else sizes[:0]
        np.cumsum(per_batch, out=off[1:])
        self._epoch_ids = torch.from_numpy(ids).to(self.g.device, non_blocking=False)
        self._offsets = off
        # locality blocks of every batch (row ranges of its METIS parts, cut at 128 rows: the
        # LDS-staged aggregation stages one block's feature tile at a time), uploaded with the ids
        blocks, boff = [], np.zeros(self.max + 1, np.int64)
        for i in range(self.max):
            edges, pos = [0], 0
            for sz in sizes[i * self.batch_size:(i + 1) * self.batch_size]:
                sz = int(sz)
                for c in range(0, sz, 128):
                    edges.append(pos + min(c + 128, sz))
                pos += sz
            blocks.append(np.asarray(edges, np.int32))
            boff[i + 1] = boff[i] + len(edges)
        self._epoch_blocks = torch.from_numpy(
            np.concatenate(blocks) if blocks else np.zeros(0, np.int32)).to(self.g.device)
        self._block_offsets = boff
        # which batch of this epoch each part belongs to, and the batch row of its first node
        self._part_tables = None
        if self._node_part is not None:
            n_parts = len(self.par_li) + 1
            tab = np.full((n_parts, 2), -1, np.int32)
            row = 0
            for s_, p in enumerate(used):
                j = s_ // self.batch_size
                if s_ % self.batch_size == 0:
                    row = 0
                if len(p):
                    pid = int(self._part_of_host[p[0]])
                    tab[pid, 0], tab[pid, 1] = j, row
                row += len(p)
            self._part_tables = torch.from_numpy(tab).to(self.g.device)

    def __iter__(self):
        self.n = 0
        if self.engine is not None:
            self.engine.check_extract()          # (last epoch's one-launch extractions all met their barrier)
        self._upload_epoch()
        return self

    def __next__(self):
        if self.n < self.max:
            a, b = int(self._offsets[self.n]), int(self._offsets[self.n + 1])
            ids = self._epoch_ids[a:b]
            from . import hip as _hip
            if self.native and _hip._prof is None:
                batch = self.batcher.lazy(ids)
            else:
                batch = self.batcher.extract(ids, self.engine.z0_left(b - a))
            if self.locality is not False:       # (None: parts of unknown quality, e.g. overlapping: as before)
                batch.row_blocks = self._epoch_blocks[int(self._block_offsets[self.n]):
                                                      int(self._block_offsets[self.n + 1])]
            if self._part_tables is not None:
                batch.parts = (self._node_part, self._part_tables, self.n)
            self.n += 1
            return batch
        if self.engine is not None:          # every batch of the epoch that just ended was extracted completely
            self.engine.check_extract()
        random.shuffle(self.par_li)
        raise StopIteration
