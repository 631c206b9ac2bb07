"""Training loops of the reference, driven through the SageEngine fast path.

  ClusterGCNTrainer   cluster_gcn/cluster_gcn.py:19-142      (single GPU, full width)
  (GIST trainer lives in gist_amd/ist.py next to the dispatch/sync code)

Timing follows the reference: `total_time` sums the epoch loops -- batch construction
included, evaluation excluded (cluster_gcn.py:91,106-111).
"""
import os
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
    Reddit at H = 32768: 2 x 30.5 GB + 4 GB instead of ~250 GB.

    node_blocks (int boundaries [0, b1, ..., N], blocks of at most 128 consecutive node ids -- the parts of a
    partition the graph's ids are ordered by): the aggregation is split once into A = A_diag + A_rest.
    A_diag, the edges inside a block, is a block-diagonal matrix whose blocks are DENSE on a clustered
    graph (Reddit-like: 246 of a row's 450 neighbours among its block's 102 rows), so its product runs as
    counts x features on the matrix cores (gist_spmm_csr_prepared_f32, exact fp32 as in training) instead of
    246 gathered rows per node; A_rest is gathered by the row kernel with accumulate, column tile by column
    tile so that the gathered slab of X stays in the Infinity Cache.  Row blocks are then cut at block
    boundaries.  None = the graph's own `node_blocks` attribute if it has one (the synthetic block
    datasets do), else one gather pass over A (any graph); False = one gather pass."""

    def __init__(self, g, dims, use_layernorm, arena, device, row_block=None,
                 block_bytes=4 << 30, node_blocks=None, pair_min_edges=300, cache_input_aggregation=True,
                 pair_bytes=2 << 30):
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
        self.split = None
        self.rest_tile = 512        # floats per column tile of the remainder's gather passes
        # Round 4: an OFF-diagonal pair of node blocks that shares at least this many edges is a dense block too and
        # runs as counts x features on the bf16x3 matrix cores (gist_spmm_block_units_f32: the batch kernel with the
        # X tile taken from the COLUMN block) instead of one gathered row per edge: 0.7 us of chip time per pair at
        # D = 4096 against 0.6-2.2 ns per gathered edge (cached neighbour parts ... uniform) = break-even ~300-1000.
        # A partition of a real graph cuts few block pairs heavily; the uniform block model cuts all pairs thinly
        # (10 edges per pair) and keeps the gathers.  0 = off.
        self.pair_min_edges = int(pair_min_edges) if os.environ.get('GIST_EVAL_PAIRS', '1') != '0' else 0
        self.pair_bytes = int(pair_bytes)
        self.use_chains = os.environ.get('GIST_EVAL_CHAINS', '1') != '0'      # (0: one launch per pair rank, round 4's form)
        self.row_cuts = list(range(0, n, self.row_block)) + [n]
        hidden = max([o for (i, o) in self.dims[:-1]] + [1])
        self.h = [torch.empty(n, hidden, **f32) for _ in range(2 if L1 > 2 else 1)] if L1 > 1 else []
        self.zb = torch.empty(self.row_block, 2 * max_in, **f32) if blocked else None
        self.yb = torch.empty(self.row_block, max_out, **f32) if blocked else None
        if node_blocks is None:
            node_blocks = getattr(g, 'node_blocks', None)      # a dataset whose ids are ordered by part says so
        if node_blocks is False:
            node_blocks = None                                  # explicit opt-out: one gather pass
        if node_blocks is not None and blocked and os.environ.get('GIST_EVAL_SPLIT', '1') != '0':
            self._split_graph(np.asarray(node_blocks, np.int64))
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
        # Layer 0 aggregates the INPUT features: the same product A^ X at every evaluation of a run (graph and features do
        # not change, the parameters do not enter).  Kept after the first forward ([N, F] floats: 370 MB for the
        # Reddit-like graph) and copied into the row blocks afterwards -- the same values bit for bit.
        # invalidate_input_aggregation() after changing g.ndata['feat'] in place.
        self.cache_input_aggregation = bool(cache_input_aggregation) and os.environ.get('GIST_EVAL_CACHE_AH0', '1') != '0'
        self._ah0 = None
        self._ah0_ready = False

    def invalidate_input_aggregation(self):
        self._ah0_ready = False

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
            keep_ah = k == 0 and self.cache_input_aggregation
            if keep_ah and self._ah0 is None:
                self._ah0 = torch.empty(n, i, dtype=torch.float32, device=self.device)
            for bi, (r0, r1) in enumerate(zip(self.row_cuts[:-1], self.row_cuts[1:])):
                z = self.zb[:r1 - r0, :2 * i]
                hip.block_gather(cur[r0:r1, :i], None, None, z[:, :i])
                if keep_ah and self._ah0_ready:
                    hip.block_gather(self._ah0[r0:r1], None, None, z[:, i:])
                elif self.split is not None and self._dense_ok(cur[r0:r1, :i], z[:, i:]):
                    sp = self.split
                    ch = sp.get('chains')
                    if ch is not None:
                        # inside the blocks AND the dense off-diagonal pairs: one chain of units per row block, the output
                        # tile's accumulators kept across the chain (gist_spmm_block_chains_f32)
                        cp, u_lo, u_hi = ch['chunks'][bi]
                        hip.spmm_block_chains(cp, ch['units'][u_lo:u_hi], ch['images'][u_lo:u_hi], cur[:, :i], z[:, i:],
                                              out_scale=self.norm[r0:r1])
                    else:
                        # inside the blocks: counts x features on the matrix cores (sources = this row block)
                        hip.spmm(sp['rowptr_d'][r0:r1 + 1], sp['col_d'], cur[r0:r1, :i], z[:, i:],
                                 out_scale=self.norm[r0:r1], row_blocks=sp['blocks'][bi], prepared=sp['prepared'][bi])
                    # everything else: gathered, accumulated -- one column tile at a time, so that the slab of
                    # X a pass gathers from (N x rest_tile floats: 238 MB at Reddit's size) stays in the
                    # Infinity Cache instead of every gather going to HBM
                    if ch is None and sp['n_pairs'] > 0:      # dense off-diagonal blocks: counts x features, accumulated --
                        # the j-th pair of every row block of this chunk in one launch (disjoint output rows)
                        for (a0, a1) in sp['rounds'].get(bi, []):
                            hip.spmm_block_units(sp['units'][a0:a1], sp['images'][a0:a1], cur[:, :i], z[:, i:],
                                                 out_scale=self.norm[r0:r1], accumulate=True)
                    ct = self.rest_tile
                    for c0 in range(0, i, ct if sp['rest_edges'] > 0 else i):
                        c1 = min(c0 + ct, i)
                        hip.spmm(sp['rowptr_r'][r0:r1 + 1], sp['col_r'], cur[:, c0:c1], z[:, i + c0:i + c1],
                                 out_scale=self.norm[r0:r1], accumulate=True)
                else:
                    hip.spmm(g.rowptr[r0:r1 + 1], g.col, cur[:, :i], z[:, i:], out_scale=self.norm[r0:r1])
                if keep_ah and not self._ah0_ready:
                    hip.block_gather(z[:, i:], None, None, self._ah0[r0:r1])
                if last:
                    hip.gemm_nt(z, W, b, dst[r0:r1, :o])
                else:
                    y = self.yb[:r1 - r0, :o]
                    hip.gemm_nt(z, W, b, y)
                    hip.ln_relu_fwd(y, dst[r0:r1, :o], None, self.use_layernorm, True)
            if keep_ah:
                self._ah0_ready = True
            cur = dst
        return self.logits[:, :self.n_classes]

    @staticmethod
    def _dense_ok(x, y):
        """The block-dense product takes 16-byte rows of a multiple of 4 floats, at least 128 wide."""
        d = x.shape[1]
        return (d >= 128 and d % 4 == 0 and x.stride(0) % 4 == 0 and y.stride(0) % 4 == 0 and
                x.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0)

    def _split_graph(self, bounds):
        """A = A_diag + A_rest for the node blocks `bounds`, once, on the device; row blocks cut at block
        boundaries; the block structure of every row block prepared for the matrix-core kernel."""
        g, n, dev = self.g, self.n, self.device
        if bounds[0] != 0 or bounds[-1] != n or (np.diff(bounds) <= 0).any() or (np.diff(bounds) > 128).any():
            raise ValueError('gist_amd: node_blocks must be increasing boundaries 0..N of blocks of 1..128 nodes')
        # row blocks: the last block boundary at or below r0 + row_block
        cuts = [0]
        while cuts[-1] < n:
            j = int(np.searchsorted(bounds, cuts[-1] + self.row_block, side='right')) - 1
            nxt = int(bounds[j])
            if nxt <= cuts[-1]:
                nxt = int(bounds[int(np.searchsorted(bounds, cuts[-1], side='right'))])
            cuts.append(nxt)
        self.row_cuts = cuts
        longest = max(b - a for a, b in zip(cuts[:-1], cuts[1:]))
        if longest > self.zb.shape[0]:
            f32 = dict(dtype=torch.float32, device=dev)
            self.zb = torch.empty(longest, self.zb.shape[1], **f32)
            self.yb = torch.empty(longest, self.yb.shape[1], **f32)
        bd = torch.from_numpy(bounds).to(dev)
        rowptr = g.rowptr.to(torch.int64)
        deg = rowptr[1:] - rowptr[:-1]
        rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
        col = g.col.to(torch.int64)
        blk_of = torch.bucketize(torch.arange(n, device=dev), bd, right=True) - 1
        inside = blk_of[rows] == blk_of[col]
        del blk_of
        cut_t = torch.tensor(cuts, dtype=torch.int64, device=dev)
        rb_of_row = torch.bucketize(torch.arange(n, device=dev), cut_t, right=True) - 1

        def csr(mask, local):
            r = rows[mask]
            c = col[mask]
            if local:                                   # sources relative to the row block's first row
                c = c - cut_t[rb_of_row[r]]
            cnt = torch.bincount(r, minlength=n)
            rp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
            torch.cumsum(cnt, 0, out=rp[1:])
            return rp.to(torch.int32), c.to(torch.int32).contiguous()

        rp_d, col_d = csr(inside, True)
        # ---- dense off-diagonal block pairs ----
        nb = len(bounds) - 1
        pairs = dict(n_pairs=0, pair_edges=0)
        rest_mask = ~inside
        if self.pair_min_edges > 0 and bool(rest_mask.any()):
            blk_of = torch.bucketize(torch.arange(n, device=dev), bd, right=True) - 1
            sel = torch.nonzero(rest_mask).squeeze(1)
            er, ec = rows[sel], col[sel]
            key = blk_of[er] * nb + blk_of[ec]
            uniq, inv, cnt = torch.unique(key, return_inverse=True, return_counts=True)
            dense = cnt >= self.pair_min_edges
            stride = int(hip._lib.load().gist_spmm_block_image_bytes()) // 2      # bf16 elements per image
            # the images stay in HBM for the evaluator's life (32 KiB per pair): at most pair_bytes of them, the heaviest
            # pairs first; the others stay with the gathers
            max_pairs = max(int(self.pair_bytes // (2 * stride)), 1)
            if int(dense.sum().item()) > max_pairs:
                thr = torch.topk(cnt[dense], max_pairs).values[-1]
                dense = dense & (cnt >= thr)
            for _ in range(2):                         # (second round only if a pair had a count above 256)
                if not bool(dense.any()):
                    break
                n_pairs = int(dense.sum().item())
                dk = uniq[dense]                       # sorted: row block major
                rb_p, cb_p = dk // nb, dk % nb
                first = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
                torch.cumsum(torch.bincount(rb_p, minlength=nb), 0, out=first[1:])
                j_p = torch.arange(n_pairs, device=dev) - first[rb_p]           # the pair's rank within its row block
                chunk_p = rb_of_row[bd[rb_p]]                                    # row chunk of the evaluator's sweep
                # launch order: (row chunk, rank, row block) -- the units of one launch have disjoint output rows
                n_j = int(j_p.max().item()) + 1
                order = torch.argsort((chunk_p * n_j + j_p) * nb + rb_p)
                pos = torch.empty_like(order)
                pos[order] = torch.arange(n_pairs, device=dev)
                rank = torch.full((uniq.numel(),), -1, dtype=torch.int64, device=dev)
                rank[dense] = pos
                e_dense = dense[inv]
                p_e = rank[inv][e_dense]
                r_loc = er[e_dense] - bd[blk_of[er[e_dense]]]
                k_loc = ec[e_dense] - bd[blk_of[ec[e_dense]]]
                flat = p_e * 16384 + ((k_loc >> 3) * 128 + r_loc) * 8 + (k_loc & 7)
                # the count images, 2048 pairs at a time (an int32 scatter-add into 128 MB of scratch: one bincount over all
                # pairs was 128 KB of int64 per pair -- 2.4 GB in flight at 18 000 pairs; advisor, round 4)
                images = torch.zeros(n_pairs, stride, dtype=torch.bfloat16, device=dev)
                big = torch.zeros(n_pairs, dtype=torch.bool, device=dev)
                eorder = torch.argsort(p_e)
                flat_s, pe_s = flat[eorder], p_e[eorder]
                ebounds = torch.searchsorted(pe_s, torch.arange(0, n_pairs + 2048, 2048, device=dev)).tolist()
                one = torch.ones(1, dtype=torch.int32, device=dev)
                for ci, p0 in enumerate(range(0, n_pairs, 2048)):
                    p1 = min(n_pairs, p0 + 2048)
                    cnt32 = torch.zeros((p1 - p0) * 16384, dtype=torch.int32, device=dev)
                    sl = flat_s[ebounds[ci]:ebounds[ci + 1]] - p0 * 16384
                    cnt32.index_add_(0, sl, one.expand(sl.numel()))
                    cnt32 = cnt32.view(p1 - p0, 16384)
                    big[p0:p1] = cnt32.max(1).values > 256       # not exact in bf16: back to the gathers
                    images[p0:p1, :16384] = cnt32.to(torch.float32).to(torch.bfloat16)
                    del cnt32
                del flat, flat_s, pe_s, eorder
                if bool(big.any()):
                    idx = torch.nonzero(dense).squeeze(1)[pos.argsort()[big]]
                    dense[idx] = False
                    del images
                    continue
                rb_o, cb_o, j_o, ch_o = rb_p[order], cb_p[order], j_p[order], chunk_p[order]
                base = cut_t[ch_o]
                units = torch.stack([bd[rb_o] - base, bd[rb_o + 1] - base, bd[cb_o], bd[cb_o + 1]], 1).to(torch.int32).contiguous()
                # slices of `units` / `images` per (row chunk, rank)
                keys = (ch_o * n_j + j_o).cpu().numpy()
                cutsk = np.flatnonzero(np.diff(np.concatenate([[-1], keys, [-2]])) != 0)
                rounds = {}
                for a0, a1 in zip(cutsk[:-1], cutsk[1:]):
                    rounds.setdefault(int(keys[a0]) // n_j, []).append((int(a0), int(a1)))
                pair_ptr = first.to(torch.int32)
                pairs = dict(n_pairs=n_pairs, pair_edges=int(e_dense.sum().item()), images=images, units=units,
                             rounds=rounds, pair_ptr=pair_ptr, pair_cb=cb_p.to(torch.int32).contiguous(),
                             pair_order=order)
                rest_mask = rest_mask.clone()
                rest_mask[sel[e_dense]] = False
                break
            del key, uniq, inv, cnt, er, ec, sel, blk_of
        rp_r, col_r = csr(rest_mask, False)
        del rows, col, inside, rest_mask
        blocks, prepared, block_range = [], [], []
        for r0, r1 in zip(cuts[:-1], cuts[1:]):
            lo = int(np.searchsorted(bounds, r0))
            hi = int(np.searchsorted(bounds, r1))
            rb = torch.from_numpy((bounds[lo:hi + 1] - r0).astype(np.int32)).to(dev)
            blocks.append(rb)
            block_range.append((lo, hi))
            # (row pointers of a row block are absolute offsets into col_d: prepare on the slice)
            prepared.append(hip.spmm_prepare(rp_d[r0:r1 + 1], col_d, rb))
        self.split = dict(rowptr_d=rp_d, col_d=col_d, rowptr_r=rp_r, col_r=col_r, blocks=blocks,
                          prepared=prepared, diag_edges=int(col_d.numel()), rest_edges=int(col_r.numel()),
                          block_range=block_range, bounds32=torch.from_numpy(bounds.astype(np.int32)).to(dev), **pairs)
        if self.use_chains and pairs['n_pairs'] > 0:
            self._build_chains(bd, cut_t, rb_of_row, nb, stride)

    def _build_chains(self, bd, cut_t, rb_of_row, nb, stride):
        """Round 5: a row block's diagonal block and its dense off-diagonal pairs as ONE chain of units for
        gist_spmm_block_chains_f32 -- one workgroup keeps the output tile's accumulators across the chain and writes y once
        (the per-pair launches read and wrote y for every pair: 128 of the 192 KB a pair-tile moved)."""
        sp, dev = self.split, self.device
        n_pairs = sp['n_pairs']
        first = sp['pair_ptr'].to(torch.int64)                              # [nb + 1]: sorted pairs of row block b
        rb_p = torch.repeat_interleave(torch.arange(nb, device=dev), first[1:] - first[:-1])
        cb_p = sp['pair_cb'].to(torch.int64)
        pos = torch.empty(n_pairs, dtype=torch.int64, device=dev)           # sorted pair -> its image in launch order
        pos[sp['pair_order']] = torch.arange(n_pairs, device=dev)
        # diagonal images: the prepared structures of the row chunks, block by block; a row that left the dense product
        # there (a count above 256: its rem_cnt is -1) keeps the per-launch path for the whole evaluator
        # (a prepared buffer = one record per block, then -- small block counts -- room for pair images: records only)
        diag = torch.cat([pb[:(rb.numel() - 1) * 2 * stride].view(torch.bfloat16).view(-1, stride)
                          for pb, rb in zip(sp['prepared'], sp['blocks'])], 0)
        rem = diag[:, 16384:16384 + 256].contiguous().view(torch.int32)       # rem_cnt[128] of every block
        if diag.shape[0] != nb or bool((rem < 0).any()):
            return
        n_units = n_pairs + nb
        slot_d = first[:-1] + torch.arange(nb, device=dev)                   # chain b starts with its diagonal block
        slot_p = torch.arange(n_pairs, device=dev) + rb_p + 1
        images = torch.empty(n_units, stride, dtype=torch.bfloat16, device=dev)
        images[slot_d] = diag
        for p0 in range(0, n_pairs, 4096):                                   # (gathered in pieces: 128 MB in flight)
            p1 = min(n_pairs, p0 + 4096)
            images[slot_p[p0:p1]] = sp['images'][pos[p0:p1]]
        base_b = cut_t[rb_of_row[bd[:-1]]]                                   # first row of the chunk a block belongs to
        units = torch.empty(n_units, 4, dtype=torch.int64, device=dev)
        units[slot_d] = torch.stack([bd[:-1] - base_b, bd[1:] - base_b, bd[:-1], bd[1:]], 1)
        units[slot_p] = torch.stack([bd[rb_p] - base_b[rb_p], bd[rb_p + 1] - base_b[rb_p], bd[cb_p], bd[cb_p + 1]], 1)
        cptr = torch.cat([slot_d, torch.tensor([n_units], device=dev)])
        chunks = []
        for (lo, hi) in sp['block_range']:
            u_lo = int(cptr[lo].item())
            chunks.append(((cptr[lo:hi + 1] - u_lo).to(torch.int32).contiguous(), u_lo, int(cptr[hi].item())))
        sp['chains'] = dict(images=images, units=units.to(torch.int32).contiguous(), chunks=chunks)
        sp['images'] = None                                                  # (the chains own the pair images now)

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
        self.engine.prefetch = True      # (train_epoch only reads the loss of a step)
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
        self.engine.check_extract()      # (outside the timed interval; the device is idle here anyway)
        return loss

    def evaluate(self, mask_name):
        if self.evaluator is None:
            self.evaluator = FullGraphEvaluator(self.g_host, self.dims, self.use_layernorm,
                                                self.engine.arena, self.device)
        return self.evaluator.accuracy(mask_name)
