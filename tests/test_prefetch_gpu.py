"""GPU: the NEXT batch of the epoch extracted beside a training step's optimiser launch (SageEngine.prefetch, round 4:
GIST_STEP_EXTRACT_NEXT / GIST_STEP_PREEXTRACTED, gist_adam_segments_extract_f32 = gist_adam_segments_f32 and
gist_extract_parts_batch in one grid) against the same run with every batch extracted at the start of its own step.
Same kernels' arithmetic on the same data: losses of every step, parameters, both Adam moments and the gradient arena
must agree BIT FOR BIT over two epochs (the first batch of an epoch is never prefetched), with and without dropout
(layer 0's mask folded into the prefetched feature gather), for an odd dropout offset parity (no fold), and when a
forward-only call or another batch comes between two training steps (the prefetched batch is discarded).

Reference: the cluster batch of sampler.py:85-93 and optimizer.step() of cluster_gcn_ist_distrib.py:413-415."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _engine(p_drop, n_layers, n_feats, hidden, parts_per_batch=5, seed=9):
    from gist_amd import datasets
    from gist_amd.engine import SageEngine, dims_for
    from gist_amd.sampler import EngineClusterIter
    random.seed(4)
    ds = datasets.toy(seed=seed, n=3000, n_blocks=30, n_feats=n_feats, n_classes=6, train_frac=1.0)
    g = ds.g
    it = EngineClusterIter('toy', g, len(ds.par_li), parts_per_batch, np.arange(g.number_of_nodes(), dtype=np.int64),
                           par_li=[p.copy() for p in ds.par_li], device=DEV)
    dims = dims_for(n_feats, hidden, 6, n_layers)
    eng = SageEngine(dims, True, p_drop, it.n_max, DEV, seed=11)
    gen = torch.Generator().manual_seed(1)
    for k, (i, o) in enumerate(dims):
        s_ = 1.0 / np.sqrt(2 * i)
        eng.arena.W[k].copy_((torch.rand(o, 2 * i, generator=gen) - 0.5) * 2 * s_)
        eng.arena.b[k].copy_((torch.rand(o, generator=gen) - 0.5) * 2 * s_)
    it.bind(eng)
    return eng, it


def _run(prefetch, p_drop, n_layers, n_feats, hidden, epochs=2, disturb=None):
    eng, it = _engine(p_drop, n_layers, n_feats, hidden)
    eng.prefetch = prefetch
    losses, used = [], 0
    for ep in range(epochs):
        for j, b in enumerate(it):
            had = it.batcher.prefetched is not None
            losses.append(eng.train_step(b, 0.01, 5e-4).clone())
            used += int(had)
            if disturb is not None and j == 2:
                disturb(eng, it, b)
    eng.check_extract()
    A = eng.arena
    return (torch.stack(losses), A.params.clone(), A.exp_avg.clone(), A.exp_avg_sq.clone(), A.grads.clone()), used


@pytest.mark.parametrize('p_drop,n_layers,n_feats,hidden', [
    (0.2, 2, 302, 512),        # layer 0's mask folded into the prefetched feature gather
    (0.0, 2, 302, 512),
    (0.2, 4, 100, 256),
    (0.2, 1, 301, 64),         # odd widths: some steps start at an odd mask offset (no fold for them)
])
def test_prefetched_extraction_trains_bit_identically(p_drop, n_layers, n_feats, hidden):
    ref, used0 = _run(False, p_drop, n_layers, n_feats, hidden)
    got, used1 = _run(True, p_drop, n_layers, n_feats, hidden)
    assert used0 == 0
    assert used1 == 2 * 5            # 6 batches per epoch: every batch but the first of an epoch came prefetched
    for name, a, b in zip(('losses', 'params', 'exp_avg', 'exp_avg_sq', 'grads'), got, ref):
        assert torch.equal(a, b), (name, (a - b).abs().max().item())


def test_prefetched_batch_is_discarded_when_something_else_uses_the_buffers():
    """After step 2 of each epoch: a forward-only evaluation of ANOTHER batch through the engine's op-by-op path
    (it extracts into the same buffers).  The next training step must notice and extract its batch itself."""
    def disturb(eng, it, b):
        ids = b.ids
        other = it.batcher.extract(ids, eng.z0_left(ids.numel()))
        eng.forward(other, training=False)

    ref, _ = _run(False, 0.2, 2, 302, 512, disturb=disturb)
    got, used = _run(True, 0.2, 2, 302, 512, disturb=disturb)
    assert used == 2 * 4             # (the batch after the disturbance was not taken from the prefetch)
    for name, a, b in zip(('losses', 'params', 'exp_avg', 'exp_avg_sq', 'grads'), got, ref):
        assert torch.equal(a, b), (name, (a - b).abs().max().item())


def test_fused_optimiser_and_extraction_launch_equals_the_two_calls():
    """gist_adam_segments_extract_f32 through the C ABI: the optimiser's results and the extracted batch equal those of
    gist_adam_segments_f32 + gist_extract_parts_batch (chunk-sum segment with dedicated blocks, a slab segment, loss)."""
    import ctypes
    from gist_amd import _lib, hip
    L = _lib.load()
    eng, it = _engine(0.2, 2, 302, 512)
    batches = list(it)
    b0, b1 = batches[0], batches[1]
    eng.train_step(b0, 0.01, 5e-4)                      # fills the plan (tables, scratch) and the gradient arena
    torch.cuda.synchronize()
    P = eng.plan
    A = eng.arena
    n = A.params.numel()
    gen = torch.Generator(device=DEV).manual_seed(5)
    grad = torch.randn(n, device=DEV, generator=gen)
    chunks = 40
    seg_b = (A.offsets[0][1], A.offsets[0][1] + 512)      # layer 0's bias as 40 chunk sums
    src_b = torch.randn(chunks, 512, device=DEV, generator=gen)
    seg_w = (A.offsets[1][0], A.offsets[1][0] + 6 * 1024)  # the class layer's weight as 3 slabs
    src_w = torch.randn(3, 6 * 1024, device=DEV, generator=gen)
    row_loss = torch.rand(b0.n, device=DEV, generator=gen)

    def state():
        return [t.clone() for t in (A.params, A.exp_avg, A.exp_avg_sq)]

    def segs():
        arr = (_lib.GradSegment * 2)()
        arr[0].begin, arr[0].end, arr[0].src, arr[0].stride, arr[0].n_src = seg_b[0], seg_b[1], src_b.data_ptr(), 512, chunks
        arr[1].begin, arr[1].end, arr[1].src, arr[1].stride, arr[1].n_src = seg_w[0], seg_w[1], src_w.data_ptr(), 6 * 1024, 3
        return arr

    def desc(ids, j, offset):
        x = _lib.ExtractPartsDesc()
        for f in ('g_rowptr', 'g_col', 'g_t_rowptr', 'g_t_col', 'node_part', 'part_slot', 'rowptr', 'col', 't_rowptr', 't_col',
                  'col_capacity', 'norm', 'feat', 'ld_feat', 'labels_all', 'labels'):
            setattr(x, f, getattr(P, f))
        x.ids, x.n, x.n_max, x.batch = ids.data_ptr(), ids.numel(), P.n_max, j
        x.n_feat, x.z0, x.ldz0 = 302, P.layer[0].Z, P.layer[0].ldz
        x.x0, x.ldx0, x.p, x.seed, x.offset, x.mask_ld = P.hsrc[0], P.ld_hsrc[0], 0.2, P.seed, offset, 2 * 302
        x.scratch = P.extract_scratch
        return x

    saved = state()
    outs = []
    for fused in (False, True):
        for t, s_ in zip((A.params, A.exp_avg, A.exp_avg_sq), saved):
            t.copy_(s_)
        g = grad.clone()
        loss = torch.zeros(1, device=DEV)
        hip.fill_i32_(it.batcher.rowptr, -7)
        it.batcher.norm.zero_()
        x = desc(b1.ids, 1, 1000)
        args = (A.params.data_ptr(), g.data_ptr(), A.exp_avg.data_ptr(), A.exp_avg_sq.data_ptr(), n, 0.01, 0.9, 0.999, 1e-8,
                5e-4, 7, segs(), 2, row_loss.data_ptr(), b0.n, b0.n, loss.data_ptr())
        if fused:
            _lib.check(L.gist_adam_segments_extract_f32(*args, ctypes.byref(x), hip._stream()), 'fused')
        else:
            _lib.check(L.gist_adam_segments_f32(*args, hip._stream()), 'adam')
            _lib.check(L.gist_extract_parts_batch(
                x.g_rowptr, x.g_col, x.g_t_rowptr, x.g_t_col, x.ids, x.n, x.n_max, x.node_part, x.part_slot, x.batch,
                x.rowptr, x.col, x.t_rowptr, x.t_col, x.col_capacity, x.norm, x.feat, x.ld_feat, x.n_feat, x.z0, x.ldz0,
                x.labels_all, x.labels, x.x0, x.ldx0, x.p, x.seed, x.offset, x.mask_ld, x.scratch, hip._stream()), 'extract')
        torch.cuda.synchronize()
        n1 = b1.ids.numel()
        nnz = int(it.batcher.rowptr[n1].item())
        outs.append(state() + [g, loss.clone(), it.batcher.rowptr[:n1 + 1].clone(), it.batcher.col[:nnz].clone(),
                               it.batcher.t_rowptr[:n1 + 1].clone(), it.batcher.norm[:n1].clone(), it.batcher.lab[:n1].clone(),
                               eng.Z[0][:n1, :302].clone(), eng.H[0][:n1, :302].clone()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert int(outs[0][7][-1].item()) > 0 and float(outs[0][1].abs().sum().item()) > 0
    eng.check_extract()


def test_step_flags_are_checked():
    """GIST_STEP_PREEXTRACTED with GIST_STEP_EXTRACT, or either prefetch flag on a forward-only call: refused."""
    import ctypes
    from gist_amd import _lib, hip
    L = _lib.load()
    eng, it = _engine(0.2, 2, 302, 512)
    b = next(iter(it))
    eng.train_step(b, 0.01, 5e-4)
    torch.cuda.synchronize()
    args = (ctypes.byref(eng.plan), b.ids.data_ptr(), b.n, 0, 0.01, 0.9, 0.999, 1e-8, 0.0, 1)
    for flags in (_lib.GIST_STEP_TRAIN | _lib.GIST_STEP_EXTRACT | _lib.GIST_STEP_PREEXTRACTED,
                  _lib.GIST_STEP_EXTRACT | _lib.GIST_STEP_EXTRACT_NEXT, _lib.GIST_STEP_PREEXTRACTED):
        assert L.gist_sage_step(*args, flags, hip._stream()) < 0
    assert L.gist_sage_step_extracts_next(ctypes.byref(eng.plan), b.n, _lib.GIST_STEP_EXTRACT) == 0      # (not a training step)
    torch.cuda.synchronize()
