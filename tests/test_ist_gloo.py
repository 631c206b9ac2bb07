"""CPU, multi-process: the IST dispatch/sync choreography of gist_amd.ist under a real
`gloo` process group (world size 2 and 4), against golden vectors recorded from the
reference's DistributedGNNWrapper under gloo (tests/golden/G4_ist_*.npz).

The HIP block kernels cannot run here, so the wrapper is given a TEST DOUBLE for the
three block movers (torch indexing, defined in this file only -- the product has no such
path); everything else is the product code: partition sampling, block index plan, the
packed all-gather over torch.distributed, the replicated-base bookkeeping.
"""
import os
import random

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


class TorchBlocks(object):
    """Test double for gist_amd.ist.HipBlocks (same contract as gist_block_gather/scatter_f32
    and gist_mean_rows_f32)."""

    def gather(self, src, row_idx, col_idx, dst):
        s = src
        if row_idx is not None:
            s = s[row_idx.long()]
        if col_idx is not None:
            s = s[:, col_idx.long()]
        dst.copy_(s)

    def scatter(self, src, row_idx, col_idx, dst):
        r = row_idx.long() if row_idx is not None else torch.arange(src.shape[0])
        c = col_idx.long() if col_idx is not None else torch.arange(src.shape[1])
        dst[r[:, None], c[None, :]] = src

    def mean_rows(self, src_flat, stride, n_src, n, out):
        acc = torch.zeros(n)
        for s in range(n_src):
            acc = acc + src_flat[s * stride:s * stride + n]
        out.copy_(acc / n_src)


def _params(d, prefix, n):
    return [(d['%sW%d' % (prefix, k)], d['%sb%d' % (prefix, k)]) for k in range(n)]


def _worker(rank, S, name, port, q):
    import argparse
    from gist_amd import ist
    try:
        d = np.load(os.path.join(GOLD, name))
        H, L = int(d['H']), int(d['L'])
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank,
                                world_size=S)
        random.seed(int(d['seed']))
        args = argparse.Namespace(num_subnet=S, n_hidden=H, n_layers=L, rank=rank, dropout=0.0,
                                  use_layernorm=True)
        base_init = _params(d, 'base0_', L + 1) if rank == 0 else None
        w = ist.DistributedGNNWrapper(args, None, int(d['fin']), int(d['ncls']),
                                      torch.device('cpu'), base_init=base_init,
                                      blocks=TorchBlocks())
        errs = []

        def same(a, b, what, tol=0.0):
            a = a.detach().numpy()
            if tol == 0.0:
                if not np.array_equal(a, b):
                    errs.append(what)
            elif np.abs(a - b).max() > tol:
                errs.append(what)

        w.ini_sync_dispatch_model()
        for l in range(L):
            for s in range(S):
                same(w.current_partition[l][s][0], d['part0_l%d_s%d' % (l, s)], 'part0')
        # every rank now holds a replica of rank 0's base model
        for k in range(L + 1):
            same(w.base.W[k], d['base0_W%d' % k], 'replica W%d' % k)
            same(w.sub.W[k], d['r%d_sub_ini_W%d' % (rank, k)], 'sub_ini W%d' % k)
            same(w.sub.b[k], d['r%d_sub_ini_b%d' % (rank, k)], 'sub_ini b%d' % k)
        # "training": load the reference's perturbed sub-model of this rank
        w.sub.load(_params(d, 'r%d_sub_pert_' % rank, L + 1))
        w.sync_model()
        for k in range(L + 1):
            same(w.base.W[k], d['base1_W%d' % k], 'base1 W%d' % k)
            same(w.base.b[k], d['base1_b%d' % k], 'base1 b%d' % k, tol=0.0 if k < L else 1e-6)
        w.dispatch_model()
        for l in range(L):
            for s in range(S):
                same(w.current_partition[l][s][0], d['part1_l%d_s%d' % (l, s)], 'part1')
        for k in range(L + 1):
            same(w.sub.W[k], d['r%d_sub_disp_W%d' % (rank, k)], 'sub_disp W%d' % k)
            if k < L:
                same(w.sub.b[k], d['r%d_sub_disp_b%d' % (rank, k)], 'sub_disp b%d' % k)
        before = [W.clone() for W in w.base.W]
        w.sync_model()                       # dispatch -> sync without training: identity
        for k in range(L + 1):
            same(w.base.W[k], before[k].numpy(), 'identity W%d' % k)
            same(w.base.W[k], d['base2_W%d' % k], 'base2 W%d' % k)
            same(w.base.b[k], d['base2_b%d' % k], 'base2 b%d' % k, tol=0.0 if k < L else 1e-6)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, errs))
    except Exception as e:          # surface the failure in the parent
        import traceback
        q.put((rank, ['EXC ' + repr(e) + traceback.format_exc()]))


@pytest.mark.parametrize('name,S,port', [('G4_ist_S2_H16_L2.npz', 2, 29811),
                                         ('G4_ist_S4_H16_L2.npz', 4, 29812),
                                         ('G4_ist_S2_H8_L1.npz', 2, 29813),
                                         ('G4_ist_S4_H16_L3.npz', 4, 29814)])
def test_ist_dispatch_sync_gloo(name, S, port):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, S, name, port, q)) for r in range(S)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(S)]
    for p in procs:
        p.join(timeout=60)
    for rank, errs in sorted(res):
        assert errs == [], 'rank %d: %s' % (rank, errs)


def test_create_partition_matches_reference():
    from gist_amd.ist import create_partition
    d = np.load(os.path.join(GOLD, 'G4_create_partition.npz'))
    for seed in (0, 3):
        for S, H in ((2, 16), (4, 16), (8, 64)):
            random.seed(seed)
            part = create_partition(S, H)
            for s, (idx, full) in enumerate(part):
                assert np.array_equal(idx.numpy(), d['cp_seed%d_S%d_H%d_s%d' % (seed, S, H, s)])
                assert np.array_equal(full.numpy(), np.concatenate([idx.numpy(), idx.numpy() + H]))
                assert idx.dtype == torch.int64
