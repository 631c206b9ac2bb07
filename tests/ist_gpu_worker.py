"""One rank of a multi-process GIST run on ONE GPU (started by tests/test_ist_multiproc_gpu.py).

    python tests/ist_gpu_worker.py <mode> <rank> <S> <port> <golden.npz> <out.json>

Every rank is its own process on cuda:0 (like the reference's launcher with `--cuda-id 0`,
script/reddit/run_ist_distrib.sh:16-18) with the PRODUCT block movers (gist_amd.ist.HipBlocks:
gist_block_gather/scatter_f32, gist_mean_rows_f32) and the product wrapper / train loop; the
one collective is host-staged over gloo (tests/host_staged_comm.py) because RCCL refuses two
ranks on one device.  mode g4: dispatch / sync choreography against the reference's
DistributedGNNWrapper run (tests/golden/G4_ist_*.npz).  mode g6: the whole train() loop against
the reference's run (tests/golden/G6_e2e_ist_S*.npz).  Writes {"rank", "errors": [...]}.
"""
import argparse
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.host_staged_comm import HostStagedComm  # noqa: E402
TOL = 1e-4


def _params(d, prefix, n):
    return [(d['%sW%d' % (prefix, k)], d['%sb%d' % (prefix, k)]) for k in range(n)]


def run_g4(rank, S, d, errs):
    import torch
    from gist_amd import ist
    dev = torch.device('cuda', 0)
    H, L = int(d['H']), int(d['L'])
    random.seed(int(d['seed']))
    args = argparse.Namespace(num_subnet=S, n_hidden=H, n_layers=L, rank=rank, dropout=0.0,
                              use_layernorm=True)
    base_init = _params(d, 'base0_', L + 1) if rank == 0 else None
    w = ist.DistributedGNNWrapper(args, None, int(d['fin']), int(d['ncls']), dev,
                                  base_init=base_init, comm=HostStagedComm())
    assert isinstance(w.blocks, ist.HipBlocks)

    def same(a, b, what, tol=0.0):
        a = a.detach().cpu().numpy()
        if tol == 0.0:
            if not np.array_equal(a, b):
                errs.append(what)
        elif np.abs(a - b).max() > tol:
            errs.append(what)

    w.ini_sync_dispatch_model()
    for l in range(L):
        for s in range(S):
            same(w.current_partition[l][s][0], d['part0_l%d_s%d' % (l, s)], 'part0')
    for k in range(L + 1):
        same(w.base.W[k], d['base0_W%d' % k], 'replica W%d' % k)
        same(w.sub.W[k], d['r%d_sub_ini_W%d' % (rank, k)], 'sub_ini W%d' % k)
        same(w.sub.b[k], d['r%d_sub_ini_b%d' % (rank, k)], 'sub_ini b%d' % k)
    w.sub.load(_params(d, 'r%d_sub_pert_' % rank, L + 1))      # "training"
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
        same(w.base.W[k], before[k].cpu().numpy(), 'identity W%d' % k)
        same(w.base.W[k], d['base2_W%d' % k], 'base2 W%d' % k)
        same(w.base.b[k], d['base2_b%d' % k], 'base2 b%d' % k, tol=0.0 if k < L else 1e-6)


def run_g6(rank, S, d, errs):
    import torch
    from gist_amd import ist
    from gist_amd.graph import Graph
    from gist_amd.sampler import EngineClusterIter
    from gist_amd.trainer import FullGraphEvaluator
    dev = torch.device('cuda', 0)
    g = Graph.from_edges(d['src'], d['dst'], int(d['n']))
    g.ndata['feat'] = torch.from_numpy(d['feat'])
    g.ndata['label'] = torch.from_numpy(d['label'])
    for m in ('train_mask', 'val_mask', 'test_mask'):
        g.ndata[m] = torch.from_numpy(d[m])
    L, H = int(d['n_layers']), int(d['n_hidden'])
    fin, ncls = d['feat'].shape[1], int(d['n_classes'])
    random.seed(int(d['rnd_seed']))
    train_nid = np.nonzero(d['train_mask'])[0].astype(np.int64)
    it = EngineClusterIter('toy', g, int(d['psize']), int(d['batch_size']), train_nid,
                           par_li=[d['part%d' % i] for i in range(int(d['psize']))], device=dev)
    args = argparse.Namespace(num_subnet=S, n_hidden=H, n_layers=L, rank=rank, dropout=0.0,
                              use_layernorm=True, lr=float(d['lr']), weight_decay=0.0,
                              iter_per_site=int(d['iter_per_site']), n_epochs=int(d['n_epochs']))
    model = ist.DistributedGNNWrapper(
        args, None, fin, ncls, dev,
        base_init=_params(d, 'r0_base_init_', L + 1) if rank == 0 else None,
        comm=HostStagedComm(), n_max=it.n_max)
    assert isinstance(model.blocks, ist.HipBlocks)
    model.ini_sync_dispatch_model()
    for k in range(L + 1):
        if not np.array_equal(model.sub.W[k].cpu().numpy(), d['r%d_sub_init_W%d' % (rank, k)]):
            errs.append('sub_init W%d' % k)
        if not np.array_equal(model.sub.b[k].cpu().numpy(), d['r%d_sub_init_b%d' % (rank, k)]):
            errs.append('sub_init b%d' % k)
    it.bind(model.engine)
    evaluator = FullGraphEvaluator(g, model.base_dims, True, model.base, dev) if rank == 0 else None
    snaps = []
    orig_apply = model.sync_apply

    def spy_apply():
        orig_apply()
        snaps.append(model.base.export())
    model.sync_apply = spy_apply
    res = ist.train(model, args, it, evaluator=evaluator, log=lambda *a: None)
    got = np.array([float(x.item()) for x in res['losses'][0]])
    if got.shape != d['r%d_losses' % rank].shape or np.abs(got - d['r%d_losses' % rank]).max() >= TOL:
        errs.append('losses of rank %d' % rank)
    # every rank's base replica after every sync == rank 0's base model in the reference
    if len(snaps) != int(d['r0_n_syncs']):
        errs.append('number of syncs %d' % len(snaps))
    for i, snap in enumerate(snaps):
        for k, (W, b) in enumerate(snap):
            if np.abs(W - d['r0_sync%d_W%d' % (i, k)]).max() >= TOL:
                errs.append('sync%d W%d' % (i, k))
            if np.abs(b - d['r0_sync%d_b%d' % (i, k)]).max() >= TOL:
                errs.append('sync%d b%d' % (i, k))
    if rank == 0:
        gold = [str(e) for e in d['r0_events']]
        dedup = [e for i, e in enumerate(gold) if not (e == 'eval' and gold[i - 1] == 'eval')]
        if res['events'] != dedup:
            errs.append('event schedule')
        tail = dict(zip([str(k) for k in d['r0_tail_keys']], d['r0_tail_vals']))
        for name, val in (('Last Val', res['val_accs'][-1]), ('Best Val', max(res['val_accs'])),
                          ('Last Test', res['test_accs'][-1]), ('Best Test', max(res['test_accs']))):
            if abs(val - tail[name]) >= 1e-4:
                errs.append(name)


def main():
    mode, rank, S, port, gold, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), \
        int(sys.argv[4]), sys.argv[5], sys.argv[6]
    errs = []
    try:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank,
                                world_size=S)
        d = np.load(gold)
        (run_g4 if mode == 'g4' else run_g6)(rank, S, d, errs)
        # the product library really is what ran
        maps = open('/proc/self/maps').read()
        if 'libgist_hip.so' not in maps:
            errs.append('libgist_hip.so not loaded')
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        errs.append('EXC ' + repr(e) + traceback.format_exc())
    with open(out, 'w') as f:
        json.dump({'rank': rank, 'errors': errs}, f)
    sys.exit(1 if errs else 0)


if __name__ == '__main__':
    main()
