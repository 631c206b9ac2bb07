"""TEST INFRASTRUCTURE ONLY -- records golden vectors from the REFERENCE's own code.

Run in the build container (needs /root/reference; nothing here runs on the GPU
box):

    python oracle/gen_golden.py            # writes tests/golden/*.npz

It puts oracle/dgl_stub (a torch-CPU stand-in for the absent `dgl` package) and
/root/reference/cluster_gcn on sys.path, imports the reference modules
UNCHANGED and records inputs + outputs as small .npz fixtures:

  G1_layer_*.npz   ISTSAGELayer fwd/bwd                 (cluster_gcn/modules.py:191-243)
  G2_model_*.npz   GCN logits / CE / grads / Adam steps (modules.py:245-314,
                                                         cluster_gcn_ist_distrib.py:405-417)
  G3_sampler.npz   ClusterIter batch order + batch 0 subgraph (sampler.py, partition_utils.py)
  G4_ist_*.npz     create_partition, dispatch/sync under gloo (cluster_gcn_ist_distrib.py:51-367)
  G5_graphconv.npz gcn/gcn.py forward on the stub's GraphConv  (parity UNPINNED: DGL recalled)
  G5_train_*.npz   gcn/train.py main() end to end (BASELINE config 1, Cora plumbing): per-epoch
                   losses, accuracies, initial and final parameters on a small citation graph
  G6_e2e_*.npz     whole training runs of cluster_gcn.py main() and
                   cluster_gcn_ist_distrib.py train() on a toy graph

Only data is written -- no reference source text.
"""
import argparse
import os
import random
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden')


def _setup_paths():
    for p in (os.path.join(HERE, 'dgl_stub'), os.path.join(REF, 'cluster_gcn')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.setdefault('MPLBACKEND', 'Agg')


# ---------------------------------------------------------------------------
# toy graphs
# ---------------------------------------------------------------------------

def toy_edges(n, seed, avg_deg=4, hub=None):
    """Directed multigraph with a zero-in-degree node (node n-1 unless tiny),
    self loops, duplicate edges and optionally a hub with many in-edges."""
    rs = np.random.RandomState(seed)
    m = n * avg_deg
    src = rs.randint(0, n, m)
    dst = rs.randint(0, max(n - 1, 1), m)          # node n-1 never a destination
    loops = np.arange(0, n - 1, 2)                  # self loops on even nodes
    src = np.concatenate([src, loops, src[:5]])     # 5 duplicate edges
    dst = np.concatenate([dst, loops, dst[:5]])
    if hub is not None:
        hs = rs.randint(0, n, hub)
        src = np.concatenate([src, hs])
        dst = np.concatenate([dst, np.full(hub, 1)])
    return src.astype(np.int64), dst.astype(np.int64)


def make_graph(n, src, dst):
    import dgl
    return dgl.DGLGraph((src, dst), num_nodes=n)


def params_of(model):
    out = {}
    for k, layer in enumerate(model.layers):
        out['W%d' % k] = layer.linear.weight.detach().numpy().copy()
        out['b%d' % k] = layer.linear.bias.detach().numpy().copy()
    return out


# ---------------------------------------------------------------------------
def gen_G1():
    import torch
    import torch.nn.functional as F
    from modules import ISTSAGELayer
    for n, fin, fout, hub in [(7, 5, 4, None), (64, 12, 9, None), (257, 33, 20, 200)]:
        src, dst = toy_edges(n, seed=n, hub=hub)
        g = make_graph(n, src, dst)
        rowptr, col = g.in_csr()
        for ln in (True, False):
            for act in (True, False):
                torch.manual_seed(100 + n)
                layer = ISTSAGELayer(fin, fout, 0.0, ln, activation=F.relu if act else None)
                h = torch.randn(n, fin, requires_grad=True)
                d_out = torch.randn(n, fout)
                out = layer(g, h)
                (out * d_out).sum().backward()
                np.savez_compressed(
                    os.path.join(OUT, 'G1_layer_n%d_ln%d_act%d.npz' % (n, ln, act)),
                    src=src, dst=dst, rowptr=rowptr, col=col, n=n,
                    h=h.detach().numpy(), W=layer.linear.weight.detach().numpy(),
                    b=layer.linear.bias.detach().numpy(), d_out=d_out.numpy(),
                    out=out.detach().numpy(), dh=h.grad.numpy(),
                    dW=layer.linear.weight.grad.numpy(), db=layer.linear.bias.grad.numpy(),
                    use_lynorm=ln, relu=act)


def gen_G2():
    import torch
    import torch.nn.functional as F
    from modules import GCN
    n, fin, ncls = 96, 10, 5
    src, dst = toy_edges(n, seed=5, avg_deg=6, hub=40)
    g = make_graph(n, src, dst)
    rowptr, col = g.in_csr()
    cases = [('full', 1, 1), ('full', 2, 1), ('sub', 1, 2), ('sub', 2, 2), ('sub', 2, 4)]
    for kind, L, S in cases:
        for wd in (0.0, 5e-4):
            for ln in ((True, False) if wd == 0.0 else (True,)):
                torch.manual_seed(7)
                H = 16
                if kind == 'full':
                    model = GCN(fin, H, ncls, L, F.relu, 0.0, ln, False, False, 1, True)
                else:
                    model = GCN(fin, H, ncls, L, F.relu, 0.0, ln, False, True, S, True)
                init = params_of(model)
                feat = torch.randn(n, fin)
                labels = torch.randint(0, ncls, (n,))
                g.ndata['feat'] = feat
                opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=wd)
                loss_f = torch.nn.CrossEntropyLoss()
                rec = dict(src=src, dst=dst, rowptr=rowptr, col=col, n=n, feat=feat.numpy(),
                           labels=labels.numpy(), L=L, S=S, H=H, wd=wd, use_layernorm=ln,
                           n_classes=ncls, kind=kind)
                rec.update({'init_' + k: v for k, v in init.items()})
                for step in range(3):
                    opt.zero_grad()
                    logits = model(g)
                    loss = loss_f(logits, labels)
                    loss.backward()
                    if step == 0:
                        rec['logits'] = logits.detach().numpy().copy()
                        rec['loss'] = np.float32(loss.item())
                        for k, layer in enumerate(model.layers):
                            rec['dW%d' % k] = layer.linear.weight.grad.numpy().copy()
                            rec['db%d' % k] = layer.linear.bias.grad.numpy().copy()
                    rec['loss_step%d' % step] = np.float32(loss.item())
                    opt.step()
                    if step in (0, 2):
                        rec.update({'step%d_%s' % (step + 1, k): v
                                    for k, v in params_of(model).items()})
                np.savez_compressed(
                    os.path.join(OUT, 'G2_model_%s_L%d_S%d_wd%d_ln%d.npz'
                                 % (kind, L, S, int(wd > 0), ln)), **rec)


def toy_dataset(n=240, n_parts=12, fin=8, ncls=4, seed=11):
    """A small graph + masks + ragged partition list of the TRAIN-induced graph."""
    import torch
    rs = np.random.RandomState(seed)
    src, dst = toy_edges(n, seed=seed, avg_deg=5, hub=30)
    # symmetrise like Reddit / Amazon (AmazonDataset.py:94-97)
    src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    g = make_graph(n, src, dst)
    role = rs.rand(n)
    train_mask = role < 0.7
    val_mask = (role >= 0.7) & (role < 0.85)
    test_mask = role >= 0.85
    g.ndata['feat'] = torch.from_numpy(rs.randn(n, fin).astype(np.float32))
    g.ndata['label'] = torch.from_numpy(rs.randint(0, ncls, n).astype(np.int64))
    g.ndata['train_mask'] = torch.from_numpy(train_mask)
    g.ndata['val_mask'] = torch.from_numpy(val_mask)
    g.ndata['test_mask'] = torch.from_numpy(test_mask)
    n_train = int(train_mask.sum())
    perm = rs.permutation(n_train)
    cuts = np.sort(rs.choice(np.arange(1, n_train), n_parts - 1, replace=False))
    parts = [p.astype(np.int64) for p in np.split(perm, cuts)]   # ragged, ids in train graph
    return g, (src, dst), parts, ncls


def save_partition_cache(dirname, dn, psize, parts):
    """Write `../data/{dn}_{psize}.npy` in the reference's cache format
    (sampler.py:44-51): object array of 1-D int64 arrays."""
    arr = np.empty(len(parts), dtype=object)
    for i, p in enumerate(parts):
        arr[i] = p
    os.makedirs(dirname, exist_ok=True)
    np.save(os.path.join(dirname, '%s_%d.npy' % (dn, psize)), arr, allow_pickle=True)


def gen_G3():
    from sampler import ClusterIter
    g, (src, dst), parts, ncls = toy_dataset()
    train_nid = np.nonzero(g.ndata['train_mask'].numpy())[0].astype(np.int64)
    psize, bs = len(parts), 3
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        work = os.path.join(td, 'work')
        os.makedirs(work)
        save_partition_cache(os.path.join(td, 'data'), 'toy', psize, parts)
        os.chdir(work)
        try:
            random.seed(3)
            it = ClusterIter('toy', g, psize, bs, train_nid, use_pp=False)
            rec = dict(src=src, dst=dst, n=g.number_of_nodes(), train_nid=train_nid,
                       psize=psize, batch_size=bs, seed=3,
                       feat=g.ndata['feat'].numpy(), label=g.ndata['label'].numpy())
            for i, p in enumerate(parts):
                rec['part%d' % i] = p
            tr_rowptr, tr_col = it.g.in_csr()
            rec['train_rowptr'], rec['train_col'] = tr_rowptr, tr_col
            for ep in range(2):
                for j, cluster in enumerate(it):
                    rec['ep%d_b%d_nid' % (ep, j)] = cluster.ndata['_ID'].numpy()
                    if ep == 0 and j == 0:
                        rp, cl = cluster.in_csr()
                        rec['b0_rowptr'], rec['b0_col'] = rp, cl
                        rec['b0_feat'] = cluster.ndata['feat'].numpy()
                        rec['b0_label'] = cluster.ndata['label'].numpy()
            rec['n_batches'] = len(it)
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, 'G3_sampler.npz'), **rec)


# ---------------------------------------------------------------------------
def _ist_args(S, H, L, rank, **kw):
    a = argparse.Namespace(num_subnet=S, n_hidden=H, n_layers=L, rank=rank, dropout=0.0,
                           use_layernorm=True, lr=0.01, weight_decay=0.0,
                           iter_per_site=3, n_epochs=6, use_f1=False, save_results=False,
                           fig_dir=None, fig_name='toy', dataset='toy', psize=12,
                           batch_size=3, use_pp=False)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _ist_worker(rank, S, H, L, fin, ncls, port, seed, outdir):
    _setup_paths()
    import torch
    import torch.distributed as dist
    import cluster_gcn_ist_distrib as ref
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port,
                            rank=rank, world_size=S)
    args = _ist_args(S, H, L, rank)
    w = ref.DistributedGNNWrapper(args, None, fin, ncls, torch.device('cpu'))
    rec = {}
    if rank == 0:
        rec.update({'base0_' + k: v for k, v in params_of(w.base_model).items()})
    w.ini_sync_dispatch_model()
    rec.update({'sub_ini_' + k: v for k, v in params_of(w.sub_model).items()})
    for l, layer_part in enumerate(w.current_partition):
        for s, (idx, full) in enumerate(layer_part):
            rec['part0_l%d_s%d' % (l, s)] = idx.numpy()
    # deterministic, rank-dependent "training" perturbation of every sub parameter
    with torch.no_grad():
        gen = torch.Generator().manual_seed(1000 + rank)
        for p in w.sub_model.parameters():
            p.add_(torch.randn(p.shape, generator=gen) * 0.1)
    rec.update({'sub_pert_' + k: v for k, v in params_of(w.sub_model).items()})
    w.sync_model()
    if rank == 0:
        rec.update({'base1_' + k: v for k, v in params_of(w.base_model).items()})
    w.dispatch_model()
    rec.update({'sub_disp_' + k: v for k, v in params_of(w.sub_model).items()})
    for l, layer_part in enumerate(w.current_partition):
        for s, (idx, full) in enumerate(layer_part):
            rec['part1_l%d_s%d' % (l, s)] = idx.numpy()
    w.sync_model()    # no training in between: base must be bit-identical (property)
    if rank == 0:
        rec.update({'base2_' + k: v for k, v in params_of(w.base_model).items()})
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), **rec)
    dist.barrier()
    dist.destroy_process_group()


def gen_G4():
    import torch.multiprocessing as mp
    from cluster_gcn_ist_distrib import create_partition
    rec = {}
    for seed in (0, 3):
        for S, H in ((2, 16), (4, 16), (8, 64)):
            random.seed(seed)
            part = create_partition(S, H)
            for s, (idx, full) in enumerate(part):
                rec['cp_seed%d_S%d_H%d_s%d' % (seed, S, H, s)] = idx.numpy()
                assert np.array_equal(full.numpy(), np.concatenate([idx.numpy(), idx.numpy() + H]))
    np.savez_compressed(os.path.join(OUT, 'G4_create_partition.npz'), **rec)

    fin, ncls = 6, 5
    port = 29640
    for S, H, L in ((2, 16, 2), (4, 16, 2), (2, 8, 1), (4, 16, 3)):
        with tempfile.TemporaryDirectory() as td:
            mp.spawn(_ist_worker, args=(S, H, L, fin, ncls, port, 3, td), nprocs=S, join=True)
            port += 1
            rec = dict(S=S, H=H, L=L, fin=fin, ncls=ncls, seed=3)
            for r in range(S):
                d = np.load(os.path.join(td, 'rank%d.npz' % r))
                for k in d.files:
                    if k.startswith('base') or k.startswith('part'):
                        if r == 0 or k.startswith('part'):
                            if k.startswith('part') and r > 0:
                                assert np.array_equal(rec[k], d[k])   # same on all ranks
                            rec[k] = d[k]
                    else:
                        rec['r%d_%s' % (r, k)] = d[k]
        np.savez_compressed(os.path.join(OUT, 'G4_ist_S%d_H%d_L%d.npz' % (S, H, L)), **rec)


# ---------------------------------------------------------------------------
def gen_G5():
    sys.path.insert(0, os.path.join(REF, 'gcn'))
    import torch
    import torch.nn.functional as F
    import importlib
    gcn_mod = importlib.import_module('gcn')
    n = 10
    src, dst = toy_edges(n, seed=2, avg_deg=3)
    src = np.concatenate([src, np.arange(n)])
    dst = np.concatenate([dst, np.arange(n)])      # self loops (gcn/train.py:66-68)
    g = make_graph(n, src, dst)
    torch.manual_seed(2)
    model = gcn_mod.GCN(g, 7, 6, 3, 1, F.relu, 0.0, True)
    model.eval()
    x = torch.randn(n, 7)
    out = model(x)
    rowptr, col = g.in_csr()
    rec = dict(src=src, dst=dst, n=n, rowptr=rowptr, col=col, x=x.numpy(),
               out=out.detach().numpy(),
               out_deg=np.bincount(src, minlength=n).astype(np.int64))
    for k, layer in enumerate(model.layers):
        rec['W%d' % k] = layer.weight.detach().numpy()
        rec['b%d' % k] = layer.bias.detach().numpy()
    np.savez_compressed(os.path.join(OUT, 'G5_graphconv.npz'), **rec)
    sys.path.remove(os.path.join(REF, 'gcn'))
    sys.modules.pop('gcn', None)


def gen_G5_train():
    """gcn/train.py main() UNCHANGED on a small Cora-like citation graph (CPU), recorded
    through three spies: the GCN constructor (initial + final parameters), the loss module
    (per-epoch training loss) and evaluate() (val / test accuracy per epoch)."""
    import contextlib
    import importlib.util
    import io
    import torch
    if REPO not in sys.path:
        sys.path.append(REPO)
    from gist_amd.datasets import citation_synth          # pure numpy data generator
    gdir = os.path.join(REF, 'gcn')
    sys.path.insert(0, gdir)
    sys.modules.pop('gcn', None)
    spec = importlib.util.spec_from_file_location('ref_gcn_train', os.path.join(gdir, 'train.py'))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    cases = [('ln1_L1', dict(use_layernorm='True', n_layers=1, n_hidden=16, lr_scheduler=False,
                             weight_decay=5e-4, n_epochs=12, lr=0.01)),
             ('ln0_L2', dict(use_layernorm='False', n_layers=2, n_hidden=24, lr_scheduler=True,
                             weight_decay=0.0, n_epochs=8, lr=0.02))]
    for tag, cfg in cases:
        data = citation_synth('cora-mini', n=300, n_undirected=700, n_feats=60, n_classes=7,
                              train_per_class=5, n_val=60, n_test=100, nnz_per_row=9, seed=12)
        state = {}
        real_gcn, real_eval, real_ce = ref.GCN, ref.evaluate, torch.nn.CrossEntropyLoss
        losses, accs = [], []

        def spy_gcn(*a, **k):
            m = real_gcn(*a, **k)
            state['model'] = m
            state['init'] = [(l.weight.detach().numpy().copy(), l.bias.detach().numpy().copy())
                             for l in m.layers]
            return m

        class SpyCE(real_ce):
            def forward(self, inp, tgt):
                out = super().forward(inp, tgt)
                losses.append(float(out.detach()))
                return out

        def spy_eval(*a, **k):
            acc = real_eval(*a, **k)
            accs.append(acc)
            return acc

        ref.GCN, ref.evaluate, ref.load_data = spy_gcn, spy_eval, (lambda args: data)
        torch.nn.CrossEntropyLoss = SpyCE
        args = argparse.Namespace(dataset='cora', dropout=0.0, gpu=-1, self_loop='True', **cfg)
        torch.manual_seed(21)
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                ref.main(args)
        finally:
            ref.GCN, ref.evaluate = real_gcn, real_eval
            torch.nn.CrossEntropyLoss = real_ce
        tail = [l for l in buf.getvalue().strip().split('\n') if 'Accuracy' in l]
        m = state['model']
        n_ep = cfg['n_epochs']
        assert len(losses) == n_ep and len(accs) == 2 * n_ep + 1
        rec = dict(src=data.src, dst=data.dst, n=data.features.shape[0], feat=data.features,
                   label=data.labels, train_mask=data.train_mask, val_mask=data.val_mask,
                   test_mask=data.test_mask, n_classes=data.num_labels,
                   losses=np.array(losses, np.float32),
                   val_accs=np.array(accs[0:2 * n_ep:2]), test_accs=np.array(accs[1:2 * n_ep:2]),
                   final_test=accs[-1], n_edges_with_loops=m.g.number_of_edges(),
                   tail_keys=np.array([t.split(':')[0] for t in tail]),
                   tail_vals=np.array([float(t.split(':')[1]) for t in tail]),
                   use_layernorm=cfg['use_layernorm'] == 'True', n_layers=cfg['n_layers'],
                   n_hidden=cfg['n_hidden'], lr=cfg['lr'], weight_decay=cfg['weight_decay'],
                   n_epochs=n_ep, lr_scheduler=cfg['lr_scheduler'], seed=21)
        for k, (W, b) in enumerate(state['init']):
            rec['init_W%d' % k], rec['init_b%d' % k] = W, b
        for k, l in enumerate(m.layers):
            rec['final_W%d' % k] = l.weight.detach().numpy()
            rec['final_b%d' % k] = l.bias.detach().numpy()
        np.savez_compressed(os.path.join(OUT, 'G5_train_%s.npz' % tag), **rec)
    sys.path.remove(gdir)
    sys.modules.pop('gcn', None)


# ---------------------------------------------------------------------------
def _patch_data(mod, g, ncls):
    from collections import namedtuple
    DataType = namedtuple('Dataset', ['num_classes', 'g'])
    mod.load_data = lambda args: DataType(g=g, num_classes=ncls)


def gen_G6_single():
    """cluster_gcn/cluster_gcn.py main() end to end on the toy graph (CPU, --gpu -1)."""
    import torch
    import cluster_gcn as ref
    g, (src, dst), parts, ncls = toy_dataset()
    _patch_data(ref, g, ncls)
    snaps, init = [], {}
    real_eval = ref.evaluate
    real_gcn = ref.GCN

    def spy_gcn(*a, **k):
        m = real_gcn(*a, **k)
        init.update(params_of(m))
        return m

    def spy_eval(model, gg, labels, mask, method='acc'):
        snaps.append(params_of(model))
        return real_eval(model, gg, labels, mask, method)
    ref.evaluate = spy_eval
    ref.GCN = spy_gcn
    psize, bs = len(parts), 3
    args = argparse.Namespace(
        dataset='toy', dropout=0.0, gpu=-1, lr=0.01, n_epochs=3, batch_size=bs, psize=psize,
        test_batch_size=1000, n_hidden=16, n_layers=2, rnd_seed=3, use_pp=False,
        normalize=False, weight_decay=0.0, model_type='sage', fig_name='toy',
        use_layernorm=True, use_f1=False, eval_cpu=False, fig_dir=None)
    cwd = os.getcwd()
    import io
    import contextlib
    buf = io.StringIO()
    with tempfile.TemporaryDirectory() as td:
        work = os.path.join(td, 'work')
        os.makedirs(work)
        args.fig_dir = os.path.join(td, 'fig')
        save_partition_cache(os.path.join(td, 'data'), 'toy', psize, parts)
        os.chdir(work)
        try:
            with contextlib.redirect_stdout(buf):
                ref.main(args)
        finally:
            os.chdir(cwd)
            ref.evaluate = real_eval
            ref.GCN = real_gcn
    lines = [l for l in buf.getvalue().strip().split('\n')]
    tail = lines[-5:]
    rec = dict(src=src, dst=dst, n=g.number_of_nodes(), psize=psize, batch_size=bs,
               feat=g.ndata['feat'].numpy(), label=g.ndata['label'].numpy(),
               train_mask=g.ndata['train_mask'].numpy(), val_mask=g.ndata['val_mask'].numpy(),
               test_mask=g.ndata['test_mask'].numpy(), n_classes=ncls,
               n_hidden=16, n_layers=2, lr=0.01, n_epochs=3, rnd_seed=3,
               val_accs=np.array([float(l.split()[-1]) for l in lines if l.startswith('Val acc')]),
               last_val=float(tail[1].split(':')[1]), best_val=float(tail[2].split(':')[1]),
               last_test=float(tail[3].split(':')[1]), best_test=float(tail[4].split(':')[1]),
               tail_keys=np.array([t.split(':')[0] for t in tail]))
    for i, p in enumerate(parts):
        rec['part%d' % i] = p
    rec.update({'init_' + k: v for k, v in init.items()})
    # evaluate() is called twice per epoch (val, test): keep the first of each pair
    for e in range(args.n_epochs):
        rec.update({'ep%d_%s' % (e, k): v for k, v in snaps[2 * e].items()})
    np.savez_compressed(os.path.join(OUT, 'G6_e2e_single.npz'), **rec)


def _e2e_ist_worker(rank, S, port, outdir, datadir):
    _setup_paths()
    import io
    import contextlib
    import torch
    import torch.distributed as dist
    import cluster_gcn_ist_distrib as ref
    from sampler import ClusterIter
    g, (src, dst), parts, ncls = toy_dataset()
    seed = 3
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port,
                            rank=rank, world_size=S)
    psize, bs = len(parts), 3
    args = _ist_args(S, 16, 2, rank, psize=psize, batch_size=bs, n_epochs=3 * S,
                     iter_per_site=3, fig_dir=os.path.join(outdir, 'fig%d' % rank))
    os.chdir(os.path.join(datadir, 'work'))
    train_mask = g.ndata['train_mask']
    train_nid = np.nonzero(train_mask.numpy())[0].astype(np.int64)
    it = ClusterIter('toy', g, psize, bs, train_nid, use_pp=False)
    events, snaps, losses = [], [], []
    w = ref.DistributedGNNWrapper(args, g, g.ndata['feat'].shape[1], ncls, torch.device('cpu'))
    rec = {}
    if rank == 0:
        rec.update({'base_init_' + k: v for k, v in params_of(w.base_model).items()})
    w.ini_sync_dispatch_model()
    rec.update({'sub_init_' + k: v for k, v in params_of(w.sub_model).items()})
    real_disp, real_sync, real_eval = w.dispatch_model, w.sync_model, ref.evaluate

    def spy_disp():
        events.append('dispatch')
        real_disp()

    def spy_sync():
        events.append('sync')
        real_sync()
        if rank == 0:
            snaps.append(params_of(w.base_model))

    def spy_eval(model, gg, labels, mask, method='acc'):
        events.append('eval')
        return real_eval(model, gg, labels, mask, method)
    w.dispatch_model, w.sync_model, ref.evaluate = spy_disp, spy_sync, spy_eval
    real_ce = torch.nn.CrossEntropyLoss

    class SpyCE(real_ce):
        def forward(self, a, b):
            l = super().forward(a, b)
            losses.append(float(l))
            events.append('step')
            return l
    torch.nn.CrossEntropyLoss = SpyCE
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            ref.train(w, args, g, it, g.ndata['label'], train_mask, g.ndata['val_mask'],
                      g.ndata['test_mask'], train_nid, torch.device('cpu'))
    finally:
        torch.nn.CrossEntropyLoss = real_ce
    rec['losses'] = np.array(losses, np.float32)
    rec['events'] = np.array(events)
    if rank == 0:
        tail = buf.getvalue().strip().split('\n')[-5:]
        rec['tail_keys'] = np.array([t.split(':')[0] for t in tail])
        rec['tail_vals'] = np.array([float(t.split(':')[1]) for t in tail])
        for i, s in enumerate(snaps):
            rec.update({'sync%d_%s' % (i, k): v for k, v in s.items()})
        rec['n_syncs'] = len(snaps)
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), **rec)
    dist.barrier()
    dist.destroy_process_group()


def gen_G6_ist():
    import torch.multiprocessing as mp
    g, (src, dst), parts, ncls = toy_dataset()
    port = 29700
    for S in (2, 4):
        with tempfile.TemporaryDirectory() as td:
            os.makedirs(os.path.join(td, 'work'))
            save_partition_cache(os.path.join(td, 'data'), 'toy', len(parts), parts)
            mp.spawn(_e2e_ist_worker, args=(S, port, td, td), nprocs=S, join=True)
            port += 1
            rec = dict(S=S, src=src, dst=dst, n=g.number_of_nodes(), psize=len(parts),
                       batch_size=3, feat=g.ndata['feat'].numpy(),
                       label=g.ndata['label'].numpy(),
                       train_mask=g.ndata['train_mask'].numpy(),
                       val_mask=g.ndata['val_mask'].numpy(),
                       test_mask=g.ndata['test_mask'].numpy(), n_classes=ncls,
                       n_hidden=16, n_layers=2, lr=0.01, n_epochs=3 * S, iter_per_site=3,
                       rnd_seed=3)
            for i, p in enumerate(parts):
                rec['part%d' % i] = p
            for r in range(S):
                d = np.load(os.path.join(td, 'rank%d.npz' % r))
                for k in d.files:
                    rec['r%d_%s' % (r, k)] = d[k]
        np.savez_compressed(os.path.join(OUT, 'G6_e2e_ist_S%d.npz' % S), **rec)


def main():
    assert os.path.isdir(REF), 'golden generation needs /root/reference'
    _setup_paths()
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ['G1', 'G2', 'G3', 'G4', 'G5', 'G5t', 'G6s', 'G6i']
    fns = dict(G1=gen_G1, G2=gen_G2, G3=gen_G3, G4=gen_G4, G5=gen_G5, G5t=gen_G5_train,
               G6s=gen_G6_single, G6i=gen_G6_ist)
    for w in which:
        print('generating', w, flush=True)
        fns[w]()
    print('done ->', OUT)


if __name__ == '__main__':
    main()
