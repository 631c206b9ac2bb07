#!/usr/bin/env python3
"""Single-GPU Cluster-GCN training -- CLI and output contract of the reference's
cluster_gcn/cluster_gcn.py (flags :145-180, five result lines :132-136), running on the
gist_amd HIP path.

    python -m gist_amd.scripts.cluster_gcn --dataset reddit-synth --lr 0.01 --n-epochs 80 \\
        --batch-size 20 --n-hidden 256 --n-layers 4 --dropout 0.2 --use-layernorm --rnd-seed 0

Datasets are the seeded synthetic stand-ins of gist_amd.datasets (no real data offline);
their block structure supplies the partition list in place of METIS.
"""
import argparse
import random

import numpy as np
import torch
import torch.nn.functional as F


def build_parser():
    parser = argparse.ArgumentParser(description='GCN')
    from gist_amd.dgl_compat.data import register_data_args
    register_data_args(parser)
    parser.add_argument("--dropout", type=float, default=0.2, help="dropout probability")
    parser.add_argument("--gpu", type=int, default=0, help="gpu")
    parser.add_argument("--lr", type=float, default=3e-2, help="learning rate")
    parser.add_argument("--n-epochs", type=int, default=40, help="number of training epochs")
    parser.add_argument("--batch-size", type=int, default=20, help="batch size")
    parser.add_argument("--psize", type=int, default=1500, help="partition number")
    parser.add_argument("--test-batch-size", type=int, default=1000, help="test batch size")
    parser.add_argument("--n-hidden", type=int, default=128, help="number of hidden gcn units")
    parser.add_argument("--n-layers", type=int, default=1, help="number of hidden gcn layers")
    parser.add_argument("--rnd-seed", type=int, default=3)
    parser.add_argument("--use-pp", action='store_true', help="whether to use precomputation")
    parser.add_argument("--normalize", action='store_true', help="whether to use normalized feature")
    parser.add_argument("--weight-decay", type=float, default=0, help="Weight for L2 loss")
    parser.add_argument("--model-type", type=str, default='sage')
    parser.add_argument("--fig-dir", type=str, default='../report/example_pic/')
    parser.add_argument("--fig-name", type=str, default='name')
    parser.add_argument("--use-layernorm", action='store_true')
    parser.add_argument("--use-f1", action='store_true')
    parser.add_argument("--eval-cpu", action='store_true')
    # (not a flag of the reference) module: the reference's own loop body, statement for statement, on gist_amd.modules.GCN /
    # nn.CrossEntropyLoss / optim.Adam / sampler.ClusterIter (gist_amd/module_engine.py); engine: one gist_sage_step per iteration
    parser.add_argument("--host-path", choices=['engine', 'module'], default='engine')
    return parser


def main(args, dataset=None, log=print):
    from gist_amd.dgl_compat.data import load_data
    from gist_amd.modules import GCN
    from gist_amd.trainer import ClusterGCNTrainer
    torch.manual_seed(args.rnd_seed)                       # cluster_gcn.py:20-22
    np.random.seed(args.rnd_seed)
    random.seed(args.rnd_seed)
    if args.gpu < 0 or args.eval_cpu:
        raise SystemExit('gist_amd runs on the GPU only (no CPU fallback): --gpu >= 0, no --eval-cpu')
    if args.model_type != 'sage':
        raise NotImplementedError(f'{args.model_type} is not a supported model type')
    data = dataset if dataset is not None else load_data(args)
    g = data.g
    device = torch.device('cuda', args.gpu)
    torch.cuda.set_device(device)
    if args.normalize:                 # StandardScaler fit on the train rows, on the device
        from gist_amd import hip
        feats = g.ndata['feat'].to(device).contiguous()
        fit = torch.nonzero(g.ndata['train_mask'].to(device)).flatten().to(torch.int32)
        hip.standard_scale_(feats, fit)
        g.ndata['feat'] = feats
    in_feats = g.ndata['feat'].shape[1]
    n_classes = data.num_classes
    par_li = getattr(data, 'par_li', None)
    psize = len(par_li) if par_li is not None else args.psize
    log('labels shape:', g.ndata['label'].shape)
    log("features shape, ", g.ndata['feat'].shape)
    # construction order as in the reference: ClusterIter (one shuffle) before the model
    model_holder = {}

    def make_model():
        m = GCN(in_feats, args.n_hidden, n_classes, args.n_layers, F.relu, args.dropout,
                args.use_layernorm, False, False, 1, True)          # :66-69
        model_holder['m'] = m
        return m
    if getattr(args, 'host_path', 'engine') == 'module':
        return main_module_path(args, data, g, device, in_feats, n_classes, par_li, psize, log)
    trainer = ClusterGCNTrainer(args.dataset, g, par_li, psize, args.batch_size, args.n_hidden,
                                args.n_layers, n_classes, args.dropout, args.use_layernorm,
                                args.lr, args.weight_decay, device, seed=args.rnd_seed)
    trainer.engine.arena.adopt_module(make_model())        # same-seed init as the reference
    val_accs, test_accs = [], []
    for epoch in range(args.n_epochs):                     # :89-127
        log(f'Running epoch {epoch} / {args.n_epochs}', flush=True)
        trainer.timed_epoch()
        val_accs.append(trainer.evaluate('val_mask'))
        test_accs.append(trainer.evaluate('test_mask'))
        log(f'Val acc {val_accs[-1]}', flush=True)
    log(f'Training Time: {trainer.total_time:.4f}', flush=True)    # :132-136
    log(f'Last Val: {val_accs[-1]:.4f}', flush=True)
    log(f'Best Val: {max(val_accs):.4f}', flush=True)
    log(f'Last Test: {test_accs[-1]:.4f}', flush=True)
    log(f'Best Test: {max(test_accs):.4f}', flush=True)
    return dict(total_time=trainer.total_time, val_accs=val_accs, test_accs=test_accs,
                model=model_holder['m'])


def main_module_path(args, data, g, device, in_feats, n_classes, par_li, psize, log):
    """cluster_gcn/cluster_gcn.py:24-136 on the drop-in classes: ClusterIter, GCN, CrossEntropyLoss, Adam, evaluate."""
    import time
    from gist_amd.modules import GCN
    from gist_amd.nn import CrossEntropyLoss
    from gist_amd.optim import Adam
    from gist_amd.sampler import ClusterIter
    from gist_amd.utils import evaluate
    train_nid = np.nonzero(g.ndata['train_mask'].numpy())[0].astype(np.int64)
    cluster_iterator = ClusterIter(args.dataset, g, psize, args.batch_size, train_nid, use_pp=args.use_pp,
                                   par_li=par_li, device=device)                                      # :44-46
    g = g.to(device)
    labels, val_mask, test_mask = g.ndata['label'], g.ndata['val_mask'], g.ndata['test_mask']
    model = GCN(in_feats, args.n_hidden, n_classes, args.n_layers, F.relu, args.dropout,
                args.use_layernorm, False, False, 1, True)                                             # :66-69
    model.cuda()
    model.set_dropout_seed(args.rnd_seed)
    loss_f = CrossEntropyLoss()                                                                        # :76
    optimizer = Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)                   # :77-80
    total_time, val_accs, test_accs = 0., [], []
    for epoch in range(args.n_epochs):                                                                 # :89-127
        log(f'Running epoch {epoch} / {args.n_epochs}', flush=True)
        torch.cuda.synchronize(device)
        start_time = time.time()
        for j, cluster in enumerate(cluster_iterator):
            cluster = cluster.to(torch.cuda.current_device())
            model.train()
            pred = model(cluster)
            batch_labels = cluster.ndata['label']
            batch_train_mask = cluster.ndata['train_mask']
            loss = loss_f(pred[batch_train_mask], batch_labels[batch_train_mask])
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
        torch.cuda.synchronize(device)
        total_time += time.time() - start_time
        val_accs.append(evaluate(model, g, labels, val_mask, 'f1' if args.use_f1 else 'acc'))
        test_accs.append(evaluate(model, g, labels, test_mask, 'f1' if args.use_f1 else 'acc'))
        log(f'Val acc {val_accs[-1]}', flush=True)
    log(f'Training Time: {total_time:.4f}', flush=True)                                                # :132-136
    log(f'Last Val: {val_accs[-1]:.4f}', flush=True)
    log(f'Best Val: {max(val_accs):.4f}', flush=True)
    log(f'Last Test: {test_accs[-1]:.4f}', flush=True)
    log(f'Best Test: {max(test_accs):.4f}', flush=True)
    return dict(total_time=total_time, val_accs=val_accs, test_accs=test_accs, model=model)


if __name__ == '__main__':
    import os
    os.environ.setdefault('GIST_GC_FREEZE', '1')      # this process is the application: sampler.freeze_setup_objects
    a = build_parser().parse_args()
    print(a)
    main(a)
