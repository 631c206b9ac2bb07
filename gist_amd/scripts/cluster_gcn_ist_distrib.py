#!/usr/bin/env python3
"""GIST distributed training -- CLI and output contract of the reference's
cluster_gcn/cluster_gcn_ist_distrib.py (flags :520-564, five result lines :475-479), one
process per GPU over RCCL, running on the gist_amd HIP path.

Launch like the reference (script/reddit/run_ist_distrib.sh): one process per rank,

    for i in 0 1 2 3; do
      python -m gist_amd.scripts.cluster_gcn_ist_distrib --num_subnet 4 --rank $i --cuda-id $i \\
          --iter_per_site 100 --n-hidden 4096 --n-layers 2 --dropout 0.2 --use_layernorm True \\
          --lr 0.01 --n-epochs 40 --rnd-seed 0 --dataset reddit-synth &
    done; wait

`--use_layernorm` keeps the reference's `type=bool` quirk (:538): any non-empty string is True.
"""
import argparse
import random

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F


def build_parser():
    parser = argparse.ArgumentParser(description='GCN')
    from gist_amd.dgl_compat.data import register_data_args
    register_data_args(parser)
    parser.add_argument("--iter_per_site", type=int, default=5)
    parser.add_argument("--num_subnet", type=int, default=2, help="number of sub networks")
    parser.add_argument("--dropout", type=float, default=0.5, help="dropout probability")
    parser.add_argument("--lr", type=float, default=0.01, help="learning rate")
    parser.add_argument("--n-epochs", type=int, default=20, help="number of training epochs")
    parser.add_argument("--n-hidden", type=int, default=16, help="number of hidden gcn units")
    parser.add_argument("--n-layers", type=int, default=1, help="number of hidden gcn layers")
    parser.add_argument("--weight-decay", type=float, default=5e-4, help="Weight for L2 loss")
    parser.add_argument("--use_layernorm", type=bool, default=False)
    parser.add_argument('--dist-backend', type=str, default='nccl')
    parser.add_argument('--dist-url', type=str, default='tcp://127.0.0.1:9971')
    parser.add_argument('--rank', type=int, default=0)
    parser.add_argument('--cuda-id', type=int, default=0)
    parser.add_argument("--batch-size", type=int, default=20, help="batch size")
    parser.add_argument("--psize", type=int, default=1500, help="partition number")
    parser.add_argument("--test-batch-size", type=int, default=1000)
    parser.add_argument("--rnd-seed", type=int, default=3)
    parser.add_argument("--use-pp", action='store_true')
    parser.add_argument("--normalize", action='store_true')
    parser.add_argument("--save_results", action='store_true')
    parser.add_argument("--fig-dir", type=str, default='../report/example_pic/')
    parser.add_argument("--fig-name", type=str, default='name')
    parser.add_argument("--use-f1", action='store_true')
    return parser


def save_results(args, res, log=print):
    """:455-473 -- with --save_results the reference writes `{fig_name}_result.pckl` (total_time,
    trn_losses, val_accs, test_accs) under --fig-dir, which its sweep drivers read back
    (script/reddit/run_ist_sweep_reddit.py), INSTEAD of printing the five result lines.  The
    validation-accuracy PNG of :457-461 is not produced (matplotlib reporting is out of scope)."""
    import os
    import pickle
    os.makedirs(args.fig_dir, exist_ok=True)
    results = {'total_time': res['total_time'], 'trn_losses': res['trn_losses'],
               'val_accs': res['val_accs'], 'test_accs': res['test_accs']}
    path = os.path.join(args.fig_dir, args.fig_name + '_result.pckl')
    with open(path, 'wb') as f:
        pickle.dump(results, f)
    return path


def main(args=None, dataset=None, log=print, ultra_wide=False):
    from gist_amd import ist
    from gist_amd.dgl_compat.data import load_data
    from gist_amd.modules import GCN
    from gist_amd.sampler import EngineClusterIter
    from gist_amd.trainer import FullGraphEvaluator
    if args is None:
        args = build_parser().parse_args()
    assert (args.n_hidden % args.num_subnet) == 0
    if args.use_pp:
        raise NotImplementedError(
            'gist_amd: --use-pp cannot work with GCN / ISTSAGELayer in the reference either (the '
            'feature width doubles after in_feats was read, SURVEY.md appendix C.8); the layer-0 '
            'pre-aggregation is offered as ClusterIter(..., use_pp=True) + GraphSAGE-style layers')
    if args.use_f1:
        log('note: --use-f1 reports micro-F1, which equals argmax accuracy for single-label '
            'classification (cluster_gcn/utils.py:47-67)', flush=True)
    log('Setting seeds', flush=True)
    torch.manual_seed(args.rnd_seed)                        # :570-572, same seed on every rank
    np.random.seed(args.rnd_seed)
    random.seed(args.rnd_seed)
    assert args.cuda_id < torch.cuda.device_count()
    device = torch.device(f'cuda:{args.cuda_id}')
    torch.cuda.set_device(device)
    log(f'{args.rank} initializing process', flush=True)
    dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url,
                            rank=args.rank, world_size=args.num_subnet)
    data = dataset if dataset is not None else load_data(args)
    g = data.g
    if args.normalize:                 # StandardScaler fit on the train rows, on the device
        from gist_amd import hip
        feats = g.ndata['feat'].to(device).contiguous()
        fit = torch.nonzero(g.ndata['train_mask'].to(device)).flatten().to(torch.int32)
        hip.standard_scale_(feats, fit)
        g.ndata['feat'] = feats
    in_feats, n_classes = g.ndata['feat'].shape[1], data.num_classes
    train_nid = np.nonzero(g.ndata['train_mask'].numpy())[0].astype(np.int64)
    par_li = getattr(data, 'par_li', None)
    psize = len(par_li) if par_li is not None else args.psize
    it = EngineClusterIter(args.dataset, g, psize, args.batch_size, train_nid, par_li=par_li,
                           device=device)                    # :507-509 (one shuffle)
    base_init = None
    if args.rank == 0:                                       # :78-85 rank 0 builds the base model
        base = GCN(in_feats, args.n_hidden, n_classes, args.n_layers, F.relu, args.dropout,
                   args.use_layernorm, False, False, 1, True)
        base_init = [(l.linear.weight.detach(), l.linear.bias.detach()) for l in base.layers]
    model = ist.DistributedGNNWrapper(args, g, in_feats, n_classes, device, base_init=base_init,
                                      n_max=it.n_max, seed=args.rnd_seed)
    log(f'{args.rank}: start initial dispatch', flush=True)
    model.ini_sync_dispatch_model()                          # :595
    log(f'{args.rank}: finish initial dispatch', flush=True)
    it.bind(model.engine)
    evaluator = None
    if args.rank == 0:
        # ultra-wide (cluster_gcn_ist_ultra_wide.py:500-504 evaluates on the CPU because [N, 2H]
        # does not fit its GPU): the evaluator works block of rows by block of rows in HBM
        evaluator = FullGraphEvaluator(g, model.base_dims, args.use_layernorm, model.base, device,
                                       block_bytes=(1 << 30) if ultra_wide else (4 << 30))
    res = ist.train(model, args, it, evaluator=evaluator, log=log)
    if args.rank == 0:
        if args.save_results:
            log('results written to %s' % save_results(args, res, log=log), flush=True)
        else:
            ist.print_results(res, log=log)                  # :475-479
    dist.destroy_process_group()
    return res


if __name__ == '__main__':
    import os
    os.environ.setdefault('GIST_GC_FREEZE', '1')      # this process is the application: sampler.freeze_setup_objects
    main()
