#!/usr/bin/env python3
"""Small full-graph GCN training (BASELINE config 1, "Cora plumbing"): the command line and the printed
lines of the reference's gcn/train.py -- flags and defaults :126-148, the data-statistics block :52-62,
the three result lines :120-123 -- over gist_amd.gcn.FullGraphTrainer, which runs every layer on the HIP
kernels (GraphConv = one scaled CSR SpMM + one fp32-MFMA GEMM, whole-tensor layer norm, CE and Adam
kernels; torch.autograd is the tape).

    python -m gist_amd.scripts.gcn_train --dataset cora-synth --n-epochs 400 --lr 0.01

`--dataset cora|citeseer|pubmed` need the real files (none offline: an error, never a silent substitute);
`cora-synth` is the seeded Cora-like stand-in of SURVEY.md section 8d.
"""
import argparse

import numpy as np

CITATION_SETS = ('cora', 'citeseer', 'pubmed', 'cora-synth')


def build_parser():
    ap = argparse.ArgumentParser(description='GCN')
    add = ap.add_argument
    add("--dataset", type=str, default="cora")
    add("--data-root", type=str, default=None)
    add("--dropout", type=float, default=0.5, help="dropout probability")
    add("--gpu", type=int, default=-1, help="gpu")
    add("--lr", type=float, default=.001, help="learning rate")
    add("--n-epochs", type=int, default=400, help="number of training epochs")
    add("--n-hidden", type=int, default=16, help="number of hidden gcn units")
    add("--n-layers", type=int, default=1, help="number of hidden gcn layers")
    add("--weight-decay", type=float, default=5e-4, help="Weight for L2 loss")
    add("--self_loop", type=str, default='True', help="graph self-loop (default=True)")
    add("--lr_scheduler", action='store_true', default=False, help="Use LR scheduler")
    add("--use_layernorm", type=str, default='True', help="Whether use layernorm (default=False)")
    return ap


def _flag(args, name):
    """The reference's string booleans (gcn/train.py:28-33): exactly 'True' or 'False'."""
    v = getattr(args, name)
    if v not in ('True', 'False'):
        raise AssertionError('Only True or False for %s, get %s' % (name, v))
    return v == 'True'


def main(args, data=None, log=print, init_params=None):
    import torch
    from gist_amd.gcn import FullGraphTrainer
    self_loop, use_layernorm = _flag(args, 'self_loop'), _flag(args, 'use_layernorm')
    if data is None:
        if args.dataset not in CITATION_SETS:
            raise NotImplementedError('%s is not a valid dataset' % args.dataset)
        from gist_amd.dgl_compat.data import load_data
        data = load_data(args)
    t = FullGraphTrainer(data, args.n_hidden, args.n_layers, args.dropout, use_layernorm, args.lr,
                         args.weight_decay, self_loop=self_loop,
                         device=torch.device('cuda', max(args.gpu, 0)) if torch.cuda.is_available() else None,
                         init_params=init_params)
    sizes = t.mask_sizes()
    log("----Data statistics------'\n"
        "      #Edges %d\n      #Classes %d\n      #Train samples %d\n      #Val samples %d\n"
        "      #Test samples %d" % (t.n_edges_raw, t.n_classes, sizes['train'], sizes['val'], sizes['test']))
    t.fit(args.n_epochs, lr_scheduler=args.lr_scheduler)
    final = t.accuracy('test')
    log("Final Test Accuracy: %.4f" % final)
    log("Best Val Accuracy: %.4f" % max(v for v, _ in t.history))
    log("Best Test Accuracy: %.4f" % max(s for _, s in t.history))
    return dict(model=t.model, losses=[float(l.item()) for l in t.losses], record=[list(r) for r in t.history],
                final_test=final, n_edges=t.n_edges,
                epoch_time=float(np.mean(t.step_seconds)) if t.step_seconds else None)


if __name__ == '__main__':
    main(build_parser().parse_args())
