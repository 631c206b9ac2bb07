#!/usr/bin/env python3
"""Small full-graph GCN training -- CLI, loop and output contract of the reference's
gcn/train.py (flags :126-148, data handling :35-72, loop :93-116, the three result lines
:120-123), BASELINE config 1 ("Cora plumbing").  The reference runs this on the CPU through
DGL; here every layer runs on the HIP kernels (GraphConv = one scaled CSR SpMM + one fp32-MFMA
GEMM, whole-tensor layer norm, CE and Adam kernels) with torch.autograd as the tape.

    python -m gist_amd.scripts.gcn_train --dataset cora-synth --n-epochs 400 --lr 0.01

`--dataset cora|citeseer|pubmed` need the real files (none offline: that is an error, never a
silent substitute); `cora-synth` is the seeded Cora-like stand-in of SURVEY.md section 8d.
"""
import argparse
import time

import numpy as np
import torch
import torch.nn.functional as F


def build_parser():
    parser = argparse.ArgumentParser(description='GCN')
    parser.add_argument("--dataset", type=str, default="cora")
    parser.add_argument("--data-root", type=str, default=None)
    parser.add_argument("--dropout", type=float, default=0.5, help="dropout probability")
    parser.add_argument("--gpu", type=int, default=-1, help="gpu")
    parser.add_argument("--lr", type=float, default=.001, help="learning rate")
    parser.add_argument("--n-epochs", type=int, default=400, help="number of training epochs")
    parser.add_argument("--n-hidden", type=int, default=16, help="number of hidden gcn units")
    parser.add_argument("--n-layers", type=int, default=1, help="number of hidden gcn layers")
    parser.add_argument("--weight-decay", type=float, default=5e-4, help="Weight for L2 loss")
    parser.add_argument("--self_loop", type=str, default='True', help="graph self-loop (default=True)")
    parser.add_argument("--lr_scheduler", action='store_true', default=False, help="Use LR scheduler")
    parser.add_argument("--use_layernorm", type=str, default='True',
                        help="Whether use layernorm (default=False)")
    return parser


def evaluate(model, features, labels, mask):
    """gcn/train.py:14-22."""
    model.eval()
    with torch.no_grad():
        logits = model(features)
        logits = logits[mask]
        labels = labels[mask]
        _, indices = torch.max(logits, dim=1)
        correct = torch.sum(indices == labels)
        return correct.item() * 1.0 / len(labels)


def main(args, data=None, log=print, init_params=None):
    import networkx as nx
    from gist_amd.dgl_compat import DGLGraph
    from gist_amd.dgl_compat.data import load_data
    from gist_amd.gcn import GCN
    from gist_amd.nn import CrossEntropyLoss
    from gist_amd.optim import Adam
    assert args.self_loop in ['True', 'False'], ["Only True or False for self_loop, get ", args.self_loop]
    assert args.use_layernorm in ['True', 'False'], ["Only True or False for use_layernorm, get ",
                                                     args.use_layernorm]
    self_loop = (args.self_loop == 'True')
    use_layernorm = (args.use_layernorm == 'True')
    if data is None:
        if args.dataset in {'cora', 'citeseer', 'pubmed', 'cora-synth'}:
            data = load_data(args)
        else:
            raise NotImplementedError(f'{args.dataset} is not a valid dataset')
    features = torch.FloatTensor(data.features)
    labels = torch.LongTensor(data.labels)
    train_mask = torch.BoolTensor(np.asarray(data.train_mask, bool))     # the reference's ByteTensor
    val_mask = torch.BoolTensor(np.asarray(data.val_mask, bool))         # masks are deprecated
    test_mask = torch.BoolTensor(np.asarray(data.test_mask, bool))       # in current torch
    in_feats = features.shape[1]
    n_classes = data.num_labels
    n_edges = data.graph.number_of_edges()
    log("""----Data statistics------'
      #Edges %d
      #Classes %d
      #Train samples %d
      #Val samples %d
      #Test samples %d""" %
        (n_edges, n_classes, train_mask.sum().item(), val_mask.sum().item(), test_mask.sum().item()))
    if not torch.cuda.is_available():
        raise RuntimeError('gist_amd: gcn_train runs on the HIP kernels and needs a GPU '
                           '(there is no CPU path)')
    device = torch.device('cuda', max(args.gpu, 0))
    features, labels = features.to(device), labels.to(device)
    train_mask, val_mask, test_mask = train_mask.to(device), val_mask.to(device), test_mask.to(device)

    g = data.graph.copy()
    if self_loop:                                             # :65-68
        g.remove_edges_from(nx.selfloop_edges(g))
        g.add_edges_from(zip(g.nodes(), g.nodes()))
    g = DGLGraph(g).to(device)
    n_edges = g.number_of_edges()

    model = GCN(g, in_feats, args.n_hidden, n_classes, args.n_layers, F.relu, args.dropout,
                use_layernorm)
    if init_params is not None:                               # tests: the reference's own init
        with torch.no_grad():
            for layer, (W, b) in zip(model.layers, init_params):
                layer.weight.copy_(torch.as_tensor(W))
                layer.bias.copy_(torch.as_tensor(b))
    model = model.to(device)
    loss_fcn = CrossEntropyLoss()
    optimizer = Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)

    record, dur, losses = [], [], []
    for epoch in range(args.n_epochs):
        if args.lr_scheduler:                                 # :95-101
            if epoch == int(0.5 * args.n_epochs) or epoch == int(0.75 * args.n_epochs):
                for pg in optimizer.param_groups:
                    pg['lr'] = pg['lr'] / 10
        model.train()
        if epoch >= 3:
            torch.cuda.synchronize(device)
            t0 = time.time()
        optimizer.zero_grad()
        logits = model(features)
        loss = loss_fcn(logits[train_mask], labels[train_mask])
        loss.backward()
        optimizer.step()
        if epoch >= 3:
            torch.cuda.synchronize(device)
            dur.append(time.time() - t0)
        losses.append(loss.detach())
        acc_val = evaluate(model, features, labels, val_mask)
        acc_test = evaluate(model, features, labels, test_mask)
        record.append([acc_val, acc_test])

    all_test_acc = [v[1] for v in record]
    all_val_acc = [v[0] for v in record]
    acc = evaluate(model, features, labels, test_mask)
    log(f"Final Test Accuracy: {acc:.4f}")
    log(f"Best Val Accuracy: {max(all_val_acc):.4f}")
    log(f"Best Test Accuracy: {max(all_test_acc):.4f}")
    return dict(model=model, losses=[float(l.item()) for l in losses], record=record,
                final_test=acc, n_edges=n_edges, epoch_time=float(np.mean(dur)) if dur else None)


if __name__ == '__main__':
    main(build_parser().parse_args())
