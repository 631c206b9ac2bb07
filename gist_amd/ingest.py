"""Dataset ingestion for real data on disk: the two on-disk formats the
reference's `utils.load_data` ends up reading, turned into a gist_amd Graph with the ndata the
training scripts use (feat, label, train/val/test masks).  Host-side numpy/json only -- no
tensorflow, networkx or DGL needed.

  * GraphSAGE directory format, as read by cluster_gcn/AmazonDataset.py:25-118 (Amazon2M):
      {name}-feats.npy, {name}-G.json (node-link), {name}-id_map.json, {name}-class_map.json
    -> edges whose endpoints are in id_map, symmetrised and de-duplicated (`adj += adj.T` then
       from_scipy, :94-97,108); val/test from the node attributes, train = the rest (:60-66);
       labels = argmax of the one-hot / multi-hot class map (:73-86); features standardised
       with the TRAIN rows' mean / std (sklearn StandardScaler semantics, :88-92).
  * DGL's RedditDataset files (dgl.data.RedditDataset, reached through utils.load_data ->
    dgl.data.load_data for `reddit` / `reddit-self-loop`):
      reddit_data.npz {feature, label, node_types: 1 train / 2 val / 3 test},
      reddit_graph.npz or reddit_self_loop_graph.npz (scipy.sparse.save_npz COO).
There is no such data in the build container; tests/test_ingest.py round-trips tiny fixtures
written in exactly these formats.
"""
import json
import os
from collections import namedtuple

import numpy as np
import torch

from .graph import Graph

Dataset = namedtuple('Dataset', ['num_classes', 'g', 'par_li', 'name'])


def standard_scale(feats, train_ids):
    """sklearn.preprocessing.StandardScaler fit on the train rows: population std, a zero
    std scales by 1."""
    x = np.asarray(feats, np.float64)
    mu = x[train_ids].mean(0)
    sd = x[train_ids].std(0)
    sd[sd == 0.0] = 1.0
    return ((x - mu) / sd).astype(np.float32)


def _attach(g, feats, labels, train, val, test):
    g.ndata['feat'] = torch.from_numpy(np.ascontiguousarray(feats, np.float32))
    g.ndata['label'] = torch.from_numpy(np.ascontiguousarray(labels, np.int64))
    g.ndata['train_mask'] = torch.from_numpy(train)
    g.ndata['val_mask'] = torch.from_numpy(val)
    g.ndata['test_mask'] = torch.from_numpy(test)
    return g


def load_graphsage_dir(root, name, num_classes=None):
    """AmazonDataset.process (cluster_gcn/AmazonDataset.py:25-118) without tensorflow/networkx."""
    pre = os.path.join(root, name)
    feats = np.load(pre + '-feats.npy').astype(np.float32)
    with open(pre + '-G.json') as f:
        G = json.load(f)
    with open(pre + '-id_map.json') as f:
        id_map = json.load(f)
    with open(pre + '-class_map.json') as f:
        class_map = json.load(f)
    is_digit = next(iter(id_map)).isdigit()
    key = (lambda k: int(k)) if is_digit else (lambda k: k)
    id_map = {key(k): int(v) for k, v in id_map.items()}
    n = len(id_map)
    nodes = G['nodes']
    node_ids = [nd['id'] for nd in nodes]
    # node-link "source"/"target" are node ids (networkx >= 2) -- fall back to positions in the
    # node list (networkx 1.x files) when an endpoint is not an id
    id_set = set(node_ids)
    links = G.get('links', G.get('edges', []))
    src, dst = [], []
    for e in links:
        a, b = e['source'], e['target']
        if a not in id_set or b not in id_set:
            a, b = node_ids[a], node_ids[b]
        if a in id_map and b in id_map:
            src.append(id_map[a])
            dst.append(id_map[b])
    src, dst = np.asarray(src, np.int64), np.asarray(dst, np.int64)
    val_nodes = np.asarray([id_map[nd['id']] for nd in nodes if nd.get('val')], np.int64)
    test_nodes = np.asarray([id_map[nd['id']] for nd in nodes if nd.get('test')], np.int64)
    val = np.zeros(n, bool)
    test = np.zeros(n, bool)
    val[val_nodes] = True
    test[test_nodes] = True
    train = ~(val | test)
    first = next(iter(class_map.values()))
    if isinstance(first, list):
        labels = np.zeros(n, np.int64)
        for k, v in class_map.items():
            labels[id_map[key(k)]] = int(np.argmax(np.asarray(v)))
        n_cls = len(first)
    else:
        labels = np.zeros(n, np.int64)
        for k, v in class_map.items():
            labels[id_map[key(k)]] = int(v)
        n_cls = len(set(int(v) for v in class_map.values()))
    train_ids = np.asarray([id_map[nd['id']] for nd in nodes if not nd.get('val') and not nd.get('test')],
                           np.int64)
    feats = standard_scale(feats, train_ids)
    # adj += adj.T, then one edge per nonzero (both directions, duplicates merged)
    u = np.concatenate([src, dst])
    v = np.concatenate([dst, src])
    code = np.unique(u * n + v)
    g = Graph.from_edges(code // n, code % n, n)
    return Dataset(num_classes=int(num_classes or n_cls), g=_attach(g, feats, labels, train, val, test),
                   par_li=None, name=name)


def load_dgl_reddit(root, self_loop=True):
    """dgl.data.RedditDataset's raw files (DGL 0.5: reddit_data.npz + reddit[_self_loop]_graph.npz)."""
    import scipy.sparse as sp
    data = np.load(os.path.join(root, 'reddit_data.npz'))
    adj = sp.load_npz(os.path.join(root, 'reddit_self_loop_graph.npz' if self_loop else 'reddit_graph.npz')).tocoo()
    n = adj.shape[0]
    g = Graph.from_edges(adj.row.astype(np.int64), adj.col.astype(np.int64), n)   # row -> col
    types = data['node_types']
    labels = data['label'].astype(np.int64)
    return Dataset(num_classes=int(labels.max()) + 1,
                   g=_attach(g, data['feature'].astype(np.float32), labels, types == 1, types == 2,
                             types == 3),
                   par_li=None, name='reddit-self-loop' if self_loop else 'reddit')


def try_load(name, data_root):
    """Real data if `data_root` holds it in one of the two formats, else None."""
    if not data_root:
        return None
    if name in ('amazon2m', 'amazon2M') and os.path.exists(os.path.join(data_root, 'amazon2M-G.json')):
        return load_graphsage_dir(data_root, 'amazon2M', num_classes=47)      # AmazonDataset.py:157-160
    if name.startswith('reddit') and os.path.exists(os.path.join(data_root, 'reddit_data.npz')):
        return load_dgl_reddit(data_root, self_loop='self-loop' in name or 'self_loop' in name)
    if os.path.exists(os.path.join(data_root, name + '-G.json')):
        return load_graphsage_dir(data_root, name)
    return None
