"""Seeded synthetic stand-ins for the datasets GIST trains on (none are available
offline; SURVEY.md section 8d fixes the statistics).  Pure host-side construction --
not on the hot path.

A dataset is a stochastic block model whose blocks play the role of METIS parts:
dense inside a block, sparse between blocks, symmetric, one self loop per node
(reddit-self-loop), a small fraction of hub nodes with 10x degree (power-law skew).

  reddit-synth   N_train=153431, F=602, C=41, 1500 blocks; a 20-block batch has
                 ~2046 rows and mean in-batch in-degree ~64
  amazon-synth   N_train=1709997 (scalable), F=100, C=47, 15000 blocks, batch of 10
                 blocks ~1140 rows, mean in-degree ~16
  toy            small graph for tests
"""
from collections import namedtuple

import numpy as np
import torch

from .graph import Graph

Dataset = namedtuple('Dataset', ['num_classes', 'g', 'par_li', 'name'])


def sbm_edges(n, n_blocks, intra_deg, inter_deg, hub_frac, hub_mult, seed, inter_locality=0.0, neigh_parts=8):
    """Directed edge list (symmetrised + self loops) of a block model on n nodes.
    inter_locality: share of a node's inter-block edges that go to one of its block's `neigh_parts` NEIGHBOUR
    blocks (the blocks at ring distance 1 .. neigh_parts / 2 on either side: a symmetric relation) instead of
    to a uniformly random node -- partitions of real graphs cut few block pairs heavily rather than all pairs
    thinly; 0 = the uniform model (the worst case for the evaluation's gathers)."""
    rs = np.random.RandomState(seed)
    base, extra = divmod(n, n_blocks)
    sizes = np.full(n_blocks, base, np.int64)
    sizes[:extra] += 1
    starts = np.zeros(n_blocks + 1, np.int64)
    np.cumsum(sizes, out=starts[1:])
    block_of = np.repeat(np.arange(n_blocks), sizes)
    mult = np.ones(n, np.int64)
    n_hub = int(n * hub_frac)
    if n_hub:
        mult[rs.choice(n, n_hub, replace=False)] = hub_mult
    nodes = np.arange(n, dtype=np.int64)
    src_i = np.repeat(nodes, intra_deg * mult)
    b = block_of[src_i]
    dst_i = starts[b] + (rs.random_sample(src_i.shape[0]) * sizes[b]).astype(np.int64)
    src_o = np.repeat(nodes, inter_deg * mult)
    dst_o = rs.randint(0, n, src_o.shape[0]).astype(np.int64)
    if inter_locality > 0.0 and n_blocks > neigh_parts:
        near = rs.random_sample(src_o.shape[0]) < inter_locality
        half = max(neigh_parts // 2, 1)
        step = rs.randint(1, half + 1, src_o.shape[0]) * np.where(rs.random_sample(src_o.shape[0]) < 0.5, -1, 1)
        nb = (block_of[src_o] + step) % n_blocks
        dst_near = starts[nb] + (rs.random_sample(src_o.shape[0]) * sizes[nb]).astype(np.int64)
        dst_o = np.where(near, dst_near, dst_o)
    src = np.concatenate([src_i, dst_i, src_o, dst_o, nodes])
    dst = np.concatenate([dst_i, src_i, dst_o, src_o, nodes])
    blocks = [np.arange(starts[k], starts[k + 1], dtype=np.int64) for k in range(n_blocks)]
    return src, dst, blocks


def make_block_dataset(name, n, n_blocks, n_feats, n_classes, intra_deg, inter_deg, seed,
                       hub_frac=0.01, hub_mult=10, train_frac=1.0, inter_locality=0.0, neigh_parts=8):
    """Graph with ndata feat/label/{train,val,test}_mask on the host + partition list of
    the TRAIN-induced graph (ids relative to the train graph, like METIS on
    ClusterIter.g, sampler.py:34,50)."""
    src, dst, blocks = sbm_edges(n, n_blocks, intra_deg, inter_deg, hub_frac, hub_mult, seed, inter_locality,
                                 neigh_parts)
    g = Graph.from_edges(src, dst, n)
    # the node ids are ordered by block: boundaries for the evaluator's block-diagonal split
    # (trainer.FullGraphEvaluator node_blocks; blocks of at most 128 nodes only)
    if max(len(b) for b in blocks) <= 128:
        g.node_blocks = np.concatenate([[0], np.cumsum([len(b) for b in blocks])]).astype(np.int64)
    gen = torch.Generator().manual_seed(seed)
    g.ndata['feat'] = torch.randn(n, n_feats, generator=gen)            # post-StandardScaler stats
    g.ndata['label'] = torch.randint(0, n_classes, (n,), generator=gen)
    rs = np.random.RandomState(seed + 1)
    if train_frac >= 1.0:
        role = np.zeros(n)
    else:
        role = rs.rand(n)
    train = role < train_frac
    val = (role >= train_frac) & (role < train_frac + (1 - train_frac) * 0.3)
    test = role >= train_frac + (1 - train_frac) * 0.3
    g.ndata['train_mask'] = torch.from_numpy(train)
    g.ndata['val_mask'] = torch.from_numpy(val)
    g.ndata['test_mask'] = torch.from_numpy(test)
    # partition list over the train-induced graph: block k -> its train nodes, relabelled
    new_id = np.cumsum(train) - 1
    par_li = [new_id[blk[train[blk]]].astype(np.int64) for blk in blocks]
    par_li = [p for p in par_li if p.shape[0] > 0]
    return Dataset(num_classes=n_classes, g=g, par_li=par_li, name=name)


def save_arrays(ds, directory):
    """Dump a Dataset as plain .npy files (one per array; a directory in /dev/shm is a memcpy): how
    bench.py hands ONE synthetic graph per node to all of its rank processes instead of letting every
    rank regenerate it (the reference makes every rank load the full dataset)."""
    import os
    g = ds.g
    np.save(os.path.join(directory, 'meta.npy'),
            np.array([ds.num_classes, g.number_of_nodes(), len(ds.par_li)], np.int64))
    for k in ('rowptr', 'col', 't_rowptr', 't_col'):
        np.save(os.path.join(directory, k + '.npy'), getattr(g, k).numpy())
    for k, v in g.ndata.items():
        np.save(os.path.join(directory, 'ndata_' + k + '.npy'), v.numpy())
    if getattr(g, 'node_blocks', None) is not None:
        np.save(os.path.join(directory, 'node_blocks.npy'), g.node_blocks)
    np.save(os.path.join(directory, 'par_sizes.npy'), np.array([len(p) for p in ds.par_li], np.int64))
    np.save(os.path.join(directory, 'par_cat.npy'),
            np.concatenate(ds.par_li) if ds.par_li else np.zeros(0, np.int64))
    with open(os.path.join(directory, 'name.txt'), 'w') as f:
        f.write(ds.name)


def load_arrays(directory):
    """Inverse of save_arrays: the same Dataset, bit for bit."""
    import os
    ld = lambda k: np.load(os.path.join(directory, k + '.npy'))
    meta = ld('meta')
    g = Graph(*[torch.from_numpy(ld(k)) for k in ('rowptr', 'col', 't_rowptr', 't_col')], int(meta[1]))
    for fn in sorted(os.listdir(directory)):
        if fn.startswith('ndata_'):
            g.ndata[fn[len('ndata_'):-4]] = torch.from_numpy(np.load(os.path.join(directory, fn)))
    if os.path.exists(os.path.join(directory, 'node_blocks.npy')):
        g.node_blocks = ld('node_blocks')
    sizes = ld('par_sizes')
    cat = ld('par_cat')
    offs = np.concatenate([[0], np.cumsum(sizes)])
    par_li = [cat[offs[i]:offs[i + 1]].copy() for i in range(len(sizes))]
    with open(os.path.join(directory, 'name.txt')) as f:
        name = f.read()
    return Dataset(num_classes=int(meta[0]), g=g, par_li=par_li, name=name)


def community_edges(n, size_lo, size_hi, exponent, out_deg, mixing, hub_frac, hub_mult, seed):
    """A graph whose parts are NOT planted for the partitioner: communities with power-law sizes in [size_lo, size_hi]
    (density ~ s^-exponent), every node sending out_deg (x hub_mult for hubs) edge stubs, a stub staying inside the node's
    community with probability 1 - mixing and going to a uniformly random node otherwise (the LFR benchmark's mixing
    parameter); symmetrised, one self loop per node; node ids are a random permutation (they say nothing about the
    communities).  Returns (src, dst, communities as lists of node ids)."""
    rs = np.random.RandomState(seed)
    sizes = []
    left = n
    a = 1.0 - exponent
    while left > 0:
        u = rs.random_sample()
        sz = int(((size_hi ** a - size_lo ** a) * u + size_lo ** a) ** (1.0 / a))      # inverse CDF of s^-exponent
        sz = min(max(sz, size_lo), size_hi, left)
        if left - sz < size_lo and left - sz > 0:
            sz = left
        sizes.append(sz)
        left -= sz
    sizes = np.array(sizes, np.int64)
    n_comm = sizes.shape[0]
    starts = np.zeros(n_comm + 1, np.int64)
    np.cumsum(sizes, out=starts[1:])
    comm_of = np.repeat(np.arange(n_comm), sizes)                 # in "community order" ...
    perm = rs.permutation(n).astype(np.int64)                     # ... which the node ids do not follow
    mult = np.ones(n, np.int64)
    n_hub = int(n * hub_frac)
    if n_hub:
        mult[rs.choice(n, n_hub, replace=False)] = hub_mult
    pos = np.repeat(np.arange(n, dtype=np.int64), out_deg * mult)
    inside = rs.random_sample(pos.shape[0]) >= mixing
    c = comm_of[pos]
    dst_pos = np.where(inside, starts[c] + (rs.random_sample(pos.shape[0]) * sizes[c]).astype(np.int64),
                       rs.randint(0, n, pos.shape[0]).astype(np.int64))
    s_, d_ = perm[pos], perm[dst_pos]
    nodes = np.arange(n, dtype=np.int64)
    src = np.concatenate([s_, d_, nodes])
    dst = np.concatenate([d_, s_, nodes])
    comms = [np.sort(perm[starts[q]:starts[q + 1]]) for q in range(n_comm)]
    return src, dst, comms


def community_dataset(name, n, n_feats, n_classes, seed, size_lo=30, size_hi=400, exponent=2.0, out_deg=48, mixing=0.3,
                      hub_frac=0.01, hub_mult=10):
    """Dataset over community_edges (all nodes are training nodes).  `par_li` holds the COMMUNITIES -- a reference point
    for quality, NOT a balanced k-way partition: training on this graph partitions it first (gist_partition_graph /
    dgl_compat.transform.metis_partition, like the reference's cache miss, cluster_gcn/sampler.py:49-51)."""
    src, dst, comms = community_edges(n, size_lo, size_hi, exponent, out_deg, mixing, hub_frac, hub_mult, seed)
    g = Graph.from_edges(src, dst, n)
    gen = torch.Generator().manual_seed(seed)
    g.ndata['feat'] = torch.randn(n, n_feats, generator=gen)
    g.ndata['label'] = torch.randint(0, n_classes, (n,), generator=gen)
    ones = torch.ones(n, dtype=torch.bool)
    g.ndata['train_mask'], g.ndata['val_mask'], g.ndata['test_mask'] = ones, ~ones, ~ones
    return Dataset(num_classes=n_classes, g=g, par_li=comms, name=name)


def reddit_communities(seed=0, n=153431):
    """Reddit-sized (N, F, C, mean degree as reddit-synth) with power-law communities of 30-400 nodes, mixing 0.3."""
    return community_dataset('reddit-communities', n, 602, 41, seed)


def reddit_synth(seed=0, n=153431, n_blocks=1500, train_frac=1.0):
    return make_block_dataset('reddit-synth', n, n_blocks, 602, 41, intra_deg=28, inter_deg=20,
                              seed=seed, train_frac=train_frac)


def amazon_synth(seed=1, n=1709997, n_blocks=15000, train_frac=1.0):
    return make_block_dataset('amazon-synth', n, n_blocks, 100, 47, intra_deg=7, inter_deg=4,
                              seed=seed, hub_frac=0.005, train_frac=train_frac)


def toy(seed=2, n=2400, n_blocks=24, n_feats=32, n_classes=5, train_frac=0.7):
    return make_block_dataset('toy', n, n_blocks, n_feats, n_classes, intra_deg=5, inter_deg=2,
                              seed=seed, train_frac=train_frac)


class CitationDataset(object):
    """The legacy DGL citation-dataset object gcn/train.py reads (:38-45,64): numpy `features`
    [N, F], `labels` [N], `train_mask` / `val_mask` / `test_mask` [N], `num_labels`, and `graph`
    = a networkx DiGraph (both directions of every citation, no self loops)."""

    def __init__(self, name, features, labels, train_mask, val_mask, test_mask, num_labels, src, dst):
        import networkx as nx
        self.name = name
        self.features, self.labels = features, labels
        self.train_mask, self.val_mask, self.test_mask = train_mask, val_mask, test_mask
        self.num_labels = int(num_labels)
        self.src, self.dst = src, dst
        g = nx.DiGraph()
        g.add_nodes_from(range(features.shape[0]))
        g.add_edges_from(zip(src.tolist(), dst.tolist()))
        self.graph = g


def citation_synth(name='cora-synth', n=2708, n_undirected=5278, n_feats=1433, n_classes=7,
                   train_per_class=20, n_val=500, n_test=1000, nnz_per_row=18, seed=2):
    """Cora-like citation graph: `n_undirected` distinct undirected
    edges without self loops, 80 % of them inside a class -- 2 * 5278 + 2708 self loops (added
    by the training script, gcn/train.py:66-68) = 13264 directed edges at the defaults; sparse
    0/1 bag-of-words features whose vocabulary is class-correlated, rows normalised to sum 1;
    Planetoid-style masks (20 per class / 500 / 1000)."""
    rs = np.random.RandomState(seed)
    labels = rs.randint(0, n_classes, n).astype(np.int64)
    by_class = [np.nonzero(labels == c)[0] for c in range(n_classes)]
    seen, src, dst = set(), [], []
    while len(src) < n_undirected:
        u = int(rs.randint(0, n))
        if rs.rand() < 0.8:
            peers = by_class[labels[u]]
            v = int(peers[rs.randint(0, len(peers))])
        else:
            v = int(rs.randint(0, n))
        if u == v or (min(u, v), max(u, v)) in seen:
            continue
        seen.add((min(u, v), max(u, v)))
        src.append(u)
        dst.append(v)
    src, dst = np.array(src, np.int64), np.array(dst, np.int64)
    feats = np.zeros((n, n_feats), np.float32)
    block = max(n_feats // n_classes, 1)
    for i in range(n):
        k_own = nnz_per_row * 2 // 3
        own = (labels[i] * block + rs.randint(0, block, k_own)) % n_feats
        other = rs.randint(0, n_feats, nnz_per_row - k_own)
        feats[i, own] = 1.0
        feats[i, other] = 1.0
    feats /= feats.sum(1, keepdims=True)
    train = np.zeros(n, bool)
    for c in range(n_classes):
        train[by_class[c][:train_per_class]] = True
    rest = np.nonzero(~train)[0]
    rest = rest[rs.permutation(len(rest))]
    val, test = np.zeros(n, bool), np.zeros(n, bool)
    val[rest[:n_val]] = True
    test[rest[n_val:n_val + n_test]] = True
    return CitationDataset(name, feats, labels, train, val, test, n_classes,
                           np.concatenate([src, dst]), np.concatenate([dst, src]))


def cora_synth(seed=2):
    return citation_synth('cora-synth', seed=seed)


SYNTHETIC = ('reddit-synth', 'amazon-synth', 'cora-synth', 'toy')


def load_dataset(name, data_root=None):
    """Real data for the reference's dataset names (`reddit`, `reddit-self-loop`, `amazon2m`:
    utils.load_data, cluster_gcn/utils.py:83-124) from `data_root` / $GIST_DATA_ROOT through
    gist_amd/ingest.py -- and an error naming what is missing when it is not there.  The seeded
    synthetic stand-ins are returned ONLY for their own explicit names (`reddit-synth`,
    `amazon-synth`, `cora-synth`, `toy`): a run under a real dataset's name never reports
    accuracies of fake data."""
    import os
    from . import ingest
    root = data_root or os.environ.get('GIST_DATA_ROOT')
    if name == 'reddit-synth':
        return reddit_synth(n=232965, train_frac=0.6586)      # 153431 / 232965 like Reddit
    if name == 'amazon-synth':
        return amazon_synth(n=2449029, train_frac=0.6982)
    if name == 'cora-synth':
        return cora_synth()
    if name == 'toy':
        return toy()
    real = ingest.try_load(name, root)          # reddit / reddit-self-loop / amazon2m / {name}-G.json
    if real is not None:
        return real
    synth = {'reddit': 'reddit-synth', 'reddit-self-loop': 'reddit-synth', 'amazon2m': 'amazon-synth'}
    raise FileNotFoundError(
        'gist_amd: dataset %r not found under data root %r (--data-root / $GIST_DATA_ROOT must hold '
        'reddit_data.npz + reddit[_self_loop]_graph.npz, or GraphSAGE-format %s-G.json, -feats.npy, '
        '-id_map.json, -class_map.json; gist_amd/ingest.py).%s Synthetic names: %s'
        % (name, root, name,
           ' For the seeded synthetic stand-in ask for it by name: --dataset %s.' % synth[name]
           if name in synth else '', ', '.join(SYNTHETIC)))
