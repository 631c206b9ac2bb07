"""Real-data ingestion (gist_amd/ingest.py): tiny fixtures written in the two
on-disk formats, checked against what cluster_gcn/AmazonDataset.py:25-118 / DGL's RedditDataset
would build from them (numpy restatement inline)."""
import json

import numpy as np
import scipy.sparse as sp

from gist_amd import datasets, ingest


def _edges(g):
    rp = g.rowptr.numpy().astype(np.int64)
    dst = np.repeat(np.arange(g.number_of_nodes()), np.diff(rp))
    return set(zip(g.col.numpy().tolist(), dst.tolist()))          # (src, dst)


def test_graphsage_dir_format(tmp_path):
    rs = np.random.RandomState(0)
    n, f, c = 12, 5, 3
    feats = rs.randn(n, f).astype(np.float32)
    feats[:, 2] = 7.0                                   # zero variance column: scale by 1
    ids = ['n%d' % i for i in range(n)]                 # non-digit ids, shuffled id_map
    perm = rs.permutation(n)
    id_map = {ids[i]: int(perm[i]) for i in range(n)}
    nodes = [{'id': ids[i], 'val': i % 4 == 1, 'test': i % 4 == 2} for i in range(n)]
    raw = [(0, 1), (1, 0), (2, 3), (3, 3), (4, 5), (4, 5), (6, 7), (8, 9), (10, 11), (11, 2)]
    links = [{'source': ids[a], 'target': ids[b]} for a, b in raw]
    nodes_json = nodes
    cls = {ids[i]: [1 if k == i % c else 0 for k in range(c)] for i in range(n)}
    np.save(tmp_path / 'tiny-feats.npy', feats)
    (tmp_path / 'tiny-G.json').write_text(json.dumps({'directed': False, 'nodes': nodes_json, 'links': links}))
    (tmp_path / 'tiny-id_map.json').write_text(json.dumps(id_map))
    (tmp_path / 'tiny-class_map.json').write_text(json.dumps(cls))

    ds = datasets.load_dataset('tiny', str(tmp_path))
    g = ds.g
    assert ds.num_classes == c and g.number_of_nodes() == n
    want = set()
    for a, b in raw:
        u, v = int(perm[a]), int(perm[b])
        want.add((u, v))
        want.add((v, u))
    assert _edges(g) == want                             # symmetrised, duplicates merged
    val = np.zeros(n, bool)
    test = np.zeros(n, bool)
    labels = np.zeros(n, np.int64)
    for i in range(n):
        val[perm[i]] = i % 4 == 1
        test[perm[i]] = i % 4 == 2
        labels[perm[i]] = i % c
    assert np.array_equal(g.ndata['val_mask'].numpy(), val)
    assert np.array_equal(g.ndata['test_mask'].numpy(), test)
    assert np.array_equal(g.ndata['train_mask'].numpy(), ~(val | test))
    assert np.array_equal(g.ndata['label'].numpy(), labels)
    tr = ~(val | test)
    mu, sd = feats[tr].astype(np.float64).mean(0), feats[tr].astype(np.float64).std(0)
    sd[sd == 0] = 1.0
    assert np.allclose(g.ndata['feat'].numpy(), (feats - mu) / sd, atol=1e-6)
    assert np.allclose(g.ndata['feat'].numpy()[:, 2], 0.0)


def test_graphsage_dir_integer_ids_and_scalar_classes(tmp_path):
    n = 6
    nodes = [{'id': i, 'val': i == 4, 'test': i == 5} for i in range(n)]
    links = [{'source': 0, 'target': 1}, {'source': 2, 'target': 1}, {'source': 5, 'target': 5}]
    np.save(tmp_path / 'amazon2M-feats.npy', np.arange(n * 2, dtype=np.float32).reshape(n, 2))
    (tmp_path / 'amazon2M-G.json').write_text(json.dumps({'nodes': nodes, 'links': links}))
    (tmp_path / 'amazon2M-id_map.json').write_text(json.dumps({str(i): i for i in range(n)}))
    (tmp_path / 'amazon2M-class_map.json').write_text(json.dumps({str(i): i % 2 for i in range(n)}))
    ds = datasets.load_dataset('amazon2m', str(tmp_path))
    assert ds.num_classes == 47                           # AmazonDataset.num_classes is a constant
    assert _edges(ds.g) == {(0, 1), (1, 0), (2, 1), (1, 2), (5, 5)}
    assert ds.g.ndata['label'].tolist() == [0, 1, 0, 1, 0, 1]
    assert ds.g.ndata['train_mask'].tolist() == [True, True, True, True, False, False]


def test_dgl_reddit_files(tmp_path):
    rs = np.random.RandomState(1)
    n = 9
    row = np.array([0, 1, 2, 3, 3, 8] + list(range(n)))
    col = np.array([1, 0, 5, 4, 4, 2] + list(range(n)))     # a duplicate edge and self loops
    sp.save_npz(tmp_path / 'reddit_self_loop_graph.npz', sp.coo_matrix((np.ones(len(row)), (row, col)), shape=(n, n)))
    feats = rs.randn(n, 4).astype(np.float32)
    labels = rs.randint(0, 5, n)
    types = np.array([1, 1, 2, 3, 1, 1, 2, 3, 1])
    np.savez(tmp_path / 'reddit_data.npz', feature=feats, label=labels, node_types=types)
    ds = datasets.load_dataset('reddit-self-loop', str(tmp_path))
    g = ds.g
    assert ds.num_classes == int(labels.max()) + 1
    assert g.number_of_edges() == len(row)                  # multi-edges kept, like from_scipy(coo)
    assert _edges(g) == set(zip(row.tolist(), col.tolist()))
    assert np.array_equal(g.ndata['feat'].numpy(), feats)
    assert np.array_equal(g.ndata['train_mask'].numpy(), types == 1)
    assert np.array_equal(g.ndata['val_mask'].numpy(), types == 2)
    assert np.array_equal(g.ndata['test_mask'].numpy(), types == 3)


def test_falls_back_to_synthetic_without_files(tmp_path):
    assert ingest.try_load('reddit-self-loop', str(tmp_path)) is None
    assert ingest.try_load('amazon2m', None) is None


def test_real_dataset_names_never_fall_back_to_synthetic(tmp_path):
    """A missing or mistyped data root under a real dataset's name is an error that names the
    missing files -- never a silent synthetic graph (its accuracies would be reported under the
    real dataset's name)."""
    import pytest
    from gist_amd import datasets
    for name in ('reddit', 'reddit-self-loop', 'amazon2m'):
        with pytest.raises(FileNotFoundError) as e:
            datasets.load_dataset(name, str(tmp_path))
        assert 'synth' in str(e.value) and name in str(e.value)
        with pytest.raises(FileNotFoundError):
            datasets.load_dataset(name, None)
    assert datasets.load_dataset('toy').name == 'toy'
