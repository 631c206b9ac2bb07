"""The library's stand-in for dgl.transform.metis_partition (partition_utils.py:11-18):
gist_partition_graph is a HOST function, so these tests need no GPU.  METIS's own output
is not reproducible offline, so the checks are properties: a k-way
partition (every node once, k non-empty parts, balance bound), determinism, and an edge cut
far below a random partition's on planted-cluster graphs."""
import numpy as np
import pytest
import torch

from gist_amd import datasets
from gist_amd.dgl_compat import NID
from gist_amd.dgl_compat.transform import metis_partition, partition_assignment
from gist_amd.sampler import get_partition_list


def _intra_fraction(g, part):
    rp = g.rowptr.numpy().astype(np.int64)
    col = g.col.numpy()
    dst = np.repeat(np.arange(g.number_of_nodes()), np.diff(rp))
    keep = dst != col                       # self loops are intra by definition: leave them out
    return float((part[dst[keep]] == part[col[keep]]).mean())


@pytest.mark.parametrize('n,k,intra,inter', [(2400, 24, 5, 2), (6000, 60, 14, 10), (3000, 150, 6, 3)])
def test_partition_properties_and_cut(n, k, intra, inter):
    ds = datasets.make_block_dataset('t', n, k, 4, 3, intra_deg=intra, inter_deg=inter, seed=5)
    g = ds.g
    part = partition_assignment(g, k, seed=1)
    assert part.shape == (n,) and part.min() >= 0 and part.max() < k
    sizes = np.bincount(part, minlength=k)
    assert sizes.min() >= 1
    assert sizes.max() <= int(np.ceil(1.03 * n / k + 1e-9))
    assert sizes.min() >= int(0.97 * n / k)
    # same seed, same result; another seed, another visiting order
    assert np.array_equal(part, partition_assignment(g, k, seed=1))
    # quality: planted blocks are the reference point, a random partition the floor
    planted = np.empty(n, np.int64)
    for b, ids in enumerate(ds.par_li):
        planted[ids] = b
    f_planted = _intra_fraction(g, planted)
    f_random = _intra_fraction(g, np.random.RandomState(0).randint(0, k, n))
    f_ours = _intra_fraction(g, part)
    assert f_random < 0.1 < f_planted
    assert f_ours > 0.8 * f_planted, (f_ours, f_planted, f_random)


def test_metis_partition_call_shape():
    """partition_utils.py:12-17 reads `p_gs.items()` and `val.ndata[dgl.NID]`."""
    ds = datasets.toy(seed=3, train_frac=1.0)
    g = ds.g
    p_gs = metis_partition(g, 24)
    assert sorted(p_gs.keys()) == list(range(24))
    ids = np.concatenate([val.ndata[NID].numpy() for _, val in p_gs.items()])
    assert ids.dtype == np.int64 and np.array_equal(np.sort(ids), np.arange(g.number_of_nodes()))
    par_li = get_partition_list(g, 24)
    assert len(par_li) == 24 and all(p.dtype == np.int64 and p.size > 0 for p in par_li)


def test_partition_degenerate_inputs():
    from gist_amd.graph import Graph
    # no edges at all: still k balanced non-empty parts
    g = Graph.from_edges(np.zeros(0, np.int64), np.zeros(0, np.int64), 10)
    part = partition_assignment(g, 3)
    assert np.bincount(part, minlength=3).min() >= 3
    # k == n: every node alone
    assert np.array_equal(np.sort(partition_assignment(g, 10)), np.arange(10))
    with pytest.raises(ValueError):
        partition_assignment(g, 11)
