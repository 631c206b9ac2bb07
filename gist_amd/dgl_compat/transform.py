"""dgl.transform.metis_partition (cluster_gcn/partition_utils.py:7,12).

METIS itself is a third-party library the reference borrows through DGL; it is
out of scope and absent offline.  Partition lists
are an INPUT here: load them from the reference's cache format
(`../data/{dataset}_{psize}.npy`, sampler.py:44-51) or pass `par_li=` to
gist_amd.sampler.ClusterIter.
"""


def metis_partition(g, k):
    raise RuntimeError(
        'gist_amd: METIS is not bundled. Provide the partition cache '
        '../data/<dataset>_<psize>.npy (reference format) or pass par_li= to ClusterIter.')
