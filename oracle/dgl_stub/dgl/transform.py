"""Stub of dgl.transform.  METIS is not available offline; partition lists are
treated as an INPUT, so this raises if called."""


def metis_partition(g, k):
    raise RuntimeError('METIS is not available in the build container; '
                       'pass partition lists explicitly')
