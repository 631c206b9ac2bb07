"""gist_amd -- MI355X-native implementation of GIST's hot path.

GraphSAGE forward/backward over METIS cluster sub-graphs, cluster batch
extraction and the IST (independent sub-GCN) dispatch/sync, as hand-written
gfx950 HIP kernels behind a C ABI (include/gist_hip.h, libgist_hip.so), with a
Python host side that mirrors the reference's module / sampler / wrapper API
(wolfecameron/GIST: cluster_gcn/modules.py, sampler.py,
cluster_gcn_ist_distrib.py).

There is no CPU fallback: compute entry points raise if libgist_hip.so is
missing or if they are handed CPU tensors.
"""
__version__ = '0.1.0'
