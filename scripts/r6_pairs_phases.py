"""Dev tool: phase stamps (s_memrealtime, 100 MHz) of the first workgroup of the blocks-with-pairs class of the prepared
block-dense SpMM on a synthetic batch whose block 0 has one sibling.
    GIST_EXTRA_FLAGS="-DMF_PROBE -DMF_PROBE_PAIRS" GIST_LIB_OUT=$PWD/ab/libgist_mfpp.so python gist_amd/build.py
    GIST_LIB_PATH=$PWD/ab/libgist_mfpp.so python scripts/r6_pairs_phases.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gist_amd import hip, _lib
from oracle import gist_oracle as O
from tests.test_kernels_gpu import _sibling_graph
dev = 'cuda:0'
L = _lib.load()
L.gist_mf_probe_read.argtypes = [ctypes.c_void_p]
d = 4096
rs = np.random.RandomState(1)
n, cuts, src, dst = _sibling_graph(rs, siblings=((0, 7),), cross_per_row=45)
rowptr, col = O.csr_from_edges(src, dst, n)
rp = torch.from_numpy(rowptr.astype(np.int32)).to(dev); cl = torch.from_numpy(col.astype(np.int32)).to(dev)
rb = torch.from_numpy(cuts.astype(np.int32)).to(dev)
prep = hip.spmm_prepare(rp, cl, rb)
z = torch.randn(n, 2 * d, device=dev)
for _ in range(5):
    hip.spmm(rp, cl, z[:, :d], z[:, d:], row_blocks=rb, blocked=True, prepared=prep)
torch.cuda.synchronize()
buf = np.zeros(64, np.uint64)
assert L.gist_mf_probe_read(buf.ctypes.data) == 0
t = (buf.astype(np.int64) - int(buf[0])) / 100.0
print('stamps (us since the workgroup started; 0 = not reached):')
for i in range(64):
    if buf[i]:
        print(i, '%.2f' % t[i])
