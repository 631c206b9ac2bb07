"""Cost of the pair units by themselves: the synthetic batch of tests/test_kernels_gpu.py::_sibling_graph (20 parts, every row
40 in-part neighbours + 1 anywhere) without sibling parts, with one pair of siblings, with the test's five sibling blocks; prepared
aggregation at D = 4096, forward form, HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gist_amd import hip
from oracle import gist_oracle as O
from tests.test_kernels_gpu import _sibling_graph
dev = 'cuda:0'


def timeit(f, it_=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(it_):
        x, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x.record(); f(); c.record(); torch.cuda.synchronize(); ts.append(x.elapsed_time(c))
    ts.sort(); return ts[len(ts) // 2] * 1e3


d = 4096
for name, sib, cross in [('no siblings', (), 0), ('one pair of siblings', ((3, 7),), 45), ('five sibling blocks', ((3, 7), (10, 11), (11, 15), (10, 15)), 45),
                         ('one pair, 12 cross edges per row', ((3, 7),), 12)]:
    rs = np.random.RandomState(1)
    n, cuts, src, dst = _sibling_graph(rs, siblings=sib, cross_per_row=cross)
    rowptr, col = O.csr_from_edges(src, dst, n)
    rp = torch.from_numpy(rowptr.astype(np.int32)).to(dev); cl = torch.from_numpy(col.astype(np.int32)).to(dev)
    rb = torch.from_numpy(cuts.astype(np.int32)).to(dev)
    prep = hip.spmm_prepare(rp, cl, rb)
    z = torch.randn(n, 2 * d, device=dev)
    t = timeit(lambda: hip.spmm(rp, cl, z[:, :d], z[:, d:], row_blocks=rb, blocked=True, prepared=prep))
    t_acc = timeit(lambda: hip.spmm(rp, cl, z[:, :d], z[:, d:], row_blocks=rb, blocked=True, prepared=prep, accumulate=True))
    print('%-34s n=%d nnz=%d: %.1f us (accumulate form %.1f us)' % (name, n, col.size, t, t_acc), flush=True)
