"""Dev tool (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes): the full-graph evaluation's
aggregation at D = 4096 on the Reddit-size synthetic graph, ONE dispatch of each form after a warm-up of the
same call: (a) one gather pass over A (spmm_csr_rowsplit_kernel), (b) A_diag on the matrix cores
(spmm_csr_mfma_kernel), (c) A_rest gathered in 512-float column tiles (spmm_csr_rowsplit_kernel, accumulate).
Prints the algorithmic bytes of each.  scripts/make_profiles_r3.py pairs them with the counters."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gist_amd import datasets, hip
from gist_amd.engine import ParamArena, dims_for
from gist_amd.trainer import FullGraphEvaluator

dev = torch.device('cuda', 0)
N, D = 232965, 4096
ds = datasets.make_block_dataset('reddit-full-synth', N, 2278, 602, 41, intra_deg=123, inter_deg=102, seed=0,
                                 train_frac=0.6586)
g = ds.g.to(dev)
n, nnz = g.number_of_nodes(), g.number_of_edges()
sizes = np.array([len(b) for b in np.array_split(np.arange(N), 2278)])
bounds = np.concatenate([[0], np.cumsum(sizes)])
dims = dims_for(602, D, 41, 2)
arena = ParamArena(dims, dev, with_grads=False)
ev = FullGraphEvaluator(ds.g, dims, True, arena, dev, row_block=n, node_blocks=bounds)
sp = ev.split
norm = g.norm()
x = torch.randn(n, D, device=dev)
y = torch.empty(n, D, device=dev)


def one_pass():
    hip.spmm(g.rowptr, g.col, x, y, out_scale=norm)


def diag():
    hip.spmm(sp['rowptr_d'], sp['col_d'], x, y, out_scale=norm, row_blocks=sp['blocks'][0], prepared=sp['prepared'][0])


def rest():
    for c0 in range(0, D, 512):
        hip.spmm(sp['rowptr_r'], sp['col_r'], x[:, c0:c0 + 512], y[:, c0:c0 + 512], out_scale=norm, accumulate=True)


times = {}
for name, f in (('one_pass', one_pass), ('diag', diag), ('rest', rest)):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize()
    times[name] = a.elapsed_time(b)
print(json.dumps({'n': n, 'nnz': nnz, 'D': D, 'diag_edges': sp['diag_edges'], 'rest_edges': sp['rest_edges'],
                  'ms': times,
                  'algorithmic_bytes': {'one_pass': 4.0 * (n + 1) + 4.0 * nnz + 8.0 * n * D,
                                        'diag': 4.0 * (n + 1) + 4.0 * sp['diag_edges'] + 8.0 * n * D,
                                        'rest': 4.0 * (n + 1) + 4.0 * sp['rest_edges'] + 12.0 * n * D},
                  'note': 'rest reads y (accumulate): 12 N D; the split as a whole moves 4(N+1)*2 + 4 nnz + 20 N D'}))
