"""Full-graph evaluation at Reddit scale: the HBM-bound SpMM
(N = 232965, ~115 M edges, X up to 3.8 GB at D = 4096) and the whole eval-mode forward of
the 4096-wide model (cluster_gcn/utils.py:70-80).  Prints one JSON line.

    python scripts/eval_bench.py [--edges-per-node 225]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument('--half-degree', type=int, default=225, help='intra+inter out-edges drawn per node')
ap.add_argument('--n', type=int, default=232965)
args = ap.parse_args()

from gist_amd import datasets, hip
from gist_amd.engine import ParamArena, dims_for
from gist_amd.trainer import FullGraphEvaluator

dev = torch.device('cuda', 0)
t0 = time.time()
intra = int(args.half_degree * 0.55)
ds = datasets.make_block_dataset('reddit-full-synth', args.n, 2278, 602, 41, intra_deg=intra,
                                 inter_deg=args.half_degree - intra, seed=0, train_frac=0.6586)
g = ds.g.to(dev)
n, nnz = g.number_of_nodes(), g.number_of_edges()
gen_s = time.time() - t0


def timeit(f, iters=5):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


out = {'n': n, 'nnz': nnz, 'graph_build_s': round(gen_s, 1), 'spmm': []}
norm = g.norm()
for d in (602, 1024, 4096):
    z = torch.randn(n, 2 * d, device=dev)
    ms = timeit(lambda: hip.spmm(g.rowptr, g.col, z[:, :d], z[:, d:], out_scale=norm))
    alg = 4.0 * (n + 1) + 4.0 * nnz + 8.0 * n * d
    sizes = np.array([len(b) for b in np.array_split(np.arange(args.n), 2278)])
    rb = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)).to(dev)
    ms_b = ms_u = None
    if d % 4 == 0:
        ms_b = timeit(lambda: hip.spmm(g.rowptr, g.col, z[:, :d], z[:, d:], out_scale=norm, row_blocks=rb), 3)
        ms_u = timeit(lambda: hip.spmm(g.rowptr, g.col, z[:, :d], z[:, d:], out_scale=norm, blocked=True), 3)
    out['spmm'].append({'D': d, 'ms': round(ms, 3), 'ms_lds_blocks': ms_b and round(ms_b, 3),
                        'ms_lds_uniform128': ms_u and round(ms_u, 3), 'algorithmic_GB': round(alg / 1e9, 3),
                        'achieved_GBps': round(alg / ms / 1e6, 1),
                        'frac_of_8TBps': round(alg / ms / 1e6 / 8000.0, 4),
                        'gather_TBps': round(4.0 * nnz * d / ms / 1e9, 2)})
    del z
dims = dims_for(602, 4096, 41, 2)
arena = ParamArena(dims, dev, with_grads=False)
rs = torch.Generator().manual_seed(0)
for k, (i, o) in enumerate(dims):
    arena.W[k].copy_((torch.rand(o, 2 * i, generator=rs) - 0.5) * (2.0 / np.sqrt(2 * i)))
ev = FullGraphEvaluator(ds.g, dims, True, arena, dev, node_blocks=False)      # one gather pass
ev.accuracy('val_mask')
torch.cuda.synchronize()
t0 = time.time()
acc = ev.accuracy('val_mask')
torch.cuda.synchronize()
out['eval_forward_H4096_s'] = round(time.time() - t0, 4)
logits_one_pass = ev.forward().clone()
del ev
# the same evaluation with the aggregation split into block-diagonal (matrix cores) + remainder (gather)
bounds = np.concatenate([[0], np.cumsum(sizes)])
t0 = time.time()
ev2 = FullGraphEvaluator(ds.g, dims, True, arena, dev, node_blocks=bounds)
torch.cuda.synchronize()
out['split_setup_s'] = round(time.time() - t0, 2)
acc2 = ev2.accuracy('val_mask')
torch.cuda.synchronize()
t0 = time.time()
acc2 = ev2.accuracy('val_mask')
torch.cuda.synchronize()
out['eval_forward_H4096_split_s'] = round(time.time() - t0, 4)
out['split_edges'] = {'inside_blocks': ev2.split['diag_edges'], 'rest': ev2.split['rest_edges']}
out['split_max_abs_diff_logits'] = float((ev2.forward() - logits_one_pass).abs().max().item())
out['val_acc_split'] = round(acc2, 4)
# the aggregation alone at D = 4096 and 1024, both parts, all row blocks
sp = ev2.split
for d in (1024, 4096):
    x = torch.randn(n, d, device=dev)
    zr = torch.empty(max(b - a for a, b in zip(ev2.row_cuts[:-1], ev2.row_cuts[1:])), d, device=dev)

    def diag():
        for bi, (r0, r1) in enumerate(zip(ev2.row_cuts[:-1], ev2.row_cuts[1:])):
            hip.spmm(sp['rowptr_d'][r0:r1 + 1], sp['col_d'], x[r0:r1], zr[:r1 - r0], out_scale=norm[r0:r1],
                     row_blocks=sp['blocks'][bi], prepared=sp['prepared'][bi])

    def rest(ct=None):
        ct = ct or d
        for bi, (r0, r1) in enumerate(zip(ev2.row_cuts[:-1], ev2.row_cuts[1:])):
            for c0 in range(0, d, ct):
                hip.spmm(sp['rowptr_r'][r0:r1 + 1], sp['col_r'], x[:, c0:c0 + ct], zr[:r1 - r0, c0:c0 + ct],
                         out_scale=norm[r0:r1], accumulate=True)

    tiles = {ct: round(timeit(lambda: rest(ct), 3), 3) for ct in (128, 256, 512, 1024) if ct <= d}
    md, mr = timeit(diag, 3), min(tiles.values())
    out.setdefault('rest_gather_ms_by_column_tile', {})[str(d)] = dict(tiles, untiled=round(timeit(rest, 3), 3))
    alg = 4.0 * (n + 1) + 4.0 * nnz + 8.0 * n * d
    out.setdefault('spmm_split', []).append({
        'D': d, 'ms_inside_blocks_matrix_cores': round(md, 3), 'ms_rest_gather': round(mr, 3),
        'ms_total': round(md + mr, 3), 'achieved_GBps': round(alg / (md + mr) / 1e6, 1),
        'frac_of_8TBps': round(alg / (md + mr) / 1e6 / 8000.0, 4),
        'rest_gather_TBps': round(4.0 * sp['rest_edges'] * d / mr / 1e9, 2)})
    del x, zr
out['eval_gemm_tflop'] = round(sum(2.0 * n * 2 * i * o for (i, o) in dims) / 1e12, 2)
out['val_acc_random_weights'] = round(acc, 4)
print(json.dumps(out))
