"""Full-graph evaluation at Reddit scale: the aggregation over N = 232965 nodes / ~115 M edges
(X up to 3.8 GB at D = 4096) and the whole eval-mode forward of the 4096-wide model (cluster_gcn/utils.py:70-80), on
THREE synthetic graphs that differ only in where a node's inter-part edges go (datasets.sbm_edges inter_locality):
  0.0  uniformly random nodes -- every off-diagonal block pair holds ~10 edges: nothing but gathers (the worst case)
  0.8  80 % to the part's 8 neighbour parts, 20 % uniform
  1.0  all to the 8 neighbour parts
Round 4: the evaluator runs dense off-diagonal block pairs (>= 300 edges) as counts x features on the bf16x3 matrix cores
(gist_spmm_block_units_f32: the batch kernel with the X tile taken from the column block) next to the block-diagonal part and
gathers only the rest.
Prints one JSON line.

    python scripts/eval_bench.py [--localities 0,0.8,1] [--half-degree 225]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument('--half-degree', type=int, default=225, help='intra+inter out-edges drawn per node')
ap.add_argument('--n', type=int, default=232965)
ap.add_argument('--localities', type=str, default='0,0.8,1')
ap.add_argument('--pair-min-edges', type=int, default=300)
args = ap.parse_args()

from gist_amd import datasets, hip
from gist_amd.engine import ParamArena, dims_for
from gist_amd.trainer import FullGraphEvaluator

dev = torch.device('cuda', 0)
N_BLOCKS = 2278


def timeit(f, iters=5):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def variant(loc, first):
    t0 = time.time()
    intra = int(args.half_degree * 0.55)
    ds = datasets.make_block_dataset('reddit-full-synth', args.n, N_BLOCKS, 602, 41, intra_deg=intra,
                                     inter_deg=args.half_degree - intra, seed=0, train_frac=0.6586,
                                     inter_locality=loc)
    g = ds.g.to(dev)
    n, nnz = g.number_of_nodes(), g.number_of_edges()
    out = {'inter_locality': loc, 'n': n, 'nnz': nnz, 'graph_build_s': round(time.time() - t0, 1)}
    norm = g.norm()
    sizes = np.array([len(b) for b in np.array_split(np.arange(args.n), N_BLOCKS)])
    bounds = np.concatenate([[0], np.cumsum(sizes)])
    if first:      # the plain one-pass gather at three widths (as in rounds 1-3)
        out['spmm_one_pass'] = []
        for d in (602, 1024, 4096):
            z = torch.randn(n, 2 * d, device=dev)
            ms = timeit(lambda: hip.spmm(g.rowptr, g.col, z[:, :d], z[:, d:], out_scale=norm))
            alg = 4.0 * (n + 1) + 4.0 * nnz + 8.0 * n * d
            out['spmm_one_pass'].append({'D': d, 'ms': round(ms, 3), 'algorithmic_GB': round(alg / 1e9, 3),
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
    acc = ev.accuracy('val_mask')
    torch.cuda.synchronize()
    ev.invalidate_input_aggregation()
    t0 = time.time()
    acc = ev.accuracy('val_mask')
    torch.cuda.synchronize()
    out['eval_forward_H4096_one_pass_first_s'] = round(time.time() - t0, 4)      # (aggregates the input features)
    t0 = time.time()
    acc = ev.accuracy('val_mask')
    torch.cuda.synchronize()
    out['eval_forward_H4096_one_pass_s'] = round(time.time() - t0, 4)           # (their aggregation kept: every later one)
    logits_one_pass = ev.forward().clone()
    del ev
    for tag, pme in (('diag_plus_gather', 0), ('diag_plus_pairs_plus_gather', args.pair_min_edges)):
        t0 = time.time()
        ev2 = FullGraphEvaluator(ds.g, dims, True, arena, dev, node_blocks=bounds, pair_min_edges=pme)
        torch.cuda.synchronize()
        r = {'setup_s': round(time.time() - t0, 2)}
        acc2 = ev2.accuracy('val_mask')
        torch.cuda.synchronize()
        ev2.invalidate_input_aggregation()
        t0 = time.time()
        acc2 = ev2.accuracy('val_mask')
        torch.cuda.synchronize()
        r['eval_forward_H4096_first_s'] = round(time.time() - t0, 4)      # (the run's first evaluation: aggregates the input features)
        t0 = time.time()
        acc2 = ev2.accuracy('val_mask')
        torch.cuda.synchronize()
        sp = ev2.split
        r['eval_forward_H4096_s'] = round(time.time() - t0, 4)            # (every later one: that product is kept)
        r['edges'] = {'inside_blocks': sp['diag_edges'], 'dense_pairs': sp['pair_edges'], 'rest_gathered': sp['rest_edges'],
                      'n_pairs': sp['n_pairs']}
        r['max_abs_diff_logits_vs_one_pass'] = float((ev2.forward() - logits_one_pass).abs().max().item())
        r['val_acc'] = round(acc2, 4)
        # the aggregation alone at D = 4096, part by part, all row blocks
        d = 4096
        x = torch.randn(n, d, device=dev)
        zr = torch.empty(max(b - a for a, b in zip(ev2.row_cuts[:-1], ev2.row_cuts[1:])), d, device=dev)
        cuts = list(zip(ev2.row_cuts[:-1], ev2.row_cuts[1:]))

        def diag():
            for bi, (r0, r1) in enumerate(cuts):
                hip.spmm(sp['rowptr_d'][r0:r1 + 1], sp['col_d'], x[r0:r1], zr[:r1 - r0], out_scale=norm[r0:r1],
                         row_blocks=sp['blocks'][bi], prepared=sp['prepared'][bi])

        def pairs():
            for bi, (r0, r1) in enumerate(cuts):
                for (a0, a1) in sp['rounds'].get(bi, []):
                    hip.spmm_block_units(sp['units'][a0:a1], sp['images'][a0:a1], x, zr[:r1 - r0], out_scale=norm[r0:r1],
                                         accumulate=True)

        def rest(ct=512):
            for bi, (r0, r1) in enumerate(cuts):
                for c0 in range(0, d, ct):
                    hip.spmm(sp['rowptr_r'][r0:r1 + 1], sp['col_r'], x[:, c0:c0 + ct], zr[:r1 - r0, c0:c0 + ct],
                             out_scale=norm[r0:r1], accumulate=True)

        ch = sp.get('chains')

        def chains():      # round 5: diagonal block + dense pairs of a row block as one chain (y written once)
            for bi, (r0, r1) in enumerate(cuts):
                cp, u_lo, u_hi = ch['chunks'][bi]
                hip.spmm_block_chains(cp, ch['units'][u_lo:u_hi], ch['images'][u_lo:u_hi], x, zr[:r1 - r0], out_scale=norm[r0:r1])

        if ch is not None:
            md, mp = timeit(chains, 3), 0.0
        else:
            md = timeit(diag, 3)
            mp = timeit(pairs, 3) if sp['n_pairs'] else 0.0
        mr = timeit(rest, 3) if sp['rest_edges'] else 0.0
        alg = 4.0 * (n + 1) + 4.0 * nnz + 8.0 * n * d
        r['aggregation_D4096'] = {
            'ms_inside_blocks_bf16x3_matrix_cores': round(md, 3), 'ms_dense_pairs_bf16x3_matrix_cores': round(mp, 3),
            'chains': ch is not None,      # True: ms_inside_blocks is diagonal blocks + dense pairs in one launch per row chunk
            'ms_rest_gather_512_float_tiles': round(mr, 3), 'ms_total': round(md + mp + mr, 3),
            'algorithmic_GB': round(alg / 1e9, 3), 'achieved_GBps': round(alg / (md + mp + mr) / 1e6, 1),
            'frac_of_8TBps': round(alg / (md + mp + mr) / 1e6 / 8000.0, 4),
            'us_per_dense_pair': round(mp * 1e3 / sp['n_pairs'], 3) if sp['n_pairs'] else None,
            'ns_per_gathered_edge': round(mr * 1e6 / sp['rest_edges'], 3) if sp['rest_edges'] else None}
        out[tag] = r
        del ev2, x, zr, sp
        torch.cuda.empty_cache()
        if pme == 0 and first is False and loc == 0.0:
            break
    out['val_acc_random_weights'] = round(acc, 4)
    return out


res = {'variants': []}
for j, loc in enumerate([float(v) for v in args.localities.split(',')]):
    res['variants'].append(variant(loc, j == 0))
    print('variant %s done' % loc, file=sys.stderr, flush=True)
    torch.cuda.empty_cache()
res['eval_gemm_tflop'] = round(sum(2.0 * args.n * 2 * i * o for (i, o) in dims_for(602, 4096, 41, 2)) / 1e12, 2)
print(json.dumps(res))
