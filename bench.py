#!/usr/bin/env python3
"""bench.py -- epochs/sec of GIST's training hot path on MI355X, Reddit-like synthetic data.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json metric: "epochs/sec + SpMM GB/s, Reddit 4096-wide GraphSAGE at
1/2/4/8 GPUs"):
  * graph: Reddit-like block model, N_train=153431, F=602, C=41, 1500 parts, batch = 20
    parts (~2046 rows, ~1.3e5 in-batch edges), 75 steps per epoch; synthetic, seed 0
  * model: GraphSAGE n_hidden=4096, n_layers=2 (3 SAGE layers), LayerNorm, dropout 0.2,
    Adam lr 0.01 -- fp32 end to end
  * N = 1: the full-width model on one GPU (the cluster_gcn.py path, which is what the
    reference's sweeps use for the 1-GPU point)
  * N > 1: GIST with S = N sub-GCNs of width 4096/N, one per GPU, weight sync every 100
    iterations through one RCCL all-gather (cluster_gcn_ist_distrib.py path)
A step = batch extraction (on device) + forward + CE + backward + Adam on one cluster
batch, plus the sync/dispatch work that falls on that iteration.  An epoch = 75 steps;
under GIST every rank trains n_epochs/S epochs (cluster_gcn_ist_distrib.py:385), so the
job's throughput is the SUM of the ranks' epochs/sec ("weak": per-GPU batch stream fixed).

Projections: fp32 in, fp32 out, fp32 accumulation.  By default (--gemm-mode f16x3) every
projection >= 16 GFLOP forms its products as 3 f16-split terms on v_mfma_f32_16x16x32_f16
(gist_amd/csrc/gemm_h3.hip; error against float64 equal to or below the fp32-MFMA kernel's,
tests/test_gemm_h3_gpu.py, tests/test_e2e_gpu.py); --gemm-mode f32 keeps all of them on
v_mfma_f32_32x32x2_f32.  At N=1 the line also carries `f32_mfma`: the same workload re-timed
in mode f32 in the same process.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: the projection GEMM),
`roofline_spmm` (the SpMM against HBM), `cpu_baseline` (the oracle timed on host cores).
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STEPS_PER_EPOCH = 75
MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: Peak FP32 (matrix)
MFMA_F16_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: Peak BF16/FP16 MFMA, dense
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E peak BW (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=150)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--n-hidden', type=int, default=4096)
    ap.add_argument('--n-layers', type=int, default=2)
    ap.add_argument('--dropout', type=float, default=0.2)
    ap.add_argument('--iter-per-site', type=int, default=100)
    ap.add_argument('--gemm-mode', choices=['f16x3', 'f32'], default='f16x3',
                    help='products of the large projections: 3-term f16 split on the f16 matrix '
                         'cores (fp32-level accuracy) or v_mfma_f32_32x32x2_f32')
    ap.add_argument('--no-f32-rerun', action='store_true',
                    help='N=1, mode f16x3: skip re-timing the workload in mode f32')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=3)
    ap.add_argument('--no-kernel-timing', action='store_true',
                    help='skip HIP-event bracketing of SpMM/GEMM launches')
    ap.add_argument('--timing-every', type=int, default=8,
                    help='bracket the SpMM/GEMM launches of every N-th timed step with HIP events '
                         '(an event pair per launch serialises kernel boundaries: ~80 us/step '
                         'when every step is instrumented, i.e. it would depress `value`)')
    return ap.parse_args()


def cpu_baseline(ds, par_order, dims, use_layernorm, n_steps, seed):
    """The oracle (oracle/gist_oracle.py, numpy + OpenBLAS + OpenMP C SpMM) on the host
    cores, same workload: first `n_steps` batches of the epoch, full step each."""
    from oracle import gist_oracle as O
    from oracle import train_oracle as TO
    g = ds.g
    rp = g.rowptr.numpy().astype(np.int64)
    cl = g.col.numpy().astype(np.int64)
    tg = TO.TrainGraph(rp, cl, g.ndata['feat'].numpy(), g.ndata['label'].numpy().astype(np.int64))
    rs = np.random.RandomState(seed)
    params = []
    for (i, o) in dims:
        stdv = 1.0 / np.sqrt(2 * i)
        params.append((rs.uniform(-stdv, stdv, (o, 2 * i)).astype(np.float32),
                       rs.uniform(-stdv, stdv, o).astype(np.float32)))
    opt = O.new_opt_state(params)
    times = []
    for j in range(n_steps + 1):
        ids = np.concatenate(par_order[j * 20:(j + 1) * 20]).astype(np.int64)
        t0 = time.time()
        rpb, clb, trp, tcl, x, y = tg.batch(ids)
        O.train_step(rpb, clb, trp, tcl, x, y, params, opt, use_layernorm, 0.01)
        times.append(time.time() - t0)
    step = float(np.median(times[1:]))           # first step = warm-up
    return step


def measured_copy_gbs(dev, n_bytes=1 << 30, reps=5):
    """Achievable HBM bandwidth ('measure achievable with a device copy
    kernel'): read + write bytes of a large device-to-device copy over its HIP-event time."""
    try:
        import torch
        src = torch.empty(n_bytes // 4, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        dst.copy_(src)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            dst.copy_(src)
        b.record()
        torch.cuda.synchronize()
        return 2.0 * n_bytes * reps / (a.elapsed_time(b) * 1e-3) / 1e9
    except Exception as e:          # reference measurement only: never fail the bench on it
        print('bench: copy-bandwidth probe failed: %r' % (e,), file=sys.stderr)
        return 0.0


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run '
                             '--nproc-per-node %d' % (args.gpus, args.gpus))
    # GIST_BENCH_SHARED_GPU=1 (validation only, never a reported number): every rank uses
    # cuda:0 and the collectives are staged through the host over gloo, so the N>1 logic can
    # be exercised on a 1-GPU box.  The product path is RCCL (backend "nccl").
    shared_gpu = os.environ.get('GIST_BENCH_SHARED_GPU') == '1'
    dev = torch.device('cuda', 0 if shared_gpu else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if shared_gpu:
            dist.init_process_group(backend='gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group(backend='nccl', rank=rank, world_size=world, device_id=dev)

    from gist_amd import datasets, hip
    from gist_amd.engine import SageEngine, dims_for
    from gist_amd.sampler import EngineClusterIter

    hip.gemm_mode(args.gemm_mode)       # before any workspace is sized
    seed = 0
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    ds = datasets.reddit_synth(seed=0)
    g = ds.g
    in_feats, n_classes = g.ndata['feat'].shape[1], ds.num_classes
    train_nid = np.arange(g.number_of_nodes(), dtype=np.int64)
    psize, batch_size = len(ds.par_li), 20
    S = world
    H, L = args.n_hidden, args.n_layers
    assert H % S == 0

    it = EngineClusterIter('reddit-synth', g, psize, batch_size, train_nid,
                           par_li=[p.copy() for p in ds.par_li], device=dev)
    first_epoch_order = [p.copy() for p in it.par_li]
    use_ln = True
    if S == 1:
        dims = dims_for(in_feats, H, n_classes, L)
        engine = SageEngine(dims, use_ln, args.dropout, it.n_max, dev, seed=seed)
        rs = np.random.RandomState(seed)
        for k, (i, o) in enumerate(dims):
            stdv = 1.0 / np.sqrt(2 * i)               # modules.py:213-216
            engine.arena.W[k].copy_(torch.from_numpy(
                rs.uniform(-stdv, stdv, (o, 2 * i)).astype(np.float32)))
            engine.arena.b[k].copy_(torch.from_numpy(rs.uniform(-stdv, stdv, o).astype(np.float32)))
        ist_model = None
    else:
        from gist_amd import ist
        ns = argparse.Namespace(num_subnet=S, n_hidden=H, n_layers=L, rank=rank,
                                dropout=args.dropout, use_layernorm=use_ln, lr=0.01,
                                weight_decay=0.0, iter_per_site=args.iter_per_site)
        base_init = None
        if rank == 0:
            rs = np.random.RandomState(seed)
            base_init = []
            for (i, o) in dims_for(in_feats, H, n_classes, L):
                stdv = 1.0 / np.sqrt(2 * i)
                base_init.append((rs.uniform(-stdv, stdv, (o, 2 * i)).astype(np.float32),
                                  rs.uniform(-stdv, stdv, o).astype(np.float32)))
        comm = None
        if shared_gpu:
            class HostStagedComm(ist.TorchDistComm):       # validation shim, see above
                def all_gather_flat(self, out, inp):
                    o = torch.empty(out.shape, dtype=out.dtype)
                    dist.all_gather_into_tensor(o, inp.cpu())
                    out.copy_(o)

                def broadcast(self, t, src=0):
                    c = t.cpu()
                    dist.broadcast(c, src=src)
                    t.copy_(c)
            comm = HostStagedComm()
        ist_model = ist.DistributedGNNWrapper(ns, None, in_feats, n_classes, dev,
                                              base_init=base_init, n_max=it.n_max, seed=seed,
                                              comm=comm)
        ist_model.ini_sync_dispatch_model()
        engine = ist_model.engine
        dims = ist_model.sub_dims
    it.bind(engine)
    lr = 0.01

    timed_ids = []          # id slices of the timed batches (replayed AFTER timing for nnz stats)
    n_log = []
    state = dict(total_iter=0, epoch=0)

    def batches():
        while True:
            for b in it:
                yield b
            state['epoch'] += 1

    gen = batches()

    def run_steps(count, log_from=None, sample_timer=None):
        for s in range(count):
            b = next(gen)
            ti = state['total_iter']
            if ist_model is not None and ti % args.iter_per_site == 0:
                if state['epoch'] > 0:                       # no re-dispatch in epoch 0 (:401-403)
                    ist_model.dispatch_model()
                ist_model.sub.reset_optimizer()              # fresh Adam (:405-407)
            if sample_timer is not None:    # HIP events around this step's SpMM/GEMM launches?
                engine.plan.timer = sample_timer if (s % args.timing_every == 0) else None
            engine.train_step(b, lr, 0.0)
            if log_from is not None:        # bookkeeping only: no device work in the timed region
                timed_ids.append(b.ids)
                n_log.append(b.n)
            state['total_iter'] = ti + 1
            if ist_model is not None and state['total_iter'] % args.iter_per_site == 0:
                ist_model.sync_model()                       # :422-427

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    run_steps(args.warmup)
    fence()
    timing = not args.no_kernel_timing
    native = engine.plan is not None
    sample_timer = None
    if timing:
        if native:          # HIP events recorded by the native step driver on the launch stream
            sample_timer = engine.enable_timer(
                (args.steps // max(args.timing_every, 1) + 1) * (12 * len(dims) + 4))
        else:
            hip.profile_begin()
    t0 = time.time()
    run_steps(args.steps, log_from=0, sample_timer=sample_timer)
    fence()
    elapsed = time.time() - t0
    prof = None
    if timing and native:
        rec = engine.read_timer()
        prof = {'gemm': [(ms, ('x', m, n, k)) for (ms, kind, m, n, k) in rec if kind == 1],
                'h3': [(ms, ('x', m, n, k)) for (ms, kind, m, n, k) in rec if kind == 2],
                'pre': [ms for (ms, kind, m, n, k) in rec if kind == 3],
                'spmm': [(ms, (m, n, k)) for (ms, kind, m, n, k) in rec if kind == 0]}
        engine.disable_timer()
    elif timing:
        prof = hip.profile_end()

    el = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if shared_gpu else dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    loss_val = float(engine.loss.item())

    # the same workload on the fp32 matrix-core path, same process (workspaces stay sized for
    # the split path, which is a superset)
    f32_rerun = None
    if world == 1 and args.gemm_mode == 'f16x3' and not args.no_f32_rerun:
        hip.gemm_mode('f32')
        engine.plan.timer = None
        n_re = max(args.steps // 3, 10)
        run_steps(3)
        fence()
        t1 = time.time()
        run_steps(n_re)
        fence()
        e1 = time.time() - t1
        f32_rerun = {'value': round(n_re / STEPS_PER_EPOCH / e1, 4), 'unit': 'epochs/s',
                     'ms_per_step': round(e1 / n_re * 1e3, 4), 'steps': n_re,
                     'gemm': 'every projection on v_mfma_f32_32x32x2_f32 (--gemm-mode f32)'}
        hip.gemm_mode('f16x3')

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        # every rank runs `steps` iterations of its own sub-GCN: S * steps / 75 epochs of work
        value = world * args.steps / STEPS_PER_EPOCH / elapsed
        out = {
            'metric': 'epochs/sec', 'value': round(value, 4), 'unit': 'epochs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32' if args.gemm_mode == 'f32' else
                     'f32 (projection products as 3 f16-split MFMA terms, fp32 accumulation)',
            'arithmetic_note': None if args.gemm_mode == 'f32' else
                     'storage, accumulation and every non-GEMM kernel are fp32; the large projections '
                     'split each fp32 operand into two f16 halves (22 bits) and accumulate ah.bh + '
                     'ah.bl + al.bh in fp32: error vs float64 <= the fp32-MFMA kernel\'s '
                     '(tests/test_gemm_h3_gpu.py); `f32_mfma` = same run with --gemm-mode f32',
            'data': 'synthetic',
            'config': {
                'workload': 'Reddit-like synthetic (N_train=153431, F=602, C=41, 1500 parts, '
                            'batch=20 parts, 75 steps/epoch); GraphSAGE n_hidden=%d n_layers=%d '
                            'LayerNorm dropout=%.2f Adam lr=0.01; %s' % (
                                H, L, args.dropout,
                                'full-width model on 1 GPU (cluster_gcn.py path)' if S == 1 else
                                'GIST %d sub-GCNs of width %d, sync every %d iters via one RCCL '
                                'all-gather (cluster_gcn_ist_distrib.py path)'
                                % (S, H // S, args.iter_per_site)),
                'n_hidden': H, 'n_layers': L, 'num_subnet': S, 'batch_parts': batch_size,
                'gemm_mode': args.gemm_mode,
                'psize': psize, 'steps_per_epoch': STEPS_PER_EPOCH,
                'epochs_per_sec_per_rank': round(value / world, 4),
            },
            'final_loss': round(loss_val, 5),
            **({'INVALID': 'GIST_BENCH_SHARED_GPU validation run: ranks share one GPU, host-staged gloo'} if shared_gpu else {}),
            'host_path': 'native step driver (gist_sage_step, 1 call/iteration)' if native else 'python op-by-op',
        }
        if prof is not None:
            # in-batch edge counts of the timed batches: re-extract them now, outside the timing
            nnz = np.zeros(args.steps, np.int64)
            for i, ids in enumerate(timed_ids):
                bb = it.batcher.extract(ids, engine.z0_left(ids.numel()))
                nnz[i] = int(bb.rowptr[bb.n].item())
            gem = prof['gemm']
            g_ms = sum(ms for ms, _ in gem) + sum(prof.get('pre', []))   # + split work outside the calls
            g_flop = sum(2.0 * m * n * k for _, (_, m, n, k) in gem)
            ach = g_flop / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
            every = args.timing_every if native else 1
            h3 = prof.get('h3', [])
            if h3:
                # dominant kernel: the split GEMM's main kernel, bracketed on its own.  achieved =
                # ALGORITHMIC flops (2mnk) / its time; it executes 3 f16 MFMA flops per
                # algorithmic flop, so its ceiling is the dense f16 peak / 3.
                h_ms = sum(ms for ms, _ in h3)
                h_flop = sum(2.0 * m * n * k for _, (_, m, n, k) in h3)
                h_ach = h_flop / (h_ms * 1e-3) / 1e12 if h_ms > 0 else 0.0
                h_traffic = None
                tf = os.path.join(ROOT, 'profiles', 'gemm_h3_traffic.json')
                if os.path.exists(tf):
                    try:
                        h_traffic = json.load(open(tf)).get('hbm_bytes_per_launch')
                    except Exception:
                        h_traffic = None
                out['roofline'] = {
                    'kernel': 'gist::gemm_h3_kernel (v_mfma_f32_16x16x32_f16, 3 MFMA flops per '
                              'algorithmic flop: ah.bh + ah.bl + al.bh)',
                    'bound': 'mfma', 'achieved': round(h_ach, 3), 'peak': MFMA_F16_PEAK_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': round(h_ach / MFMA_F16_PEAK_TFLOPS, 4),
                    'mfma_flops_per_algorithmic_flop': 3,
                    'mfma_rate_tflops': round(3 * h_ach, 1),
                    'frac_mfma_rate_of_peak': round(3 * h_ach / MFMA_F16_PEAK_TFLOPS, 4),
                    # fabric bytes per launch from PMC (profiles/gemm_h3_traffic.json; Infinity-Cache
                    # hits are counted): up to 8 XCDs x (8 + 8) operand panels, 2.6-3.6x the operand bytes
                    'traffic': h_traffic, 'launches': len(h3),
                    'sampled': 'every %d-th timed step' % args.timing_every if native else 'every step',
                    'avg_launch_ms': round(h_ms / max(len(h3), 1), 5),
                    'share_of_step': round(h_ms * every / (elapsed * 1e3), 4),
                    # every projection call of the step: split pre-pass + main kernel, and the
                    # class-layer GEMMs that stay on gist::gemm_f32_kernel
                    'all_projection_calls': {
                        'achieved': round(ach, 3), 'unit': 'TFLOP/s (algorithmic)',
                        'vs_f32_mfma_peak': round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                        'calls': len(gem), 'avg_call_ms': round(g_ms / max(len(gem), 1), 5),
                        'share_of_step': round(g_ms * every / (elapsed * 1e3), 4),
                        'split_prepass_share_of_step': round(
                            (sum(ms for ms, (_, m, n, k) in gem
                                 if any((m, n, k) == s[1:] for _, s in h3)) - h_ms
                             + sum(prof.get('pre', []))) * every / (elapsed * 1e3), 4),
                    },
                }
            else:
                out['roofline'] = {
                    'kernel': 'gist::gemm_f32_kernel (v_mfma_f32_32x32x2_f32; NT/NN/TN)',
                    'bound': 'mfma', 'achieved': round(ach, 3), 'peak': MFMA_F32_PEAK_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                    'traffic': None, 'launches': len(gem),
                    'sampled': 'every %d-th timed step' % args.timing_every if native else 'every step',
                    'avg_launch_ms': round(g_ms / max(len(gem), 1), 5),
                    'share_of_step': round(g_ms * every / (elapsed * 1e3), 4),
                }
            if f32_rerun is not None:
                out['f32_mfma'] = f32_rerun
            sp = prof['spmm']
            per_step = 2 * len(dims) - 1                 # SpMM launches per instrumented step
            stride = args.timing_every if native else 1  # which timed step a record belongs to
            s_ms = sum(ms for ms, _ in sp)
            s_bytes = 0.0
            for idx, (ms, (n, n_src, d)) in enumerate(sp):
                z = int(nnz[min((idx // per_step) * stride, args.steps - 1)])
                s_bytes += 4.0 * (n + 1) + 4.0 * z + 4.0 * n_src * d + 4.0 * n * d
            s_ach = s_bytes / (s_ms * 1e-3) / 1e9 if s_ms > 0 else 0.0
            traffic = None
            tf = os.path.join(ROOT, 'profiles', 'spmm_traffic.json')
            if os.path.exists(tf):
                try:
                    traffic = json.load(open(tf)).get('hbm_bytes_per_launch')
                except Exception:
                    traffic = None
            copy_gbs = measured_copy_gbs(dev)
            out['roofline_spmm'] = {
                'kernel': 'gist::spmm_csr_rowsplit_kernel', 'bound': 'hbm', 'achieved': round(s_ach, 2),
                'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(s_ach / HBM_PEAK_GBS, 4),
                # achievable HBM bandwidth on this box: a 1 GiB device-to-device copy (read+write)
                'peak_measured_copy': round(copy_gbs, 1),
                'frac_of_measured_copy': round(s_ach / copy_gbs, 4) if copy_gbs > 0 else None,
                'traffic': traffic, 'launches': len(sp),
                'avg_launch_ms': round(s_ms / max(len(sp), 1), 5),
                'avg_algorithmic_bytes': round(s_bytes / max(len(sp), 1), 1),
                'share_of_step': round(s_ms * (args.timing_every if native else 1) / (elapsed * 1e3), 4),
                'mean_batch_rows': round(float(np.mean(n_log)), 1),
                'mean_batch_nnz': round(float(nnz.mean()), 1),
            }
        if world == 1 and not args.no_cpu_baseline:
            try:
                step_s = cpu_baseline(ds, first_epoch_order, dims, use_ln, args.cpu_steps, seed)
                out['cpu_baseline'] = {
                    'value': round(1.0 / (STEPS_PER_EPOCH * step_s), 6), 'unit': 'epochs/s',
                    'cores': os.cpu_count(), 'kind': 'port',
                    'sample': 'oracle (numpy/OpenBLAS + OpenMP C SpMM) full training step on the '
                              'first %d batches of the same workload after 1 warm-up step, '
                              'median %.3f s/step, dropout off' % (args.cpu_steps, step_s),
                }
            except Exception as e:                          # report, never fake
                out['cpu_baseline'] = {'value': None, 'error': repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
