#!/usr/bin/env python3
"""bench.py -- epochs/sec of GIST's training hot path on MI355X, Reddit-like synthetic data.

    python bench.py --gpus N --steps K --warmup W
        N = 1: runs in this process.  N > 1 without WORLD_SIZE in the environment: this
        process starts N rank processes itself (one per GPU, RCCL) BEFORE it touches the GPU,
        waits for them and exits non-zero if any of them fails.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
        the same rank code, launched by torchrun (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).

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

Arithmetic of the headline `value` (--gemm-mode bf16x3, the default): fp32 storage and fp32
accumulation everywhere; the large projections carry every fp32 operand as three bf16 pieces
(3 x 8 = all 24 significant bits, exact, no scaling) and accumulate the six leading cross terms in
fp32 on v_mfma_f32_16x16x32_bf16 -- what is dropped is below one fp32 rounding of a product, and the
error against float64 is at the fp32-MFMA kernel's level on every operand class including
cancellation, 2^40 in-row range and exponent extremes (tests/test_gemm_b3_gpu.py); the golden training
run is repeated forced onto this kernel (tests/test_e2e_gpu.py).  Small projections (class layer,
per-rank widths below the tile threshold) run on v_mfma_f32_32x32x2_f32.  At N=1 the same process
then re-times the workload with every projection on the fp32 MFMA (leg `f32_mfma`, --gemm-mode f32)
and with the 3-term f16 split (leg `f16x3_split`: 22 of 24 operand bits, narrower than fp32 and
therefore never `value`), each with its own roofline.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: the projection GEMM),
`roofline_spmm` (the SpMM against HBM), `cpu_baseline` (the oracle timed on host cores, all
cores and one thread).
"""
import argparse
import gc
import json
import os
import random
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: Peak FP32 (matrix)
MFMA_F16_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: Peak BF16/FP16 MFMA, dense
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E peak BW (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=150)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', type=int, choices=[2, 3, 4, 5], default=None,
                    help='BASELINE.json config preset (SURVEY.md section 8d).  2: Reddit-like hidden=256 L=4 on '
                         '1 GPU (script/reddit/run_baseline_reddit.sh:6-8).  3: the default workload (Reddit-like '
                         'H=4096 L=2, S = --gpus).  4: Amazon-like F=100 C=47, 15000 parts, batch 10, L=4, sync '
                         'every 500 (script/amazon/run_ist_distrib_amazon.sh:16-18); H=4096 over 8 GPUs, or with '
                         '--gpus 1 the per-rank sub-GCN (width 512) of that run.  5: Reddit-like ultra-wide '
                         'H=32768 over 8 GPUs, or with --gpus 1 the per-rank sub-GCN (width 4096) plus the '
                         'H=32768 S=8 weight exchange measured with 8 base replicas on this GPU.  Explicit '
                         '--n-hidden / --n-layers / --dataset / --batch-parts / --iter-per-site override a preset')
    ap.add_argument('--dataset', choices=['reddit-synth', 'amazon-synth', 'reddit-communities'], default=None,
                    help='reddit-communities: Reddit-sized graph with power-law communities (30-400 nodes, mixing 0.3, random node '
                         'ids): NO planted parts -- implies --partition own')
    ap.add_argument('--partition', choices=['planted', 'own'], default=None,
                    help='planted: the block model\'s own blocks stand in for the METIS parts (default on the block models); own: '
                         'the parts come from gist_partition_graph (the reference\'s cache-miss path, cluster_gcn/sampler.py:49-51)')
    ap.add_argument('--psize', type=int, default=None, help='number of parts with --partition own (default: as many as planted / 1500)')
    ap.add_argument('--batch-parts', type=int, default=None, help='METIS parts per cluster batch')
    ap.add_argument('--n-hidden', type=int, default=None)
    ap.add_argument('--n-layers', type=int, default=None)
    ap.add_argument('--dropout', type=float, default=0.2)
    ap.add_argument('--iter-per-site', type=int, default=None)
    ap.add_argument('--gemm-mode', choices=['f32', 'f16x3', 'bf16x3'], default='bf16x3',
                    help='products of the large projections behind `value`: three bf16 pieces per fp32 '
                         'operand = all 24 bits, six cross terms (default), v_mfma_f32_32x32x2_f32, or the '
                         '3-term f16 split (22 bits)')
    ap.add_argument('--host-path', choices=['engine', 'module'], default='engine',
                    help='engine: one gist_sage_step call per iteration (SageEngine.train_step).  module: the reference\'s '
                         'own loop body -- pred = model(cluster); loss = loss_f(pred[mask], labels[mask]); '
                         'optimizer.zero_grad(); loss.backward(); optimizer.step() (cluster_gcn/cluster_gcn.py:96-105) -- on '
                         'gist_amd.modules.GCN / nn.CrossEntropyLoss / optim.Adam / sampler.ClusterIter (N = 1 only)')
    ap.add_argument('--no-module-leg', action='store_true',
                    help='skip the module_path leg of the default N = 1 line')
    ap.add_argument('--no-unplanted-leg', action='store_true',
                    help='skip the unplanted_graph leg of the default N = 1 line (the same model on the power-law community '
                         'graph cut by gist_partition_graph, in a child process)')
    ap.add_argument('--no-second-leg', action='store_true',
                    help='N=1: skip re-timing the workload in the other GEMM mode')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=0,
                    help='oracle steps timed on all host cores after one warm-up step (0 = one full epoch); the '
                         '1-thread figure times 12')
    ap.add_argument('--cpu-budget', type=float, default=90.0,
                    help='wall-time bound (s) of the all-cores oracle sample; the 1-thread sample gets half')
    ap.add_argument('--tune', action='append', default=[], metavar='KNOB=VALUE',
                    help='development A/B: set a tuning hook of the library (gist_amd._lib.TUNE) before the run; recorded in '
                         'the line as config.tune')
    ap.add_argument('--no-kernel-timing', action='store_true',
                    help='skip HIP-event bracketing of SpMM/GEMM launches')
    ap.add_argument('--timing-every', type=int, default=16,
                    help='bracket the SpMM/GEMM launches of every N-th timed step with HIP events '
                         '(an event pair per launch serialises kernel boundaries: ~80 us/step '
                         'when every step is instrumented, i.e. it would depress `value`)')
    args = ap.parse_args()
    # presets: the default invocation (no --config) is the metric's workload, unchanged
    n = args.gpus
    preset = {
        None: dict(dataset='reddit-synth', batch_parts=20, n_hidden=4096, n_layers=2, iter_per_site=100),
        3: dict(dataset='reddit-synth', batch_parts=20, n_hidden=4096, n_layers=2, iter_per_site=100),
        2: dict(dataset='reddit-synth', batch_parts=20, n_hidden=256, n_layers=4, iter_per_site=100),
        4: dict(dataset='amazon-synth', batch_parts=10, n_hidden=4096 if n > 1 else 4096 // 8, n_layers=4,
                iter_per_site=500),
        5: dict(dataset='reddit-synth', batch_parts=20, n_hidden=32768 if n > 1 else 32768 // 8, n_layers=2,
                iter_per_site=100),
    }[args.config]
    args.emulated_rank_of = None          # N=1 run of one rank's sub-GCN of an 8-GPU config
    if args.config in (4, 5) and n == 1 and args.n_hidden is None:
        args.emulated_rank_of = 8
    for k, v in preset.items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    if args.config == 2 and n != 1:
        ap.error('--config 2 is the single-GPU baseline (cluster_gcn.py): use --gpus 1')
    if args.partition is None:
        args.partition = 'own' if args.dataset == 'reddit-communities' else 'planted'
    if args.dataset == 'reddit-communities' and args.partition != 'own':
        ap.error('reddit-communities has no planted k-way partition: --partition own')
    return args


# --------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the rank processes from here (before any GPU call)
# --------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, deadline_s=None):
    """Parent of a self-launched N-rank run: never imports torch, never touches the GPU.  One
    child per rank with the torchrun environment contract; children share stdout/stderr (rank
    0 prints the JSON line).  Any child failing -> the others are terminated (SIGTERM, then SIGKILL
    after a grace period), exit code != 0.  SIGTERM / SIGINT / SIGHUP to the parent, an exception in
    it, or the overall deadline (GIST_BENCH_DEADLINE_S, default 3 h) take the children down the
    same way: no rank process outlives the launcher."""
    import signal
    port = os.environ.get('MASTER_PORT') or str(_free_port())
    if deadline_s is None:
        deadline_s = float(os.environ.get('GIST_BENCH_DEADLINE_S', '10800'))
    procs = []

    def stop_all(grace=5.0):
        live = [p for p in procs if p.poll() is None]
        for p in live:
            try:
                p.terminate()
            except OSError:
                pass
        t_end = time.time() + grace
        for p in live:
            try:
                p.wait(timeout=max(0.0, t_end - time.time()))
            except subprocess.TimeoutExpired:
                pass
        for p in live:
            if p.poll() is None:
                try:
                    p.kill()
                except OSError:
                    pass
        for p in live:
            try:
                p.wait(timeout=5.0)
            except subprocess.TimeoutExpired:
                pass

    class _Signalled(Exception):
        pass

    def on_signal(signum, frame):
        raise _Signalled(signum)

    previous = {}
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            previous[sig] = signal.signal(sig, on_signal)
        except (ValueError, OSError):        # not the main thread
            pass

    def die_with_parent():
        # the kernel delivers SIGTERM to the child if the launcher dies without running its handlers
        try:
            import ctypes
            ctypes.CDLL(None).prctl(1, signal.SIGTERM)      # PR_SET_PDEATHSIG
        except Exception:
            pass

    rc = 0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                       LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=port,
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
                       NCCL_DEBUG=os.environ.get('NCCL_DEBUG', 'WARN'))
            # GIST_BENCH_RANK_CMD (tests only): the rank program, so that the launcher's handling of
            # hung / signal-ignoring ranks can be exercised without a GPU
            cmd = ([sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
                   if not os.environ.get('GIST_BENCH_RANK_CMD')
                   else [sys.executable, '-c', os.environ['GIST_BENCH_RANK_CMD']])
            procs.append(subprocess.Popen(cmd, env=env, preexec_fn=die_with_parent))
        t_dead = time.time() + deadline_s
        alive = list(procs)
        while alive:
            time.sleep(0.2)
            failed = None
            for p in alive:
                code = p.poll()
                if code is not None and code != 0:
                    failed = (p, code)
                    break
            if failed is not None:
                # the FIRST failing rank decides the launcher's exit code; the others are stopped
                p, code = failed
                rc = code if code > 0 else 128 - code          # killed by signal s: 128 + s
                print('bench: rank process %d exited with %d; stopping the others'
                      % (procs.index(p), code), file=sys.stderr, flush=True)
                stop_all()
                break
            alive = [p for p in alive if p.poll() is None]
            if alive and time.time() > t_dead:
                print('bench: deadline of %.0f s passed with %d rank processes still running; stopping them'
                      % (deadline_s, len(alive)), file=sys.stderr, flush=True)
                rc = 124
                stop_all()
                break
    except _Signalled as e:
        print('bench: signal %d; stopping the rank processes' % e.args[0], file=sys.stderr, flush=True)
        rc = 128 + int(e.args[0])
    except BaseException:
        rc = rc or 1
        raise
    finally:
        stop_all()
        for sig, h in previous.items():
            try:
                signal.signal(sig, h)
            except (ValueError, OSError):
                pass
    return rc


# --------------------------------------------------------------------------------------------
# CPU baseline (the oracle) -- the only place this file touches oracle/
# --------------------------------------------------------------------------------------------
def cpu_baseline(ds, par_order, dims, use_layernorm, p_drop, n_steps, seed, threads, batch_parts=20,
                 budget_s=None):
    """The oracle (oracle/gist_oracle.py, numpy + OpenBLAS + OpenMP C SpMM) on `threads` host
    threads, same workload: the batches of the first epoch in order, full step each (extraction,
    forward with dropout masks, CE, backward, Adam) -- the reference's timing definition
    (cluster_gcn.py:91,106-108: the epoch loop's wall time, evaluation excluded).  One untimed
    warm-up step, then up to n_steps steps; `budget_s` bounds the timed steps' wall time (a slow host
    must not turn the default bench run into a quarter of an hour).
    Returns (median step s, steps timed, wall s of the timed steps, thread pools)."""
    from threadpoolctl import threadpool_limits
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
    n_batches = len(par_order) // batch_parts
    with threadpool_limits(limits=threads):
        from threadpoolctl import threadpool_info
        pools = sorted('%s:%d' % (p_.get('internal_api', '?'), p_.get('num_threads', 0))
                       for p_ in threadpool_info())
        for j in range(n_steps + 1):
            jb = j % n_batches
            ids = np.concatenate(par_order[jb * batch_parts:(jb + 1) * batch_parts]).astype(np.int64)
            t0 = time.time()
            rpb, clb, trp, tcl, x, y = tg.batch(ids)
            masks = None
            if p_drop > 0:      # nn.Dropout on the concatenated [h | ah] of every layer
                masks = [(rs.random_sample((len(ids), 2 * i)) >= p_drop).astype(np.float32)
                         for (i, o) in dims]
            O.train_step(rpb, clb, trp, tcl, x, y, params, opt, use_layernorm, 0.01,
                         drop_masks=masks, drop_p=p_drop)
            if j > 0:                                   # step 0 = warm-up
                times.append(time.time() - t0)
                if budget_s is not None and sum(times) > budget_s:
                    break
    return float(np.median(times)), len(times), float(sum(times)), pools


def ultra_wide_exchange(dev, in_feats, n_classes, H, S, L, iter_per_site, reps=3):
    """BASELINE config 5's weight exchange at FULL size on one GPU: S base replicas of the H-wide
    model (8.8 GB each at H=32768) and S sub-GCNs in this process (ist.LocalCommGroup), HIP-event
    times of ONE rank's device work per exchange: scatter of the S gathered sub arenas into its base
    replica (sync_model minus the collective, cluster_gcn_ist_ultra_wide.py:143-248 keeps the base on
    the host and stages every slice over PCIe) and the local gather of its next sub-model
    (dispatch_model).  The all-gather itself needs S GPUs: its bytes are reported, its time is not."""
    import argparse as _ap
    import torch
    from gist_amd import ist
    from gist_amd.engine import dims_for
    group = ist.LocalCommGroup(S)
    gen = torch.Generator(device=dev).manual_seed(0)
    base_init = [((torch.rand(o, 2 * i, device=dev, generator=gen) * 2 - 1) / np.sqrt(2 * i),
                  (torch.rand(o, device=dev, generator=gen) * 2 - 1) / np.sqrt(2 * i))
                 for (i, o) in dims_for(in_feats, H, n_classes, L)]
    models = []
    for r in range(S):
        ns = _ap.Namespace(num_subnet=S, n_hidden=H, n_layers=L, rank=r, dropout=0.0, use_layernorm=True)
        models.append(ist.DistributedGNNWrapper(ns, None, in_feats, n_classes, dev,
                                                base_init=base_init if r == 0 else None,
                                                comm=group.handle(r)))
    del base_init
    rstate = random.getstate()
    part = models[0].sample_partitions()
    for m in models:
        m.ini_sync_dispatch_model(part)
    m0 = models[0]
    t_gather, t_apply, t_disp = [], [], []
    for _ in range(reps + 1):
        part = models[0].sample_partitions()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for m in models[1:]:
            m.sync_gather()
        ev[0].record()
        m0.sync_gather()                 # S device copies of the sub arenas (stand-in for the all-gather)
        ev[1].record()
        m0.sync_apply()
        ev[2].record()
        m0.dispatch_model(part)
        ev[3].record()
        for m in models[1:]:
            m.sync_apply()
            m.dispatch_model(part)
        torch.cuda.synchronize(dev)
        t_gather.append(ev[0].elapsed_time(ev[1]))
        t_apply.append(ev[1].elapsed_time(ev[2]))
        t_disp.append(ev[2].elapsed_time(ev[3]))
    random.setstate(rstate)
    P = m0.sub.numel
    med = lambda v: float(np.median(v[1:]))
    out = {
        'n_hidden': H, 'num_subnet': S, 'n_layers': L,
        'base_replica_bytes': int(4 * m0.base.numel), 'sub_arena_bytes': int(4 * P),
        'sync_scatter_ms': round(med(t_apply), 4), 'dispatch_gather_ms': round(med(t_disp), 4),
        'local_copy_of_gathered_arenas_ms': round(med(t_gather), 4),
        'all_gather_bytes_per_rank': int(4 * P), 'all_gather_bytes_total': int(4 * P * S),
        'amortized_device_ms_per_step': round((med(t_apply) + med(t_disp)) / iter_per_site, 5),
        'measured': 'median of %d exchanges, HIP events, one rank\'s device work with %d base replicas + %d sub-GCNs '
                    'resident on this GPU; the RCCL all-gather over xGMI is NOT measured here (needs %d GPUs)'
                    % (reps, S, S, S),
    }
    del models, group
    torch.cuda.empty_cache()
    return out


def host_cores():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a
    GPU box hands a 1-GPU job a share of the host, not all of os.cpu_count())."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


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


def _traffic(name):
    """HBM/fabric bytes per launch from a committed PMC profile (NOT measured in this run)."""
    tf = os.path.join(ROOT, 'profiles', name)
    if not os.path.exists(tf):
        return None, None
    try:
        return json.load(open(tf)).get('hbm_bytes_per_launch'), 'profiles/' + name + \
            ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; not measured in this run)'
    except Exception:
        return None, None


# --------------------------------------------------------------------------------------------
# first-contact hardening of the N > 1 path (cluster_gcn_ist_distrib.py:577-584 is the reference's
# init_process_group + spawn; it has no time-outs and no device checks)
# --------------------------------------------------------------------------------------------
def fail_line(reason, code, **extra):
    """A rank that cannot run prints ONE JSON error line (stdout, so that a harness that scrapes the
    result line sees WHY there is no value) and exits non-zero.  os._exit: a rank stuck inside a
    collective cannot be unwound, and nothing here re-executes a process that touched the GPU."""
    out = {'metric': 'epochs/sec', 'value': None, 'unit': 'epochs/s', 'error': reason,
           'rank': int(os.environ.get('RANK', '0')), 'world_size': int(os.environ.get('WORLD_SIZE', '1'))}
    out.update(extra)
    # (stdout carries ONE line per run: rank 0's; the other ranks report on stderr)
    print(json.dumps(out), file=sys.stdout if out['rank'] == 0 else sys.stderr, flush=True)
    print('bench: rank %s: %s' % (os.environ.get('RANK', '0'), reason), file=sys.stderr, flush=True)
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(code)


class Watchdog(object):
    """Per-rank deadline per phase (rendez-vous, set-up, the run): a daemon thread that ends the process
    with exit code 124 and a JSON error line when a phase overruns -- e.g. a peer that never joins, or
    an RCCL collective that never returns.  The launcher (ours or torchrun) then stops the other ranks."""

    def __init__(self):
        import threading
        self._lock = threading.Lock()
        self._phase, self._deadline = None, None
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def phase(self, name, seconds):
        with self._lock:
            self._phase = name
            self._deadline = None if seconds is None else time.time() + float(seconds)

    def _run(self):
        while True:
            time.sleep(0.5)
            with self._lock:
                name, dl = self._phase, self._deadline
            if dl is not None and time.time() > dl:
                fail_line('watchdog: phase %r exceeded its deadline' % name, 124, phase=name)


def device_identity(torch, dev):
    """What tells two GPUs apart: UUID and PCI location of the device this rank computes on."""
    p = torch.cuda.get_device_properties(dev)
    ident = {}
    for k in ('uuid', 'pci_domain_id', 'pci_bus_id', 'pci_device_id'):
        v = getattr(p, k, None)
        if v is not None:
            ident[k] = str(v)
    return ident


def shared_dataset(name, local_rank, world, wait_s=600.0):
    """One synthetic graph per NODE, not per rank: local rank 0 builds it (8-25 s of host work, 0.5-1.5 GB)
    and writes the arrays to /dev/shm BEFORE any rank touches the GPU; the other ranks load them.  The
    reference makes every rank load and upload the full graph itself.  Returns
    (dataset, seconds, 'built' | 'loaded' | 'built, N=1')."""
    from gist_amd import datasets
    t0 = time.time()
    build = {'reddit-synth': lambda: datasets.reddit_synth(seed=0), 'amazon-synth': lambda: datasets.amazon_synth(seed=1),
             'reddit-communities': lambda: datasets.reddit_communities(seed=0)}[name]
    if world == 1:
        return build(), time.time() - t0, 'built, N=1'
    tag = _shared_tag(name)
    root = os.environ.get('GIST_BENCH_SHM', '/dev/shm')
    d = os.path.join(root, tag)
    done = os.path.join(d, 'DONE')
    if local_rank == 0:
        _drop_stale_shared(root)
        ds = build()
        tmp = d + '.tmp%d' % os.getpid()
        os.makedirs(tmp, exist_ok=True)
        datasets.save_arrays(ds, tmp)
        if os.path.isdir(d):
            import shutil
            shutil.rmtree(d, ignore_errors=True)
        os.rename(tmp, d)
        open(done, 'w').write('ok')
        return ds, time.time() - t0, 'built'
    while not os.path.exists(done):
        if time.time() - t0 > wait_s:
            fail_line('dataset cache %s not written by local rank 0 within %.0f s' % (d, wait_s), 5)
        time.sleep(0.2)
    return datasets.load_arrays(d), time.time() - t0, 'loaded'


def _shared_tag(name):
    # one directory per LAUNCH: the ranks of a launch share MASTER_PORT and their parent (the launcher / the elastic
    # agent), so a directory left behind by a run that died (same port, DONE present) is never mistaken for this one's
    return 'gist_bench_%s_%s_%d' % (name, os.environ.get('MASTER_PORT', '0'), os.getppid())


def _drop_stale_shared(root, older_than_s=1800.0):
    """Directories of launches that never got to drop_shared_dataset hold 0.5-1.5 GB of memory each."""
    import shutil
    try:
        for fn in os.listdir(root):
            path = os.path.join(root, fn)
            if fn.startswith('gist_bench_') and os.path.isdir(path) and os.stat(path).st_uid == os.getuid() \
                    and time.time() - os.stat(path).st_mtime > older_than_s:
                shutil.rmtree(path, ignore_errors=True)
    except OSError:
        pass


def drop_shared_dataset(name):
    import shutil
    shutil.rmtree(os.path.join(os.environ.get('GIST_BENCH_SHM', '/dev/shm'), _shared_tag(name)), ignore_errors=True)


def main():
    args = parse()
    # Host hygiene (profiles/r05_module_path.md): the OpenMP workers of any multi-threaded CPU tensor operation of the set-up
    # (parameter initialisation, a pinned copy) spin-wait afterwards by default; inside a container with a CPU quota that burns
    # the period's quota and the kernel freezes the whole process for tens of ms -- in the middle of a timed region.  Passive
    # waiting, before torch loads its OpenMP runtime.
    # (and this process is the application: the library leaves the cyclic collector alone unless told, sampler.freeze_setup_objects)
    for k_, v_ in (('OMP_WAIT_POLICY', 'PASSIVE'), ('GOMP_SPINCOUNT', '0'), ('KMP_BLOCKTIME', '0'), ('GIST_GC_FREEZE', '1')):
        os.environ.setdefault(k_, v_)
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

    import datetime
    import torch
    import torch.distributed as dist
    t_start = time.time()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    # GIST_BENCH_SHARED_GPU=1 (validation only, never a reported number): every rank uses
    # cuda:0 and the collectives are staged through the host over gloo, so the N>1 logic can
    # be exercised on a 1-GPU box.  The product path is RCCL (backend "nccl").
    shared_gpu = os.environ.get('GIST_BENCH_SHARED_GPU') == '1'
    # one GPU per rank, or no run: device_count() does not initialise the GPU
    n_dev = torch.cuda.device_count()
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
    if not shared_gpu and (n_dev < local_world or local_rank >= n_dev):
        fail_line('%d rank processes on this node but torch.cuda.device_count() = %d: one MI355X per rank is '
                  'required (RCCL refuses two ranks on one device)' % (local_world, n_dev), 2,
                  devices_visible=n_dev)
    wd = Watchdog()
    t_rdv = float(os.environ.get('GIST_BENCH_RENDEZVOUS_TIMEOUT_S', '300'))
    os.environ.setdefault('NCCL_DEBUG', 'WARN')         # RCCL warnings go to this rank's stderr
    # -- data set-up on the host, before anything touches the GPU --------------------------------
    wd.phase('dataset', float(os.environ.get('GIST_BENCH_SETUP_TIMEOUT_S', '900')))
    seed = 0
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    ds, t_dataset, dataset_how = shared_dataset(args.dataset, local_rank, world)
    # -- rendez-vous -----------------------------------------------------------------------------
    dev = torch.device('cuda', 0 if shared_gpu else local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        wd.phase('rendezvous', t_rdv + 30)
        pg_timeout = datetime.timedelta(seconds=t_rdv)
        try:
            if shared_gpu:
                dist.init_process_group(backend='gloo', rank=rank, world_size=world, timeout=pg_timeout)
                torch.cuda.set_device(dev)
            else:
                torch.cuda.set_device(dev)
                dist.init_process_group(backend='nccl', rank=rank, world_size=world, device_id=dev,
                                        timeout=pg_timeout)
        except Exception as e:
            fail_line('rendezvous failed within %.0f s: %r' % (t_rdv, e), 3, phase='rendezvous')
    else:
        torch.cuda.set_device(dev)
    # which physical device does each rank compute on?  (rccl_ranks alone would not show two ranks on one GPU)
    wd.phase('device census', t_rdv + 30)
    ident = device_identity(torch, dev)
    idents = [ident]
    if world > 1:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
    # (every rank has loaded its copy of the node's dataset by now -- the all-gather above is the first collective after
    # the load: the /dev/shm directory can go, instead of holding 0.5-1.5 GB until the end of a run that may not get there)
    if world > 1 and local_rank == 0:
        drop_shared_dataset(args.dataset)
    keys = [json.dumps(i, sort_keys=True) for i in idents]
    devices_distinct = len(set(keys)) == world and all(i for i in idents)
    if world > 1 and not shared_gpu and not devices_distinct:
        if rank == 0:
            fail_line('ranks do not sit on distinct GPUs', 4, rank_devices=idents)
        os._exit(4)
    wd.phase('setup', float(os.environ.get('GIST_BENCH_SETUP_TIMEOUT_S', '900')))

    from gist_amd import datasets, hip
    from gist_amd.engine import SageEngine, dims_for
    from gist_amd.sampler import EngineClusterIter
    for kv in args.tune:
        knob, _, val = kv.partition('=')
        hip.tuning(knob, float(val))

    # workspaces are sized for the split path (a superset of what mode f32 needs), then the
    # requested mode is selected for the headline run
    second_leg = world == 1 and not args.no_second_leg
    # (sized in the mode with the largest workspaces among the ones this run will use)
    hip.gemm_mode('bf16x3' if (second_leg or args.gemm_mode == 'bf16x3') else args.gemm_mode)
    g = ds.g
    in_feats, n_classes = g.ndata['feat'].shape[1], ds.num_classes
    train_nid = np.arange(g.number_of_nodes(), dtype=np.int64)
    partition_info = None
    if args.partition == 'own':
        # the parts the training loop runs on come from the library's partitioner, like a cache miss of the reference
        # (every rank computes the same parts: the partitioner is a function of (graph, k, seed))
        from gist_amd.dgl_compat.transform import partition_assignment
        k_parts = args.psize or (1500 if args.dataset != 'amazon-synth' else 15000)
        t_p = time.time()
        assign = partition_assignment(g, k_parts, seed=0)
        t_p = time.time() - t_p
        order_ = np.argsort(assign, kind='stable')
        bounds_ = np.searchsorted(assign[order_], np.arange(k_parts + 1))
        own_parts = [order_[bounds_[i]:bounds_[i + 1]].astype(np.int64) for i in range(k_parts)]
        sizes_ = np.diff(bounds_)
        planted_sizes = np.array([len(p) for p in ds.par_li])
        ds = ds._replace(par_li=own_parts)
        partition_info = {'partitioner': 'gist_partition_graph (multilevel, host)', 'parts': k_parts, 'seconds': round(t_p, 2),
                          'part_size_histogram': {'min': int(sizes_.min()), 'p05': int(np.percentile(sizes_, 5)),
                                                  'median': int(np.median(sizes_)), 'p95': int(np.percentile(sizes_, 95)),
                                                  'max': int(sizes_.max())},
                          'ground_truth_groups': {'count': int(planted_sizes.size), 'size_min': int(planted_sizes.min()),
                                                  'size_median': int(np.median(planted_sizes)), 'size_max': int(planted_sizes.max())}}
    psize, batch_size = len(ds.par_li), args.batch_parts
    STEPS_PER_EPOCH = psize // batch_size                      # sampler.py:54
    S = world
    H, L = args.n_hidden, args.n_layers
    assert H % S == 0

    it = EngineClusterIter(ds.name, g, psize, batch_size, train_nid,
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
        comm = None                                              # default: TorchDistComm (RCCL)
        if shared_gpu:                                           # validation only: test infrastructure
            from tests.host_staged_comm import HostStagedComm
            comm = HostStagedComm()
        ist_model = ist.DistributedGNNWrapper(ns, None, in_feats, n_classes, dev,
                                              base_init=base_init, n_max=it.n_max, seed=seed,
                                              comm=comm)
        ist_model.ini_sync_dispatch_model()
        engine = ist_model.engine
        dims = ist_model.sub_dims
    it.bind(engine)                  # (the kept-split workspace is sized for either split mode)
    # the timed loop only reads the loss: let a step's optimiser launch extract the next batch of the epoch as well
    engine.prefetch = os.environ.get('GIST_BENCH_PREFETCH', '1') != '0'
    if os.environ.get('GIST_BENCH_TUNE_CLASS_FUSED'):       # (A/B of the class layer's launch forms: include/gist_hip.h)
        hip.tuning('class_fused', int(os.environ['GIST_BENCH_TUNE_CLASS_FUSED']))
    hip.gemm_mode(args.gemm_mode)
    lr = 0.01
    native = engine.plan is not None
    # HIP events around every SpMM / GEMM launch of a step cost that step ~80 us (the pairs serialise kernel boundaries):
    # at most FIVE instrumented steps per timed region, at least --timing-every apart -- 0.2 % of `value` at H = 4096, under
    # 1 % at the 0.3-ms steps of the narrow widths (10 instrumented steps of 150 were 1.8 % there)
    every = max(args.timing_every, -(-args.steps // 5), 1) if native else 1

    state = dict(total_iter=0, epoch=0)
    sync_ms = []            # HIP-event time of every sync (+ re-dispatch) inside a timed region

    # ---- the module path: the reference's loop body, statement for statement (cluster_gcn/cluster_gcn.py:89-105) ----
    class ModuleLoop(object):
        """model(cluster) / loss_f / zero_grad / backward / step on gist_amd's drop-in classes.  Same graph, parts,
        batch size, model shape, initial parameters and dropout seed as the engine path's headline."""

        def __init__(self):
            import torch.nn.functional as F
            from gist_amd.modules import GCN
            from gist_amd.nn import CrossEntropyLoss
            from gist_amd.optim import Adam
            from gist_amd.sampler import ClusterIter
            rstate = random.getstate()
            self.cluster_iterator = ClusterIter(ds.name, g, psize, batch_size, train_nid, use_pp=False,
                                                par_li=[p.copy() for p in ds.par_li], device=dev)
            random.setstate(rstate)
            self.model = GCN(in_feats, H, n_classes, L, F.relu, args.dropout, use_ln, False, False, 1, True)
            rs_ = np.random.RandomState(seed)
            for layer, (i, o) in zip(self.model.layers, dims):
                stdv = 1.0 / np.sqrt(2 * i)
                layer.linear.weight.data.copy_(torch.from_numpy(rs_.uniform(-stdv, stdv, (o, 2 * i)).astype(np.float32)))
                layer.linear.bias.data.copy_(torch.from_numpy(rs_.uniform(-stdv, stdv, o).astype(np.float32)))
            self.model.cuda()
            self.model.set_dropout_seed(seed)
            self.loss_f = CrossEntropyLoss()
            self.optimizer = Adam(self.model.parameters(), lr=lr, weight_decay=0.0)
            self.gen = self._batches()
            self.engine = None

        def _batches(self):
            while True:
                for cluster in self.cluster_iterator:
                    yield cluster

        breakdown = None

        def _run_timed(self, cluster):
            pc = time.perf_counter
            t = [pc()]
            cluster = cluster.to(torch.cuda.current_device()); self.model.train(); t.append(pc())
            pred = self.model(cluster); t.append(pc())
            if t[-1] - t[-2] > 2e-3 and self.engine is not None:
                self.breakdown['forward_native_call_ms'] = round(self._native_ms[-1], 3)
            batch_labels = cluster.ndata['label']; batch_train_mask = cluster.ndata['train_mask']; t.append(pc())
            loss = self.loss_f(pred[batch_train_mask], batch_labels[batch_train_mask]); t.append(pc())
            self.optimizer.zero_grad(); t.append(pc())
            loss.backward(); t.append(pc())
            self.optimizer.step(); t.append(pc())
            d = [round((b - a) * 1e3, 3) for a, b in zip(t[:-1], t[1:])]
            if sum(d) > self.breakdown.get('total', 0.0):
                self.breakdown.update(total=sum(d), parts=dict(zip(('to+train', 'forward', 'ndata', 'loss', 'zero_grad',
                                                                    'backward', 'step'), d)))

        def run(self, count, sample_timer=None, ids_log=None, n_log=None, loss_log=None, every_=1, stamps=None):
            model, loss_f, optimizer = self.model, self.loss_f, self.optimizer
            for s in range(count):
                if stamps is not None:
                    stamps.append(time.perf_counter())
                cluster = next(self.gen)
                if self.engine is not None:
                    self.engine.plan.timer = sample_timer if (sample_timer is not None and s % every_ == every_ // 2) else None
                if self.breakdown is not None:          # (dev: where a slow iteration spends its host time)
                    self._run_timed(cluster)
                    continue
                cluster = cluster.to(torch.cuda.current_device())
                model.train()
                pred = model(cluster)
                batch_labels = cluster.ndata['label']
                batch_train_mask = cluster.ndata['train_mask']
                loss = loss_f(pred[batch_train_mask], batch_labels[batch_train_mask])
                optimizer.zero_grad()
                loss.backward()
                optimizer.step()
                if self.engine is None:
                    mes = getattr(model, '_module_engines', None)
                    me = list(mes.values())[0] if mes else None
                    if not me:
                        raise RuntimeError('bench.py --host-path module: the model did not bind to the step plan')
                    self.engine = me.engine
                if ids_log is not None:
                    ids_log.append(cluster._ids)
                    n_log.append(cluster._n)
                if loss_log is not None and (s == 0 or s == count - 1):
                    loss_log.append(loss.detach().clone())

    module_headline = args.host_path == 'module'
    if module_headline and (world != 1 or ist_model is not None):
        fail_line('--host-path module is the N = 1 loop (cluster_gcn.py)', 2)
        os._exit(2)
    mloop = None
    if module_headline:
        mloop = ModuleLoop()
        mloop.run(1)                                   # (binds the model: its engine carries the timers from here on)
        torch.cuda.synchronize(dev)
        engine = mloop.engine
        it = mloop.cluster_iterator
        native = True

    def batches():
        while True:
            for b in it:
                yield b
            state['epoch'] += 1

    gen = batches()

    host_stamps = []        # host clock at the start of every step of the LAST timed region (issue gaps: diagnostics)

    def run_steps(count, sample_timer=None, ids_log=None, n_log=None, loss_log=None):
        del host_stamps[:]
        if module_headline:
            return mloop.run(count, sample_timer, ids_log, n_log, loss_log, every, stamps=host_stamps)
        for s in range(count):
            host_stamps.append(time.perf_counter())
            b = next(gen)
            ti = state['total_iter']
            if ist_model is not None and ti % args.iter_per_site == 0:
                if state['epoch'] > 0:                       # no re-dispatch in epoch 0 (:401-403)
                    ist_model.dispatch_model()
                ist_model.sub.reset_optimizer()              # fresh Adam (:405-407)
            if native:      # HIP events around this step's SpMM/GEMM launches?
                # (instrumented steps sit in the MIDDLE of their stretch: the first step behind a fence is not a sample)
                engine.plan.timer = sample_timer if (sample_timer is not None and s % every == every // 2) else None
            engine.train_step(b, lr, 0.0)
            if ids_log is not None:         # bookkeeping only: no device work in the timed region
                ids_log.append(b.ids)
                n_log.append(b.n)
            if loss_log is not None and (s == 0 or s == count - 1):
                loss_log.append(engine.loss.clone())         # device copy, no host sync
            state['total_iter'] = ti + 1
            if ist_model is not None and state['total_iter'] % args.iter_per_site == 0:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ist_model.sync_model()                       # :422-427
                e1.record()
                sync_ms.append((e0, e1))

    gc_log, gc_t0 = [], [0.0]

    def gc_cb(phase, info):      # (diagnostics: collections of the cyclic GC that take more than a millisecond of host time)
        if phase == 'start':
            gc_t0[0] = time.perf_counter()
        else:
            d_ = (time.perf_counter() - gc_t0[0]) * 1e3
            if d_ > 1.0:
                gc_log.append((info.get('generation'), round(d_, 2), info.get('collected')))
    gc.callbacks.append(gc_cb)

    def host_counters():
        """What can keep a runnable Python thread off its core: run-queue wait of this thread (/proc schedstat), CFS
        bandwidth throttling of the container (cgroup cpu.stat), context switches, page faults."""
        import resource
        c = {}
        try:
            a, b, n_ = open('/proc/thread-self/schedstat').read().split()
            c['on_cpu_ms'], c['runqueue_wait_ms'], c['timeslices'] = int(a) / 1e6, int(b) / 1e6, int(n_)
        except Exception:
            pass
        for f_ in ('/sys/fs/cgroup/cpu.stat', '/sys/fs/cgroup/cpu/cpu.stat', '/sys/fs/cgroup/cpu,cpuacct/cpu.stat'):
            try:
                for line in open(f_):
                    k_, v_ = line.split()
                    if k_ in ('nr_throttled', 'throttled_usec', 'throttled_time', 'nr_periods'):
                        c['cgroup_' + k_] = int(v_)
                break
            except Exception:
                continue
        ru = resource.getrusage(resource.RUSAGE_SELF)
        c['minor_faults'], c['major_faults'] = ru.ru_minflt, ru.ru_majflt
        c['vol_ctx'], c['invol_ctx'] = ru.ru_nvcsw, ru.ru_nivcsw
        return c

    def counter_delta(a, b):
        return {k: (round(b[k] - a[k], 2) if isinstance(b[k], float) else b[k] - a[k]) for k in b if k in a}

    def gap_stats(gaps):
        """Host time from the start of one step's issue to the next's (no synchronisation inside a timed region: the host
        runs ahead of the GPU; a long gap is the host, not a kernel)."""
        if gaps.size == 0:
            return None
        big = np.flatnonzero(gaps > 1.0)
        return {'median': round(float(np.median(gaps)), 4), 'max': round(float(gaps.max()), 3),
                'over_1ms': [(int(i), round(float(gaps[i]), 2)) for i in big[:12]]}

    def fence(collect=False):
        if collect:       # (host hygiene, before a timed region only: no generation-2 pass of the cyclic GC over the
            gc.collect()  # process's ~10^6 long-lived objects inside it: 20-70 ms of host time, profiles/r05_module_path.md)
            gc.freeze()
            torch.cuda.synchronize(dev)
            time.sleep(0.12)      # (the set-up's CPU bursts belong to an earlier CFS period than the region's first step)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed_region(steps, with_timer, warmup=0):
        """`warmup` untimed steps, then exactly `steps` steps between two fences (barrier + synchronize on both sides);
        returns (elapsed, kernel records, logs).  The host hygiene of a region (a full collection, the 0.12-s wait for a
        fresh CFS period) comes BEFORE its warm-up steps, not between them and the timed steps: 120 ms of idle GPU in front
        of the first timed step is a clock ramp inside the region -- 2.69 ms per step over 20 steps against 2.59 over 150
        (the driver's invocation is 20 steps) for a sleep this script put there itself."""
        sample_timer = None
        if with_timer and native:   # HIP events recorded by the native step driver on the launch stream
            sample_timer = engine.enable_timer((steps // every + 1) * (12 * len(dims) + 4))
        elif with_timer:
            hip.profile_begin()
        ids_log, n_log, loss_log = [], [], []
        fence(collect=True)
        if warmup > 0:
            run_steps(warmup)
            del sync_ms[:]             # syncs of the warm-up are not part of the timed region
            fence()
        t0 = time.time()
        run_steps(steps, sample_timer, ids_log, n_log, loss_log)
        fence()
        elapsed = time.time() - t0
        prof = None
        if with_timer and native:
            rec = engine.read_timer()
            prof = {'gemm': [(ms, (m, n, k)) for (ms, kind, m, n, k) in rec if kind == 1],
                    'h3': [(ms, (m, n, k)) for (ms, kind, m, n, k) in rec if kind == 2],
                    'pre': [ms for (ms, kind, m, n, k) in rec if kind == 3],
                    'spmm': [(ms, (m, n, k)) for (ms, kind, m, n, k) in rec if kind == 0]}
            engine.disable_timer()
        elif with_timer:
            p = hip.profile_end()
            prof = {'gemm': [(ms, s[1:]) for ms, s in p['gemm']], 'h3': [], 'pre': [],
                    'spmm': p['spmm']}
        return elapsed, prof, ids_log, n_log, loss_log

    def gemm_roofline(prof, elapsed, steps, mode):
        """`roofline` of the dominant kernel of a timed region run in GEMM mode `mode`."""
        n_instr = len(range(every // 2, steps, every))      # instrumented steps of the region
        step_ms = elapsed * 1e3 / steps
        share = lambda ms: round(ms / (n_instr * step_ms), 4) if n_instr else None
        gem = prof['gemm']
        g_ms = sum(ms for ms, _ in gem) + sum(prof['pre'])       # + split work outside the calls
        g_flop = sum(2.0 * m * n * k for _, (m, n, k) in gem)
        ach = g_flop / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        sampled = 'every %d-th timed step (%d of %d)' % (every, n_instr, steps)
        h3 = prof['h3']
        # (per-rank widths below 2048: only the dZ projection of the middle layers runs on a bf16x3 kernel --
        # the convert-on-load one --; the dominant kernel is then the fp32 one and the line says so)
        h3_dominant = bool(h3) and sum(ms for ms, _ in h3) >= 0.5 * g_ms
        if mode in ('f16x3', 'bf16x3') and h3 and h3_dominant:
            terms = 3 if mode == 'f16x3' else 6
            # the split GEMM's main kernel, bracketed on its own.  achieved = ALGORITHMIC flops
            # (2mnk) / its time; it executes 3 f16 MFMA flops per algorithmic flop.
            h_ms = sum(ms for ms, _ in h3)
            h_flop = sum(2.0 * m * n * k for _, (m, n, k) in h3)
            h_ach = h_flop / (h_ms * 1e-3) / 1e12 if h_ms > 0 else 0.0
            traffic, src = _traffic('gemm_h3_traffic.json' if terms == 3 else 'gemm_b3_traffic.json')
            shapes = set(s for _, s in h3)
            return {
                'kernel': ('gist::gemm_h3_kernel (v_mfma_f32_16x16x32_f16, 3 MFMA flops per algorithmic '
                           'flop: ah.bh + ah.bl + al.bh)') if terms == 3 else
                          ('gist::gemm_b3_kernel (v_mfma_f32_16x16x32_bf16, 6 MFMA flops per algorithmic '
                           'flop: three bf16 pieces per operand = all 24 bits, six cross terms)'),
                # achieved = ALGORITHMIC flops (2mnk) / the main kernel's time.  The kernel executes
                # `terms` dense 16-bit MFMA flops per algorithmic flop, so the roofline that bounds it is
                # the 16-bit dense MFMA peak / terms; `frac` = achieved / that = executed MFMA rate / 2500.
                'bound': 'mfma', 'achieved': round(h_ach, 3),
                'peak': round(MFMA_F16_PEAK_TFLOPS / terms, 2),
                'unit': 'TFLOP/s', 'frac': round(terms * h_ach / MFMA_F16_PEAK_TFLOPS, 4),
                'peak_note': 'dense bf16/f16 MFMA peak %.0f TFLOP/s / %d MFMA flops per algorithmic flop'
                             % (MFMA_F16_PEAK_TFLOPS, terms),
                'mfma_flops_per_algorithmic_flop': terms,
                'mfma_rate_tflops': round(terms * h_ach, 1),
                'peak_16bit_mfma': MFMA_F16_PEAK_TFLOPS,
                'frac_algorithmic_of_16bit_peak': round(h_ach / MFMA_F16_PEAK_TFLOPS, 4),
                'achieved_over_f32_mfma_peak': round(h_ach / MFMA_F32_PEAK_TFLOPS, 3),
                'traffic': traffic, 'traffic_source': src, 'launches': len(h3), 'sampled': sampled,
                'avg_launch_ms': round(h_ms / max(len(h3), 1), 5), 'share_of_step': share(h_ms),
                # every projection call of the step: split pre-pass + main kernel, and the
                # class-layer GEMMs that stay on gist::gemm_f32_kernel
                'all_projection_calls': {
                    'achieved': round(ach, 3), 'unit': 'TFLOP/s (algorithmic)',
                    'calls': len(gem), 'avg_call_ms': round(g_ms / max(len(gem), 1), 5),
                    'share_of_step': share(g_ms),
                    'split_prepass_share_of_step': share(
                        sum(ms for ms, s in gem if s in shapes) - h_ms + sum(prof['pre'])),
                },
            }
        traffic, src = _traffic('gemm_f32_traffic.json')
        extra = {}
        if h3:
            h_ms = sum(ms for ms, _ in h3)
            h_flop = sum(2.0 * m * n * k for _, (m, n, k) in h3)
            extra['bf16x3_projections'] = {
                'kernel': 'gist::gemm_b3c_kernel (convert on load: operands split into three bf16 pieces inside '
                          'the GEMM, six cross terms on v_mfma_f32_16x16x32_bf16) -- the dZ projections only',
                'launches': len(h3), 'avg_launch_ms': round(h_ms / len(h3), 5),
                'achieved_tflops_algorithmic': round(h_flop / (h_ms * 1e-3) / 1e12, 2) if h_ms > 0 else 0.0,
                'share_of_step': share(h_ms)}
        return {
            **extra,
            'kernel': 'gist::gemm_f32_kernel (v_mfma_f32_32x32x2_f32; NT/NN/TN)',
            'bound': 'mfma', 'achieved': round(ach, 3), 'peak': MFMA_F32_PEAK_TFLOPS,
            'unit': 'TFLOP/s', 'frac': round(ach / MFMA_F32_PEAK_TFLOPS, 4),
            'traffic': traffic, 'traffic_source': src, 'launches': len(gem), 'sampled': sampled,
            'avg_launch_ms': round(g_ms / max(len(gem), 1), 5), 'share_of_step': share(g_ms),
        }

    # ---- headline: W untimed warm-up steps, then exactly K timed steps ------------------------
    t_setup = time.time() - t_start
    # a step is 0.3-3 ms: an hour for warm-up + timed region + legs means a collective never returned
    wd.phase('run', float(os.environ.get('GIST_BENCH_RUN_TIMEOUT_S', '3600')))
    if ist_model is not None and world > 1:
        # the run's first all-gather -- RCCL's kernel load and whatever it sets up lazily for that collective -- happens
        # HERE, not at iteration 100 inside a timed region of ~50 ms: phase 1 of sync_model alone (sync_gather: every
        # rank's flat sub-model into the gather buffer), which changes no weight
        ist_model.sync_gather()
        torch.cuda.synchronize(dev)
    timing = not args.no_kernel_timing
    hc_head0 = host_counters()
    elapsed_local, prof, timed_ids, n_log, loss_log = timed_region(args.steps, timing, warmup=args.warmup)
    hc_head = counter_delta(hc_head0, host_counters())
    head_gaps = np.diff(np.asarray(host_stamps)) * 1e3
    engine.check_extract()         # no batch of the warm-up / timed region was built from a timed-out extraction
    n_sync_timed = len(sync_ms)
    sync_timed = list(sync_ms)

    stats = torch.tensor([elapsed_local], dtype=torch.float64, device='cpu' if shared_gpu else dev)
    per_rank = [stats.clone() for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, stats)
    else:
        per_rank = [stats]
    per_rank_s = [float(t.item()) for t in per_rank]
    elapsed = max(per_rank_s)                                   # MAX over ranks
    loss_first, loss_last = (float(loss_log[0].item()), float(loss_log[-1].item()))

    # ---- the launch floor: how many kernels a step issues, and what as many EMPTY kernels cost on this box ------------
    # (a narrow-width step is launch-bound: 14-27 kernels of 5-50 us; the chain of empty launches is the time below which
    # no arrangement of the same number of kernels can go -- reported in the same line as the step it bounds)
    launch_floor = None
    if native and world == 1:      # (N > 1: extra steps on one rank would enter a weight exchange alone)
        try:
            Lc = hip._lib.load()
            c0 = int(Lc.gist_launch_count())
            run_steps(8)
            fence()
            per_step = (int(Lc.gist_launch_count()) - c0) / 8.0
            n_l = int(round(per_step))
            strm = torch.cuda.current_stream(dev).cuda_stream
            e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps_ = 200
            for _ in range(20):
                Lc.gist_empty_launches(n_l, 256, 256, strm)
            torch.cuda.synchronize(dev)
            e0_.record()
            for _ in range(reps_):
                Lc.gist_empty_launches(n_l, 256, 256, strm)
            e1_.record()
            torch.cuda.synchronize(dev)
            chain_us = e0_.elapsed_time(e1_) / reps_ * 1e3
            launch_floor = {'launches_per_step': round(per_step, 2), 'empty_chain_us': round(chain_us, 2),
                            'us_per_empty_launch': round(chain_us / max(n_l, 1), 3),
                            'share_of_step': round(chain_us / (elapsed / args.steps * 1e6), 4),
                            'note': 'gist_empty_launches: %d back-to-back launches of a kernel that does nothing (256 x 256 '
                                    'threads), the step\'s own launch count (gist_launch_count over 8 steps), HIP events on '
                                    'the step\'s stream, mean of %d chains' % (n_l, reps_)}
            engine.check_extract()
        except Exception as e:                          # report, never fake
            launch_floor = {'error': repr(e)}

    # ---- weight exchange, measured on its own (outside the timed region) ----------------------
    sync_info = None
    if ist_model is not None:
        reps = 3
        fence()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
               torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        rstate = random.getstate()                   # dispatch consumes python `random`: restore
        for a, b_, c in ev:
            a.record()
            ist_model.sync_model()
            b_.record()
            ist_model.dispatch_model()
            c.record()
        random.setstate(rstate)
        fence()
        s_ms = float(np.median([a.elapsed_time(b_) for a, b_, c in ev]))
        d_ms = float(np.median([b_.elapsed_time(c) for a, b_, c in ev]))
        t = torch.tensor([s_ms, d_ms], dtype=torch.float64, device='cpu' if shared_gpu else dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        s_ms, d_ms = float(t[0].item()), float(t[1].item())
        P = ist_model.sub.numel
        sync_info = {
            'sync_ms_per_exchange': round(s_ms, 4), 'dispatch_ms': round(d_ms, 4),
            'measured': 'median of %d sync_model + dispatch_model pairs after the timed region, '
                        'max over ranks (HIP events)' % reps,
            'all_gather_bytes_per_rank': int(4 * P), 'all_gather_bytes_total': int(4 * P * world),
            'syncs_inside_timed_region': n_sync_timed,
            'sync_ms_inside_timed_region': [round(a.elapsed_time(b_), 4) for a, b_ in sync_timed],
            'amortized_ms_per_step': round((s_ms + d_ms) / args.iter_per_site, 5),
        }

    # ---- --config 5 on one GPU: the H=32768, S=8 weight exchange, measured on this GPU ----------
    uw_info = None
    if args.config == 5 and world == 1 and args.emulated_rank_of:
        uw_info = ultra_wide_exchange(dev, in_feats, n_classes, args.emulated_rank_of * H, args.emulated_rank_of,
                                      L, args.iter_per_site)

    # ---- N=1: the same workload in the other GEMM modes, same process --------------------------
    legs = {}
    if second_leg:
        for other in ('f32', 'bf16x3', 'f16x3'):
            if other == args.gemm_mode:
                continue
            hip.gemm_mode(other)
            n_re = max(args.steps // 3, 10)
            e2, prof2, _, _, _ = timed_region(n_re, timing, warmup=3)
            leg = {'value': round(n_re / STEPS_PER_EPOCH / e2, 4), 'unit': 'epochs/s',
                   'ms_per_step': round(e2 / n_re * 1e3, 4), 'steps': n_re, 'gemm_mode': other,
                   'dtype': {'f32': 'f32',
                             'bf16x3': 'f32 storage/accumulation; projection products from 3 bf16 pieces per '
                                       'operand (all 24 bits), 6 cross terms on v_mfma_f32_16x16x32_bf16',
                             'f16x3': 'f32 storage/accumulation; projection products as 3 f16-split MFMA '
                                      'terms (22 of 24 operand bits)'}[other],
                   'note': 'same process, same workload, re-timed after the headline run; not `value`'}
            if prof2 is not None:
                leg['roofline'] = gemm_roofline(prof2, e2, n_re, other)
            legs[{'f32': 'f32_mfma', 'bf16x3': 'bf16x3_split', 'f16x3': 'f16x3_split'}[other]] = leg
        hip.gemm_mode(args.gemm_mode)
        engine.check_extract()

    # ---- N=1: the same workload through the reference's own loop body (module path), same process ----------------
    if world == 1 and ist_model is None and not module_headline and not args.no_module_leg:
        try:
            ml = ModuleLoop()
            n_re = max(args.steps, 10)      # (as many steps as the headline: a 10-step leg is noise-limited)
            fence(collect=True)
            ml.run(max(min(args.warmup, 10), 3))
            fence()
            t0 = time.time()
            stamps = []
            hc0 = host_counters()
            if os.environ.get('GIST_BENCH_STEP_BREAKDOWN'):
                ml.breakdown = {}
                ml._native_ms = [0.0]
                orig_native = ml.engine._native_step

                def timed_native(*a_, **k_):
                    t_ = time.perf_counter()
                    r_ = orig_native(*a_, **k_)
                    ml._native_ms.append((time.perf_counter() - t_) * 1e3)
                    if len(ml._native_ms) > 8:
                        del ml._native_ms[:4]
                    return r_
                ml.engine._native_step = timed_native
            ml.run(n_re, stamps=stamps)
            fence()
            e3 = time.time() - t0
            gaps = np.diff(np.asarray(stamps)) * 1e3
            hc1 = host_counters()
            ml.engine.check_extract()
            legs['module_path'] = {
                'value': round(n_re / STEPS_PER_EPOCH / e3, 4), 'unit': 'epochs/s', 'ms_per_step': round(e3 / n_re * 1e3, 4),
                'steps': n_re, 'vs_engine_path_ms_per_step': round((e3 / n_re) / (elapsed / args.steps), 4),
                'host_issue_ms_per_step': gap_stats(gaps), 'gc_collections_over_1ms': list(gc_log),
                'host_counters_delta': counter_delta(hc0, hc1),
                **({'slowest_step_host_ms': ml.breakdown} if ml.breakdown is not None else {}),
                'host_path': 'pred = model(cluster); loss = loss_f(pred[mask], labels[mask]); optimizer.zero_grad(); '
                             'loss.backward(); optimizer.step() -- cluster_gcn/cluster_gcn.py:96-105 on gist_amd.modules.GCN, '
                             'nn.CrossEntropyLoss, optim.Adam, sampler.ClusterIter: three gist_sage_step phase calls per '
                             'iteration (gist::gcn_forward, gist::gcn_backward, the flat-arena Adam)',
                'note': 'same process, same workload, timed after the headline run; not `value`'}
            del ml
        except Exception as e:                          # report, never fake
            legs['module_path'] = {'value': None, 'error': repr(e)}
        torch.cuda.empty_cache()

    # ---- N=1: the same model on a graph whose parts are NOT planted (a child process: its own dataset, partition, engine) ----
    # The headline's batches are 20 planted blocks of 102 / 103 rows: 2045-2046 rows = exactly eight 256-row tiles.  Parts cut
    # by a partitioner are uneven (every second batch of this graph has a ninth row tile) and now and then two parts of one
    # community share a batch: the kernels' schedule for those shapes is part of the product (sampler.py:85-93 batches
    # whatever METIS returns), so its number sits in the same line.
    if (world == 1 and ist_model is None and not args.no_unplanted_leg and args.dataset == 'reddit-synth'
            and args.partition == 'planted' and args.host_path == 'engine'):
        import subprocess
        n_un = max(args.steps // 2, 50)
        cmd = [sys.executable, os.path.abspath(__file__), '--dataset', 'reddit-communities', '--steps', str(n_un),
               '--warmup', str(max(min(args.warmup, 10), 3)), '--n-hidden', str(H), '--n-layers', str(L),
               '--dropout', str(args.dropout), '--gemm-mode', args.gemm_mode, '--batch-parts', str(batch_size),
               '--no-second-leg', '--no-module-leg', '--no-cpu-baseline', '--no-unplanted-leg']
        for kv in args.tune:
            cmd += ['--tune', kv]
        try:
            torch.cuda.synchronize()
            cp = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                timeout=float(os.environ.get('GIST_BENCH_UNPLANTED_TIMEOUT_S', '600')))
            line = [l for l in cp.stdout.splitlines() if l.startswith('{"metric"')]
            if cp.returncode != 0 or not line:
                raise RuntimeError('child exited %d: %s' % (cp.returncode, cp.stderr[-400:]))
            cu = json.loads(line[-1])
            legs['unplanted_graph'] = {
                'value': cu['value'], 'unit': 'epochs/s', 'ms_per_step': cu['ms_per_step'], 'steps': cu['steps'],
                'vs_planted_ms_per_step': round(cu['ms_per_step'] / (elapsed / args.steps * 1e3), 4),
                'workload': cu['config']['workload'], 'partition': cu.get('partition'),
                'roofline': cu.get('roofline'), 'roofline_spmm': cu.get('roofline_spmm'),
                'batch_rows': cu.get('batch_rows'),
                'note': 'a child process of this run, after the headline: same model, GEMM mode and step count rule on '
                        'the Reddit-sized power-law community graph, parts from gist_partition_graph (99-106 rows, '
                        'every second batch has more than 2048 rows); not `value`'}
        except Exception as e:                          # report, never fake
            legs['unplanted_graph'] = {'value': None, 'error': repr(e)}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        # every rank runs `steps` iterations of its own sub-GCN: S * steps / 75 epochs of work
        value = world * args.steps / STEPS_PER_EPOCH / elapsed
        # a split mode whose thresholds no projection of this run reached (small per-rank widths) IS the
        # fp32-MFMA run: say so
        eff_mode = args.gemm_mode
        if prof is not None and args.gemm_mode != 'f32' and (
                not prof['h3'] or sum(ms for ms, _ in prof['h3']) <
                0.5 * (sum(ms for ms, _ in prof['gemm']) + sum(prof['pre']))):
            eff_mode = 'f32'
        out = {
            'metric': 'epochs/sec', 'value': round(value, 4), 'unit': 'epochs/s',
            'epochs_per_sec_per_rank': round(value / world, 4),
            'scaling_note': ('value = SUM over the %d ranks of each rank\'s epochs/s: under GIST every rank trains its '
                             'OWN sub-GCN of width n_hidden/%d on the full batch stream for n_epochs/%d epochs '
                             '(cluster_gcn_ist_distrib.py:385), so per-rank work SHRINKS as N grows and `value` at '
                             'N>1 is not a speed-up of the N=1 model; compare epochs_per_sec_per_rank and '
                             'per_rank_ms_per_step across N, and wall time per GIST epoch = 1 / epochs_per_sec_per_rank'
                             % (world, world, world)) if world > 1 else
                            'N=1: the full-width model on one GPU; value == epochs_per_sec_per_rank',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,
            'dtype': {'f32': 'f32',
                      'bf16x3': 'f32 (projection products from 3 bf16 pieces per operand = all 24 bits, 6 cross '
                                'terms, fp32 accumulation)',
                      'f16x3': 'f32 (projection products as 3 f16-split MFMA terms = 22 bits, fp32 '
                               'accumulation)'}[eff_mode],
            'arithmetic_note': {
                'f32': 'fp32 storage, products and accumulation everywhere: every projection on '
                       'v_mfma_f32_32x32x2_f32',
                'bf16x3': 'storage, accumulation and every non-GEMM kernel are fp32; the large projections carry '
                          'each fp32 operand as b1 + b2 + b3 (bf16, exact: all 24 significant bits, fp32 exponent '
                          'range, no scales) and accumulate a1.b1 + a1.b2 + a2.b1 + a2.b2 + a1.b3 + a3.b1 in fp32 '
                          'on v_mfma_f32_16x16x32_bf16; dropped terms are < 2^-23 of a product; error vs float64 '
                          'at the fp32-MFMA kernel\'s level on Gaussian, cancellation, 2^40 in-row range and '
                          'exponent-extreme operands (tests/test_gemm_b3_gpu.py), golden training run forced onto '
                          'this kernel (tests/test_e2e_gpu.py); the fp32-MFMA run of the same workload is leg '
                          '`f32_mfma`',
                'f16x3': 'storage, accumulation and every non-GEMM kernel are fp32; the large projections '
                         'split each fp32 operand into two f16 halves (22 bits) and accumulate ah.bh + '
                         'ah.bl + al.bh in fp32 -- narrower than fp32'}[eff_mode],
            'data': 'synthetic',
            'config': {
                **({'tune': list(args.tune)} if args.tune else {}),
                'workload': '%s synthetic (N_train=%d, F=%d, C=%d, %d parts, '
                            'batch=%d parts, %d steps/epoch); GraphSAGE n_hidden=%d n_layers=%d '
                            'LayerNorm dropout=%.2f Adam lr=0.01; %s' % (
                                {'reddit-synth': 'Reddit-like block model', 'amazon-synth': 'Amazon-like block model',
                                 'reddit-communities': 'Reddit-sized power-law community graph (no planted parts)'}[args.dataset]
                                + (', parts from gist_partition_graph' if args.partition == 'own' else ', planted blocks as parts'),
                                g.number_of_nodes(), in_feats, n_classes, psize, batch_size, STEPS_PER_EPOCH,
                                H, L, args.dropout,
                                ('ONE RANK of the %d-GPU GIST run of BASELINE config %d emulated on 1 GPU: its '
                                 'sub-GCN of width %d (n_hidden %d / %d), no weight exchange in the timed region'
                                 % (args.emulated_rank_of, args.config, H, H * args.emulated_rank_of,
                                    args.emulated_rank_of)) if args.emulated_rank_of else
                                'full-width model on 1 GPU (cluster_gcn.py path)' if S == 1 else
                                'GIST %d sub-GCNs of width %d, sync every %d iters via one RCCL '
                                'all-gather (cluster_gcn_ist_distrib.py path)'
                                % (S, H // S, args.iter_per_site)),
                'baseline_config': args.config if args.config is not None else 3,
                'dataset': args.dataset,
                'n_hidden': H, 'n_layers': L, 'num_subnet': S, 'batch_parts': batch_size,
                'gemm_mode': args.gemm_mode if eff_mode == args.gemm_mode else
                             ('%s requested; at this width only the dZ projections run on a bf16x3 kernel (convert on '
                              'load), every other GEMM on the fp32 MFMA' % args.gemm_mode) if (prof and prof['h3']) else
                             '%s requested; no projection of this width reaches its thresholds: every GEMM on the fp32 MFMA' % args.gemm_mode,
                'psize': psize, 'steps_per_epoch': STEPS_PER_EPOCH,
                'epochs_per_sec_per_rank': round(value / world, 4),
            },
            'rccl_ranks': (dist.get_world_size() if world > 1 else 1),
            'rccl_devices_distinct': bool(devices_distinct),
            'rank_devices': idents,
            'layer0_aggregation': ('formed by the batch extraction: a batch is a union of whole METIS parts, so the sum of the input '
                                   'features over a row\'s in-neighbours INSIDE its own part is the same in every batch -- computed once per '
                                   'run on the device (setup_seconds.intra_part_sums), the timed extraction adds the kept neighbours in the '
                                   'batch\'s other parts and the in-degree norm; GIST_STEP_PREAGG=0 keeps it a launch of its own (+13 to +28 us '
                                   'per step, profiles/r04_preagg.txt)') if (engine.plan is not None and engine.plan.feat_intra) else
                                  'its own launch in every step',
            'setup_seconds': {'total_before_first_step': round(t_setup, 2), 'dataset': round(t_dataset, 2),
                              'intra_part_sums': round(getattr(it, 'intra_part_sums_seconds', 0.0), 3),
                              'dataset_how': ('rank 0 %s the synthetic graph%s' % (
                                  dataset_how, '' if world == 1 else
                                  ' and wrote it to /dev/shm before any GPU call; the other %d ranks loaded it' % (world - 1)))},
            # one-time work of a run that `value` does not contain, amortised over a run of the reference's default
            # length (cluster_gcn/cluster_gcn.py:154: --n-epochs 40; per rank under GIST: 40 / S local epochs)
            'amortised_over_a_40_epoch_run': (lambda ips, per_rank: {
                'intra_part_sums_s': round(ips, 3),
                'local_epochs_per_rank': 40 // world,
                'value_with_that_second_spread_over_the_run': round(
                    world * (40 // world) / ((40 // world) / per_rank + ips), 4) if per_rank > 0 else None,
                'loss_of_value': round(1.0 - ((40 // world) / ((40 // world) / per_rank + ips)) / per_rank, 5) if per_rank > 0 else None,
            })(getattr(it, 'intra_part_sums_seconds', 0.0), value / world),
            'backend': ('gloo, host-staged (validation)' if shared_gpu else 'nccl (RCCL)') if world > 1 else None,
            'per_rank_ms_per_step': [round(s / args.steps * 1e3, 4) for s in per_rank_s],
            'loss_first': round(loss_first, 5), 'loss_last': round(loss_last, 5),
            'loss_note': 'CE of the first and last timed batch (ln 41 = 3.71 for uniform logits). Adam at '
                         'lr=0.01 on the 4096-wide random init overshoots for the first ~40 steps (loss '
                         'rises to ~100), then trains: the default 150-step run goes 78 -> 4.6, 3000 steps '
                         'reach 2.8 (DESIGN.md section 5); a --steps 20 run sees only the transient',
            **({'INVALID': 'GIST_BENCH_SHARED_GPU validation run: ranks share one GPU, host-staged gloo'} if shared_gpu else {}),
            'host_issue_ms_per_step': gap_stats(head_gaps), 'host_counters_delta': hc_head,
            'host_path': ('module path: the reference loop body on gist_amd.modules.GCN / nn.CrossEntropyLoss / optim.Adam / '
                          'sampler.ClusterIter (three gist_sage_step phase calls per iteration)') if module_headline else
                         'native step driver (gist_sage_step, 1 call/iteration)' if native else 'python op-by-op',
        }
        if sync_info is not None:
            out['weight_sync'] = sync_info
        if uw_info is not None:
            out['ultra_wide_exchange'] = uw_info
        if prof is not None:
            out['roofline'] = gemm_roofline(prof, elapsed_local, args.steps, eff_mode)
            # in-batch edge counts of the timed batches: re-extract them now, outside the timing
            nnz = np.zeros(args.steps, np.int64)
            for i, ids in enumerate(timed_ids):
                bb = it.batcher.extract(ids, engine.z0_left(ids.numel()))
                nnz[i] = int(bb.rowptr[bb.n].item())
            sp = prof['spmm']
            # aggregation launches per instrumented step: 2 (L + 1) - 1, one fewer when layer 0's is formed by the extraction
            n_instr = len(range(every // 2, args.steps, every))      # instrumented steps (as in gemm_roofline)
            per_step = max(len(sp) // max(n_instr, 1), 1)
            s_ms = sum(ms for ms, _ in sp)
            s_bytes = 0.0
            for idx, (ms, (n, n_src, d)) in enumerate(sp):
                z = int(nnz[min((idx // per_step) * every + every // 2, args.steps - 1)])      # (the sampled step's batch)
                s_bytes += 4.0 * (n + 1) + 4.0 * z + 4.0 * n_src * d + 4.0 * n * d
            s_ach = s_bytes / (s_ms * 1e-3) / 1e9 if s_ms > 0 else 0.0
            traffic, src = _traffic('spmm_traffic.json')
            copy_gbs = measured_copy_gbs(dev)
            out['roofline_spmm'] = {
                'kernel': ('gist::spmm_csr_mfma_kernel (block-dense, wide layers)' if H // S >= 1536 else 'gist::spmm_csr_lds2_kernel (LDS gather, wide layers)') + ("; the F=%d input layer's aggregation is formed by the batch extraction (no launch)" % in_feats if engine.plan is not None and engine.plan.feat_intra else '; gist::spmm_csr_rowsplit_kernel / spmm_csr_kernel for the F=%d input layer' % in_feats), 'bound': 'hbm', 'achieved': round(s_ach, 2),
                'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(s_ach / HBM_PEAK_GBS, 4),
                # achievable HBM bandwidth on this box: a 1 GiB device-to-device copy (read+write)
                'peak_measured_copy': round(copy_gbs, 1),
                'frac_of_measured_copy': round(s_ach / copy_gbs, 4) if copy_gbs > 0 else None,
                'traffic': traffic, 'traffic_source': src, 'launches': len(sp),
                'avg_launch_ms': round(s_ms / max(len(sp), 1), 5),
                'avg_algorithmic_bytes': round(s_bytes / max(len(sp), 1), 1),
                'share_of_step': round(s_ms / (n_instr * elapsed_local * 1e3 / args.steps), 4),
                'mean_batch_rows': round(float(np.mean(n_log)), 1),
                'mean_batch_nnz': round(float(nnz.mean()), 1),
            }
        if n_log:
            nl = np.asarray(n_log)
            out['batch_rows'] = {'min': int(nl.min()), 'mean': round(float(nl.mean()), 1), 'max': int(nl.max()),
                                 'share_over_2048': round(float((nl > 2048).mean()), 3)}
        if partition_info is not None or os.environ.get('GIST_BENCH_BATCH_STATS'):
            # what the aggregation kernels see: of a batch row's in-batch neighbours, the share inside the row's own part
            # (the diagonal blocks) and the rows with more than 8 neighbours in the batch's OTHER parts (the block-dense
            # kernel keeps 8 per row inline and walks the edges of the others again)
            ins, tot, over8, rows_ = 0, 0, 0, 0
            for ids in timed_ids[:8]:
                n_b = ids.numel()
                bb = it.batcher.extract(ids, engine.z0_left(n_b))
                rp_ = bb.rowptr[:n_b + 1].cpu().numpy().astype(np.int64)
                cl_ = bb.col[:int(rp_[-1])].cpu().numpy().astype(np.int64)
                po_ = it._node_part[:, 0][ids.long()].cpu().numpy()
                row_ = np.repeat(np.arange(n_b), np.diff(rp_))
                same = po_[row_] == po_[cl_]
                ins += int(same.sum()); tot += same.size
                over8 += int((np.bincount(row_[~same], minlength=n_b) > 8).sum()); rows_ += n_b
            batch_stats = {'in_batch_edges_inside_the_rows_part': round(ins / max(tot, 1), 4),
                           'rows_with_more_than_8_neighbours_in_other_parts': round(over8 / max(rows_, 1), 4),
                           'sampled_batches': min(len(timed_ids), 8),
                           'blocked_aggregation': bool(it.locality is not False), 'locality_stats': it.locality_stats}
            out['batch_locality'] = batch_stats
        if partition_info is not None:
            out['partition'] = partition_info
        if launch_floor is not None:
            out['launch_floor'] = launch_floor
        out.update(legs)
        if world == 1 and not args.no_cpu_baseline:
            try:
                ncpu = host_cores()
                # SURVEY 8d / VERDICT r3 item 7: one warm-up step, then ONE FULL EPOCH on all cores (the reference's
                # own timing definition is the epoch loop), bounded by a wall-time budget; 12 steps on one thread
                t_all, n_all, wall_all, pools_all = cpu_baseline(
                    ds, first_epoch_order, dims, use_ln, args.dropout, args.cpu_steps or STEPS_PER_EPOCH, seed,
                    ncpu, batch_size, budget_s=args.cpu_budget)
                t_one, n_one, wall_one, pools_one = cpu_baseline(
                    ds, first_epoch_order, dims, use_ln, args.dropout, 12, seed, 1, batch_size,
                    budget_s=args.cpu_budget / 2)
                full = n_all >= STEPS_PER_EPOCH
                out['cpu_baseline'] = {
                    'value': round((n_all / STEPS_PER_EPOCH) / wall_all, 6) if full else
                             round(1.0 / (STEPS_PER_EPOCH * t_all), 6),
                    'unit': 'epochs/s',
                    'cores': ncpu, 'kind': 'port', 'thread_pools': pools_all,
                    'steps_timed': n_all, 'wall_s': round(wall_all, 2), 'median_step_s': round(t_all, 4),
                    'full_epoch': bool(full),
                    'sample': ('oracle (numpy/OpenBLAS + OpenMP C SpMM) full training step with dropout masks, %d '
                               'threads: 1 untimed warm-up step, then %s -- %d steps in %.1f s wall, median %.3f '
                               's/step; value = %s'
                               % (ncpu,
                                  'ONE FULL EPOCH of the same workload (its first epoch\'s %d batches in order, the '
                                  'reference\'s own timing definition cluster_gcn.py:91,106-108)' % STEPS_PER_EPOCH
                                  if full else
                                  'the first batches of the epoch until the %.0f s budget ran out (an epoch is %d steps)'
                                  % (args.cpu_budget, STEPS_PER_EPOCH),
                                  n_all, wall_all, t_all,
                                  'epochs / wall time of that epoch' if full else '1 / (steps per epoch x median step)')),
                    'one_thread': {'value': round(1.0 / (STEPS_PER_EPOCH * t_one), 6),
                                   'unit': 'epochs/s', 'cores': 1, 'thread_pools': pools_one,
                                   'steps_timed': n_one, 'wall_s': round(wall_one, 2),
                                   'sample': 'same step, first %d batches after 1 warm-up step, 1 thread, median %.2f '
                                             's/step' % (n_one, t_one)},
                }
            except Exception as e:                          # report, never fake
                out['cpu_baseline'] = {'value': None, 'error': repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    wd.phase('done', None)


if __name__ == '__main__':
    main()
