"""GPU, multi-process: GIST with one PROCESS per rank, all on the box's one GPU, with the product
block movers (HipBlocks) and the product wrapper / train loop -- the gather -> all-gather ->
scatter chain outside LocalCommGroup.  The collective is host-staged over gloo
(tests/host_staged_comm.py; RCCL refuses two ranks on one device), everything around it is the
path the N-GPU run takes.  Checked against the reference's own runs: G4 (dispatch / sync under
gloo) and G6 (whole train() loop).  Also: `python bench.py --gpus 2` starts its own rank
processes (no torchrun) and prints one JSON line.

At most 5 processes use the GPU at once (this runner + 4 ranks)."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
WORKER = os.path.join(ROOT, 'tests', 'ist_gpu_worker.py')


def _run_ranks(mode, S, port, golden, tmp_path):
    outs = [str(tmp_path / ('rank%d.json' % r)) for r in range(S)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, WORKER, mode, str(r), str(S), str(port),
                               os.path.join(GOLD, golden), outs[r]], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(S)]
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=420)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r in range(S):
        assert os.path.exists(outs[r]), 'rank %d wrote no result:\n%s' % (r, logs[r][-1500:])
        res = json.load(open(outs[r]))
        assert res['errors'] == [], 'rank %d: %s' % (r, res['errors'])
    assert all(p.returncode == 0 for p in procs)


@pytest.mark.parametrize('golden,S,port', [('G4_ist_S2_H16_L2.npz', 2, 29831),
                                           ('G4_ist_S2_H8_L1.npz', 2, 29832),
                                           ('G4_ist_S4_H16_L3.npz', 4, 29833)])
def test_dispatch_sync_one_process_per_rank(golden, S, port, tmp_path):
    _run_ranks('g4', S, port, golden, tmp_path)


@pytest.mark.parametrize('S,port', [(2, 29841), (4, 29842)])
def test_train_loop_one_process_per_rank(S, port, tmp_path):
    _run_ranks('g6', S, port, 'G6_e2e_ist_S%d.npz' % S, tmp_path)


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher: the parent starts 2 rank processes itself
    (before any GPU call) and rank 0 prints ONE JSON line; a failing rank makes the parent exit
    non-zero.  Shared-GPU validation mode (marked INVALID in the line), small width."""
    env = dict(os.environ, GIST_BENCH_SHARED_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6',
           '--warmup', '2', '--n-hidden', '256', '--iter-per-site', '4', '--timing-every', '2']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().split('\n') if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['rccl_ranks'] == 2 and out['steps'] == 6
    assert out['config']['num_subnet'] == 2 and out['dtype'] == 'f32'      # width 128 per rank: fp32 MFMA
    assert len(out['per_rank_ms_per_step']) == 2 and all(t > 0 for t in out['per_rank_ms_per_step'])
    assert abs(out['value'] - 2 * 6 / 75.0 / (out['ms_per_step'] * 6e-3)) < 1e-3 * out['value']
    ws = out['weight_sync']
    assert ws['sync_ms_per_exchange'] > 0 and ws['syncs_inside_timed_region'] >= 1
    assert out['roofline']['frac'] > 0 and out['roofline_spmm']['achieved'] > 0
    assert 'INVALID' in out                       # shared-GPU validation run, never a result
    # a rank that fails takes the whole launch down with a non-zero code
    bad = subprocess.run(cmd + ['--n-hidden', '255'], env=env, capture_output=True, text=True,
                         timeout=600)
    assert bad.returncode != 0


def test_bench_two_ranks_on_the_split_projection_path():
    """Two rank processes at the metric's width (per-rank 2048: the bf16x3 projections with split-K, the
    step's kept split operands, the prepared block-dense aggregation) with a weight exchange every 4
    iterations: the sub-GCN weights are re-dispatched between steps, so the splits the step keeps must be
    rebuilt from them every step.  Shared-GPU validation mode; finite losses, the split path really taken."""
    env = dict(os.environ, GIST_BENCH_SHARED_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '10', '--warmup', '2',
           '--n-hidden', '4096', '--iter-per-site', '4', '--timing-every', '2']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.strip().split('\n') if l.startswith('{')][-1])
    assert out['n_gpus'] == 2 and out['config']['num_subnet'] == 2
    assert out['dtype'].startswith('f32 (projection products from 3 bf16 pieces')
    assert out['roofline']['kernel'].startswith('gist::gemm_b3_kernel') and out['roofline']['frac'] > 0.1
    assert out['roofline_spmm']['kernel'].startswith('gist::spmm_csr_mfma_kernel')
    assert math.isfinite(out['loss_first']) and math.isfinite(out['loss_last'])
    assert out['weight_sync']['syncs_inside_timed_region'] >= 2


@pytest.mark.parametrize('n_hidden,per_rank', [(4096, 1024), (2048, 512)])
def test_bench_four_ranks_per_rank_widths_with_redispatch(n_hidden, per_rank):
    """Four rank processes (the pool allows at most six processes on one card, so the 8-rank point is
    rehearsed by its per-rank WIDTH instead: 2048 / 4 = 512 = 4096 / 8) through 78 iterations with a weight
    exchange every 20: the run crosses into epoch 1, so the sub-GCNs are RE-DISPATCHED under new partitions
    (cluster_gcn_ist_distrib.py:400-403) and trained on; the fused step at the per-rank widths of the N = 4
    and N = 8 points (fp32 MFMA projections, one-launch extraction, deferred reductions).  Shared-GPU
    validation mode: a first SCALE run must not die on a width-specific path."""
    env = dict(os.environ, GIST_BENCH_SHARED_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '78', '--warmup', '2',
           '--n-hidden', str(n_hidden), '--iter-per-site', '20', '--timing-every', '16']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.strip().split('\n') if l.startswith('{')][-1])
    assert out['n_gpus'] == 4 and out['config']['num_subnet'] == 4 and out['rccl_ranks'] == 4
    assert out['dtype'] == 'f32'                           # every projection of this width on the fp32 MFMA
    assert out['epochs_per_sec_per_rank'] > 0 and 'SUM over the 4 ranks' in out['scaling_note']
    assert abs(out['value'] - 4 * out['epochs_per_sec_per_rank']) < 1e-3 * out['value']
    assert len(out['per_rank_ms_per_step']) == 4
    assert math.isfinite(out['loss_first']) and math.isfinite(out['loss_last'])
    ws = out['weight_sync']
    assert ws['syncs_inside_timed_region'] == 4             # iterations 20, 40, 60, 80
    assert ws['all_gather_bytes_total'] == 4 * ws['all_gather_bytes_per_rank']
    assert out['config']['workload'].endswith('(cluster_gcn_ist_distrib.py path)')
    assert 'width %d' % per_rank in out['config']['workload']
