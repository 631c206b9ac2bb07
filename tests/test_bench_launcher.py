"""CPU: bench.py's own rank launcher (`python bench.py --gpus N` without torchrun).  There is no
GPU here, so every rank process fails at device selection: the parent must notice, stop the
other ranks and exit non-zero (the success path runs on the GPU box:
tests/test_ist_multiproc_gpu.py::test_bench_starts_its_own_ranks)."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_propagates_rank_failure():
    if torch.cuda.is_available():
        import pytest
        pytest.skip('GPU present: covered by the gpu test')
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2',
                        '--warmup', '1'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'rank process' in r.stderr and 'stopping the others' in r.stderr
    lines = [l for l in r.stdout.split('\n') if l.startswith('{')]         # no result line: one error line, rank 0's
    # (rank 0 prints the error line unless another rank's exit got it stopped first)
    assert len(lines) <= 1 and all('"value": null' in l and 'device_count' in l for l in lines)


def test_launcher_is_not_used_under_torchrun_env():
    """With WORLD_SIZE in the environment the process is a rank, not a launcher; a mismatch with
    --gpus is an error, not a silent single-GPU run."""
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)


def _ps_alive(pids):
    out = []
    for pid in pids:
        try:
            os.kill(pid, 0)
            # a zombie still answers kill(0): look at its state
            with open('/proc/%d/stat' % pid) as f:
                if f.read().split(')')[-1].split()[0] != 'Z':
                    out.append(pid)
        except (OSError, IOError):
            pass
    return out


HANG = ("import os, signal, sys, time\n"
        "signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"       # a rank stuck in a collective ignores SIGTERM
        "open(os.environ['PIDDIR'] + '/' + os.environ['RANK'], 'w').write(str(os.getpid()))\n"
        "time.sleep(600)\n")


def _launcher_env(tmp_path, **kw):
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(GIST_BENCH_RANK_CMD=HANG, PIDDIR=str(tmp_path), **kw)
    return env


def _wait_pids(tmp_path, n, timeout=60):
    import time
    t0 = time.time()
    while time.time() - t0 < timeout:
        names = [f for f in os.listdir(str(tmp_path)) if f.isdigit()]
        if len(names) == n:
            pids = []
            for f in names:
                txt = open(os.path.join(str(tmp_path), f)).read()
                if txt:
                    pids.append(int(txt))
            if len(pids) == n:
                return pids
        time.sleep(0.1)
    raise AssertionError('rank processes did not start')


def test_launcher_sigterm_kills_hung_ranks(tmp_path):
    """SIGTERM to the launcher (a harness timeout) must not leave rank processes behind,
    even ranks that ignore SIGTERM: terminate, grace period, kill; exit code 128+15."""
    import signal
    import time
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '3'],
                         env=_launcher_env(tmp_path), stderr=subprocess.PIPE, text=True)
    pids = _wait_pids(tmp_path, 3)
    assert len(_ps_alive(pids)) == 3
    p.send_signal(signal.SIGTERM)
    _, err = p.communicate(timeout=60)
    assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err)
    t0 = time.time()
    while _ps_alive(pids) and time.time() - t0 < 10:
        time.sleep(0.1)
    assert _ps_alive(pids) == [], err


def test_launcher_deadline_stops_hung_ranks(tmp_path):
    """No rank ever exits (a deadlock): the launcher's own deadline ends the run with a non-zero code."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'],
                       env=_launcher_env(tmp_path, GIST_BENCH_DEADLINE_S='3'), capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 124, (r.returncode, r.stderr)
    assert 'deadline' in r.stderr
    pids = [int(open(os.path.join(str(tmp_path), f)).read()) for f in os.listdir(str(tmp_path)) if f.isdigit()]
    assert len(pids) == 2 and _ps_alive(pids) == []


def test_launcher_exit_code_is_the_failing_ranks(tmp_path):
    """rank 0 exits 3 while the others sleep -- the launcher stops them and exits with 3
    itself, without a traceback (it used to crash in its own bookkeeping and exit 1)."""
    cmd = ("import os, sys, time\n"
           "open(os.environ['PIDDIR'] + '/' + os.environ['RANK'], 'w').write(str(os.getpid()))\n"
           "sys.exit(3) if os.environ['RANK'] == '0' else time.sleep(600)\n")
    env = _launcher_env(tmp_path)
    env['GIST_BENCH_RANK_CMD'] = cmd
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '3'], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert 'Traceback' not in r.stderr, r.stderr
    assert 'rank process 0 exited with 3' in r.stderr
    pids = [int(open(os.path.join(str(tmp_path), f)).read()) for f in os.listdir(str(tmp_path)) if f.isdigit()]
    assert len(pids) == 3 and _ps_alive(pids) == []


def _error_lines(stdout):
    import json
    out = []
    for l in stdout.split('\n'):
        if l.startswith('{'):
            out.append(json.loads(l))
    return out


def test_rank_fails_fast_without_one_gpu_per_rank():
    """First-contact hardening: a rank of a 2-rank run on a node that shows fewer than 2 GPUs prints ONE
    JSON error line (value null) and exits 2 -- before any rendez-vous, dataset work or GPU call."""
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip('two GPUs visible')
    import time
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0', LOCAL_WORLD_SIZE='2',
               MASTER_ADDR='127.0.0.1', MASTER_PORT='29871')
    env.pop('GIST_BENCH_SHARED_GPU', None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr)
    lines = _error_lines(r.stdout)
    assert len(lines) == 1 and lines[0]['value'] is None
    assert 'device_count' in lines[0]['error'] and lines[0]['rank'] == 0
    assert time.time() - t0 < 120


def test_rank_rendezvous_times_out_with_an_error_line():
    """A peer that never joins: init_process_group's time-out (or the watchdog behind it) ends the rank
    with a JSON error line naming the rendez-vous and a non-zero exit code, instead of waiting forever.
    Exercised through the shared-GPU validation mode, whose rendez-vous (gloo) precedes every GPU call."""
    import socket
    import time
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0', LOCAL_WORLD_SIZE='2',
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), GIST_BENCH_SHARED_GPU='1',
               GIST_BENCH_RENDEZVOUS_TIMEOUT_S='4', GIST_BENCH_SHM=os.environ.get('TMPDIR', '/tmp'))
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dataset', 'reddit-synth'],
                       env=env, capture_output=True, text=True, timeout=600)
    import shutil
    shutil.rmtree(os.path.join(env['GIST_BENCH_SHM'], 'gist_bench_reddit-synth_%d' % port), ignore_errors=True)
    assert r.returncode in (3, 124), (r.returncode, r.stderr[-2000:])
    lines = _error_lines(r.stdout)
    assert len(lines) == 1 and lines[0]['value'] is None and lines[0].get('phase') == 'rendezvous', lines
    assert time.time() - t0 < 300


def test_dataset_arrays_round_trip_and_launch_tag(tmp_path, monkeypatch):
    """One synthetic graph per node: what local rank 0 writes for the other ranks (datasets.save_arrays) loads back bit
    for bit, and the directory's name is per LAUNCH (port + parent process), so a directory left by a run that died is
    not this launch's; stale ones are removed."""
    import importlib.util
    import os
    import time
    import numpy as np
    from gist_amd import datasets
    ds = datasets.toy(seed=3, n=600, n_blocks=6, n_feats=17, n_classes=4, train_frac=0.7)
    d = tmp_path / 'one'
    d.mkdir()
    datasets.save_arrays(ds, str(d))
    back = datasets.load_arrays(str(d))
    assert back.num_classes == ds.num_classes and back.name == ds.name
    for k in ('rowptr', 'col', 't_rowptr', 't_col'):
        assert np.array_equal(getattr(back.g, k).numpy(), getattr(ds.g, k).numpy())
    assert sorted(back.g.ndata) == sorted(ds.g.ndata)
    for k in ds.g.ndata:
        assert np.array_equal(back.g.ndata[k].numpy(), ds.g.ndata[k].numpy())
    assert len(back.par_li) == len(ds.par_li) and all(np.array_equal(a, b) for a, b in zip(back.par_li, ds.par_li))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(__file__)), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setenv('MASTER_PORT', '29123')
    tag = bench._shared_tag('reddit-synth')
    assert tag == 'gist_bench_reddit-synth_29123_%d' % os.getppid()
    stale, fresh = tmp_path / 'gist_bench_x_1_2', tmp_path / 'gist_bench_x_1_3'
    stale.mkdir(); fresh.mkdir()
    old = time.time() - 7200
    os.utime(str(stale), (old, old))
    bench._drop_stale_shared(str(tmp_path))
    assert not stale.exists() and fresh.exists() and d.exists()
