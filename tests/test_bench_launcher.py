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
    assert not [l for l in r.stdout.split('\n') if l.startswith('{')]      # no result line


def test_launcher_is_not_used_under_torchrun_env():
    """With WORLD_SIZE in the environment the process is a rank, not a launcher; a mismatch with
    --gpus is an error, not a silent single-GPU run."""
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)
