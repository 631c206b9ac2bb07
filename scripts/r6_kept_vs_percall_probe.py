"""Dev probe: parameter checksums after 3 steps of the metric configuration in the kept-split and the per-call-split form
(tests/test_e2e_gpu.py::test_step_kept_split_operands_equal_per_call_splits), for comparing two library builds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_e2e_gpu import _metric_config_engine
from gist_amd import _lib
os.environ['GIST_STEP_PREAGG'] = '0'
for env in ('0', '1'):
    os.environ['GIST_STEP_H3'] = env
    ds, it, eng, dims, params = _metric_config_engine('bf16x3')
    eng.p_drop = 0.2
    it.bind(eng)
    eng.plan.p_drop = 0.2
    L = _lib.load()
    c0 = int(L.gist_launch_count()) if hasattr(L, 'gist_launch_count') else 0
    losses = []
    for j, b in enumerate(it):
        losses.append(float(eng.train_step(b, 0.01, 0.0).item()))
        if j == 2:
            break
    p = eng.arena.params.double()
    print('GIST_STEP_H3=%s' % env, 'losses', ['%.9f' % x for x in losses], 'sum %.12e' % p.sum().item(), 'abs %.12e' % p.abs().sum().item(),
          'launches', int(L.gist_launch_count()) - c0, flush=True)
