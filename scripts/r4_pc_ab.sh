#!/bin/bash
# round 4: the producer/consumer block-dense kernel against the one-workgroup-per-CU kernel (tuning hook spmm_kernel = 4):
# the step (bench.py) and the evaluator's dense pairs (eval_bench, locality 1)
set -o pipefail
mkdir -p gpurun_out/r4
for k in 0 5; do
  timeout -k 10 300 python scripts/bench_tuned.py spmm_kernel=$k -- --steps 150 --warmup 10 --no-cpu-baseline --no-second-leg > gpurun_out/r4/pc_bench_k$k.json 2> gpurun_out/r4/pc_bench_k$k.err || { tail -5 gpurun_out/r4/pc_bench_k$k.err; exit 1; }
done
python - <<'PY'
import json
for k in (0, 5):
    d=json.loads(open('gpurun_out/r4/pc_bench_k%d.json'%k).read().strip().splitlines()[-1])
    s=d['roofline_spmm']
    print('spmm_kernel', k, d['value'], d['ms_per_step'], 'spmm avg us', s['avg_launch_ms']*1e3, 'frac', s['frac'])
PY
cat > /tmp/eb.py <<'PY'
import sys, runpy
sys.path.insert(0, '.')
from gist_amd import hip
hip.tuning('spmm_kernel', int(sys.argv[1]))
sys.argv = ['scripts/eval_bench.py', '--localities', '1']
runpy.run_path('scripts/eval_bench.py', run_name='__main__')
PY
for k in 0 5; do
  timeout -k 10 400 python /tmp/eb.py $k > gpurun_out/r4/pc_eval_k$k.json 2> gpurun_out/r4/pc_eval_k$k.err || { tail -5 gpurun_out/r4/pc_eval_k$k.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4/pc_eval_k$k.json').read().strip().splitlines()[-1])
v=d['variants'][0]['diag_plus_pairs_plus_gather']
print('spmm_kernel $k eval', v['aggregation_D4096'], v['eval_forward_H4096_s'])
PY
done
