#!/bin/bash
# round 4: the class layer's dW slabs in the grid of the LayerNorm backward below it (tuning hook class_fused = 2: own launch)
set -o pipefail
mkdir -p gpurun_out/r4dw
for rep in 1 2; do
for cfg in "--n-hidden 512 --steps 300" "--config 2 --steps 300" "--config 4 --steps 600"; do
  for t in 2 0; do
    GIST_BENCH_TUNE_CLASS_FUSED=$t python bench.py $cfg --no-cpu-baseline --no-second-leg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('class_fused=$t | $cfg |', d['value'], d['ms_per_step'], d['loss_last'])"
  done
done
done
