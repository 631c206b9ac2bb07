#!/bin/bash
# A/B on one box: the LayerNorm backward in the reverse aggregation's store (default) against its own launch (--tune lnb_fused=1)
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "--config 2 --steps 300" "--n-hidden 256 --steps 300" "--n-hidden 128 --n-layers 3 --steps 300"; do
  for t in "" "--tune lnb_fused=1"; do
    python bench.py $cfg $t --no-cpu-baseline --no-second-leg --no-module-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$cfg | $t |', d['value'], d['ms_per_step'], d['loss_last'])"
  done
done
done
