#!/bin/bash
# A/B on one box: k slices of the convert-on-load bf16x3 projections at the per-rank widths (two workgroups per CU)
set -o pipefail
for rep in 1 2; do
for cfg in "--n-hidden 512 --steps 300" "--n-hidden 1024 --steps 300" "--config 4 --steps 600" "--n-hidden 2048 --steps 200"; do
  for t in "" "--tune b3c_splits=2" "--tune b3c_splits=3"; do
    python bench.py $cfg $t --no-cpu-baseline --no-second-leg --no-module-leg --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$cfg | $t |', d['value'], d['ms_per_step'], d['loss_last'])"
  done
done
done
