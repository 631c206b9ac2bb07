#!/bin/bash
# A/B of one tuning knob on bench.py: r6_tune_ab.sh "<bench args>" knob=value   (two runs each, alternating)
: ${GRAFT_REPO_ROOT:?run under gpurun}
cd $GRAFT_REPO_ROOT
args=$1; kv=$2
for rep in 1 2; do
  for t in "" "--tune $kv"; do
    python3 bench.py $args $t --no-cpu-baseline --no-second-leg --no-module-leg --no-unplanted-leg 2>/dev/null | grep '^{"metric"' | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read())
print('${t:-default}', 'ms/step', d['ms_per_step'], 'gemm avg', d.get('roofline',{}).get('avg_launch_ms'), 'spmm avg', d.get('roofline_spmm',{}).get('avg_launch_ms'))"
  done
done
