#!/bin/bash
# Planted block model vs the power-law community graph partitioned by gist_partition_graph (bench lines ->
# gpurun_out/r6/unplanted/*.json).  Args: widths (default "4096 512"); TUNE="knob=v ..." adds --tune flags.
: ${GRAFT_REPO_ROOT:?run under gpurun}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6/unplanted${TAG:+_$TAG}
mkdir -p $O
T=""; for kv in $TUNE; do T="$T --tune $kv"; done
for ds in ${DATASETS:-reddit-synth reddit-communities}; do
  for h in ${@:-4096 512}; do
    steps=150; [ $h = 512 ] && steps=300
    GIST_BENCH_BATCH_STATS=1 python3 $R/bench.py --dataset $ds --n-hidden $h --steps $steps --no-cpu-baseline --no-second-leg --no-module-leg $T > $O/${ds}_h$h.log 2>&1 || { tail -3 $O/${ds}_h$h.log; exit 1; }
    grep '^{"metric"' $O/${ds}_h$h.log | tail -1 > $O/${ds}_h$h.json
    python3 -c "
import json; d=json.load(open('$O/${ds}_h$h.json'))
print('$ds', $h, 'ms/step', d['ms_per_step'], 'gemm avg', d.get('roofline',{}).get('avg_launch_ms'), 'spmm', d.get('roofline_spmm',{}).get('achieved'), d.get('roofline_spmm',{}).get('avg_launch_ms'))"
  done
done
