#!/bin/bash
# Planted block model vs the power-law community graph partitioned by gist_partition_graph, at the metric's width and at
# config 3's 8-GPU per-rank width: bench lines -> gpurun_out/r5/unplanted/*.json (scripts/make_profiles_r5.py condenses them)
: ${GRAFT_REPO_ROOT:?run under gpurun}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5/unplanted
mkdir -p $O
for ds in reddit-synth reddit-communities; do
  for h in 4096 512; do
    steps=150; [ $h = 512 ] && steps=300
    GIST_BENCH_BATCH_STATS=1 python3 $R/bench.py --dataset $ds --n-hidden $h --steps $steps --no-cpu-baseline --no-second-leg > $O/${ds}_h$h.log 2>&1 || { tail -3 $O/${ds}_h$h.log; exit 1; }
    grep '^{"metric"' $O/${ds}_h$h.log | tail -1 > $O/${ds}_h$h.json
    python3 -c "
import json; d=json.load(open('$O/${ds}_h$h.json'))
print('$ds', $h, d['ms_per_step'], d.get('roofline_spmm',{}).get('achieved'), d.get('roofline_spmm',{}).get('avg_launch_ms'), d.get('batch_locality'), (d.get('partition') or {}).get('seconds'), d.get('module_path',{}).get('ms_per_step'))"
  done
done
