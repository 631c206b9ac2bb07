#!/bin/bash
# A/B of whole libraries on bench.py: r6_lib_ab.sh "<bench args>" lib1.so lib2.so ...  (ms/step, GEMM and SpMM averages)
: ${GRAFT_REPO_ROOT:?run under gpurun}
cd $GRAFT_REPO_ROOT
args=$1; shift
for lib in "$@"; do
  for rep in 1 2; do
    GIST_LIB_PATH=$PWD/$lib python3 bench.py $args --no-cpu-baseline --no-second-leg --no-module-leg 2>/dev/null | grep '^{"metric"' | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$lib', 'ms/step', d['ms_per_step'], 'gemm avg', d.get('roofline',{}).get('avg_launch_ms'), 'spmm avg', d.get('roofline_spmm',{}).get('avg_launch_ms'))"
  done
done
