#!/bin/bash
# kernel traces of the planted and the unplanted graph at one width, compared kernel by kernel:
# r6_ktrace_pair.sh <n_hidden> <steps>
: ${GRAFT_REPO_ROOT:?run under gpurun}
R=$GRAFT_REPO_ROOT
h=$1; steps=$2
O=$R/gpurun_out/r6/pair_h$h
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ds in reddit-synth reddit-communities; do
  rocprofv3 --kernel-trace --output-format csv -d $O/$ds -o k -- python3 $R/bench.py --dataset $ds --n-hidden $h --steps $steps --warmup 10 --no-cpu-baseline --no-second-leg --no-module-leg --no-kernel-timing > $O/$ds.log 2>&1 || { tail -5 $O/$ds.log; exit 1; }
done
fa=$(find $O/reddit-synth -name '*kernel_trace.csv' | head -1)
fb=$(find $O/reddit-communities -name '*kernel_trace.csv' | head -1)
tot=$((steps + 10))
python3 $R/scripts/ktrace_compare.py $fa $tot $fb $tot > $O/compare.txt
cat $O/compare.txt
rm -f $fa $fb
