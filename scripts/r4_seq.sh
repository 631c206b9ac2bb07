#!/bin/bash
# On the GPU box: kernel sequence (per-launch durations and gaps) of the step for a bench.py configuration.
#   bash scripts/r4_seq.sh <tag> <bench args...>
: ${GRAFT_REPO_ROOT:?}
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/r4/seq_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python3 $R/bench.py "$@" --steps 60 --warmup 10 --no-second-leg --no-cpu-baseline --no-kernel-timing > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 $R/scripts/step_seq.py $(find $O -name 'run_kernel_trace.csv' | head -1) > $O/seq.txt
grep '^{"metric"' $O/run.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" >> $O/seq.txt
find $O -name '*.csv' -delete
cat $O/seq.txt
