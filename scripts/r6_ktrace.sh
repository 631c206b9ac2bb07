#!/bin/bash
# per-kernel launch-duration percentiles of one bench.py run: r6_ktrace.sh <tag> "<kernel substrings>" <bench args...>
: ${GRAFT_REPO_ROOT:?run under gpurun}
R=$GRAFT_REPO_ROOT
tag=$1; pats=$2; shift 2
O=$R/gpurun_out/r6/$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o k -- python3 $R/bench.py "$@" --no-cpu-baseline --no-second-leg --no-module-leg --no-kernel-timing > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
f=$(find $O -name '*kernel_trace.csv' | head -1)
python3 $R/scripts/r5_kernel_hist.py $f $pats
[ -n "$KEEP_TRACE" ] || rm -f $f
