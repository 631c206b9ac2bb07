#!/bin/bash
# Dev tool: per-kernel durations (rocprofv3 --kernel-trace) of a python script -- run on the GPU box.
#   bash scripts/ktrace.sh <tag> <script.py> [args...]
: ${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/ktrace_$tag
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python3 $GRAFT_REPO_ROOT/$@ > $OUT/run.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/ktrace_summary.py $OUT/run_kernel_trace.csv ${KFILTER:-_} ${KMODE:-} 
