#!/bin/bash
# Dev tool: per-kernel durations (rocprofv3 --kernel-trace) of a python script -- run on the GPU box.
#   bash scripts/ktrace.sh <tag> <script.py> [args...]
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/ktrace_$tag
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python3 $GRAFT_REPO_ROOT/$@ > $OUT/run.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/ktrace_summary.py $OUT/run_kernel_trace.csv ${KFILTER:-_} ${KMODE:-} ; exit 0
python3 - <<PY
import csv, glob, collections
f = glob.glob('$OUT/*kernel_trace.csv')[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][-44:] + ' grid=' + r['Grid_Size'] + ' wg=' + r['Workgroup_Size']
    d[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print('%-90s n=%4d median %8.1f us  min %8.1f' % (k, len(v), v[len(v) // 2], v[0]))
PY
