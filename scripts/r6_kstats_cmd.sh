#!/bin/bash
# kernel call counts of an arbitrary python command under rocprofv3: r6_kstats_cmd.sh <tag> <python args...>
: ${GRAFT_REPO_ROOT:?run under gpurun}
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/r6/$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 "$@" > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
f=$(find $O -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:r['Name'])
for r in rows:
    if 'gist::' in r['Name']:
        print('%-90s calls %5s avg_us %8.2f' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
find $O -name '*kernel_trace.csv' -delete
