#!/bin/bash
# rocprofv3 kernel stats of one bench.py invocation: r5_kstats.sh <tag> <bench args...>
: ${GRAFT_REPO_ROOT:?run under gpurun}
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/${ROUND:-r6}/$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/bench.py "$@" --no-cpu-baseline --no-second-leg --no-kernel-timing > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
grep '^{"metric"' $O/run.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d.get('module_path',{}).get('ms_per_step'))"
f=$(find $O -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:22]:
    print('%-70s calls %6s avg_us %8.2f total_ms %9.2f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
# keep the trace small: drop the per-launch csv unless asked
[ -n "$KEEP_TRACE" ] || find $O -name '*kernel_trace.csv' -delete
