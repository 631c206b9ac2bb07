#!/bin/bash
# HIP API + kernel trace of one bench.py invocation; lists the API calls that took longer than 5 ms.
: ${GRAFT_REPO_ROOT:?run under gpurun}
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/r5/$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $O -o t -- python3 $R/bench.py "$@" --no-cpu-baseline --no-second-leg --no-kernel-timing > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
grep '^{"metric"' $O/run.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); m=d.get('module_path',{}); print('$tag', d['ms_per_step'], d['host_issue_ms_per_step']['over_1ms'][:8], 'module', m.get('ms_per_step'), (m.get('host_issue_ms_per_step') or {}).get('over_1ms'))"
ls $O
python3 - $O <<'PY'
import csv,sys,glob
O=sys.argv[1]
api=glob.glob(O+'/*hip_api_trace.csv')
rows=list(csv.DictReader(open(api[0])))
print(len(rows),'api rows; columns',list(rows[0].keys()))
t0=min(int(r['Start_Timestamp']) for r in rows)
big=[r for r in rows if int(r['End_Timestamp'])-int(r['Start_Timestamp'])>5e6]
for r in big[-40:]:
    print('%-32s start %10.3f ms dur %8.3f ms tid %s' % (r['Function'], (int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, r.get('Thread_Id')))
PY
find $O -name '*_trace.csv' -size +40M -delete
