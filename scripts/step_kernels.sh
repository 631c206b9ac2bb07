#!/bin/bash
# Dev tool: per-kernel time of the training step at a given width (rocprofv3 --kernel-trace --stats).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?}
EXTRA="${@:2}"
H=${1:-512}
rm -rf /tmp/sk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sk -o t -- python3 $R/bench.py --n-hidden $H $EXTRA --steps 200 --warmup 20 --no-second-leg --no-cpu-baseline --no-kernel-timing > /tmp/sk.log 2>&1
grep '^{"metric"' /tmp/sk.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'])"
python3 - <<PY
import csv
rows = list(csv.DictReader(open('/tmp/sk/t_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time per step (us): %.1f' % (tot / 220 / 1e3))
for r in rows[:40]:
    print('%-70s calls/step %5.2f  avg %7.2f us  per step %7.2f us  %5.1f%%' % (r['Name'][:70], int(r['Calls']) / 220.0, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 220 / 1e3, float(r['Percentage'])))
PY
