#!/bin/bash
# GPU box: kernel-level durations (rocprofv3 --kernel-trace) of scripts/r4_kbench.py, grouped by kernel and grid
: ${GRAFT_REPO_ROOT:?}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4/kbench_$1
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python3 $R/scripts/r4_kbench.py $2 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 - <<PY > $O/summary.txt
import csv, collections, glob
f = glob.glob('$O/**/run_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# consecutive runs of the same (kernel, grid): one line per run (the bench calls each variant 65 times in a row)
run = []
def flush():
    if len(run) >= 20:
        ts = sorted(t for _, t in run)
        print('%-60s grid %8s x%-3s n=%3d  median %7.2f us  min %7.2f' % (run[0][0][0][:60], run[0][0][1], run[0][0][2], len(ts), ts[len(ts)//2], ts[0]))
for r in rows + [None]:
    key = None if r is None else (r['Kernel_Name'].split('(')[0].replace('gist::','').replace('void ','').replace('(anonymous namespace)::',''), r['Grid_Size_X'], r['Grid_Size_Y'])
    if run and key != run[0][0]:
        flush(); run = []
    if r is not None:
        run.append((key, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
find $O -name '*.csv' -delete
cat $O/run.log | grep -v amdgpu.ids
cat $O/summary.txt
