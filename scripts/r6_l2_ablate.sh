#!/bin/bash
# GPU-side durations (rocprofv3 kernel trace) of the LDS-staged aggregation in dev builds that leave a phase out:
# r6_l2_ablate.sh <lib suffix>...   (gist_amd/libgist_<suffix>.so, built with GIST_EXTRA_FLAGS=-DL2_PROBE_...)
: ${GRAFT_REPO_ROOT:?run under gpurun}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for l in "$@"; do
  O=$R/gpurun_out/r6/l2_$l; rm -rf $O; mkdir -p $O
  GIST_LIB_PATH=$R/gist_amd/libgist_$l.so rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/scripts/r5_spmm_narrow_probe.py > $O/run.log 2>&1 || { tail -3 $O/run.log; exit 1; }
  echo "== $l"
  python3 - $O <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/k_kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'lds2' in r['Kernel_Name']:
        d[(r['Kernel_Name'].split('(')[0][-40:], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X'))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
import statistics
for k, v in sorted(d.items()):
    print(k, 'n', len(v), 'median %.2f us  p10 %.2f  p90 %.2f' % (statistics.median(v), sorted(v)[len(v)//10], sorted(v)[9*len(v)//10]))
PY
  rm -f $(find $O -name '*kernel_trace.csv')
done
