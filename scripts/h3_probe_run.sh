#!/bin/bash
# on the GPU box: kernel-trace each probe build on the big shapes, print gemm_h3_kernel durations
: ${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "" _NO_DMA _NO_MFMA $EXTRA_VARIANTS; do
  export GIST_LIB_PATH=$R/gist_amd/libgist_hip$v.so
  rm -rf /tmp/p$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p$v -o t -- python $R/scripts/h3_bench.py 3 > /tmp/p$v.log 2>&1
  echo "== variant '$v'"
  python - <<PY
import csv
rows=[r for r in csv.DictReader(open('/tmp/p$v/t_kernel_trace.csv')) if 'gemm_h3_kernel' in r['Kernel_Name']]
import collections
d=collections.defaultdict(list)
for r in rows: d[int(r['Grid_Size_X'])//256].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for g,v in sorted(d.items()): v.sort(); print('  blocks %5d  n=%3d  median %.1f us  min %.1f'%(g,len(v),v[len(v)//2],v[0]))
PY
done
