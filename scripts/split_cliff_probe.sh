#!/bin/bash
# Dev tool: do launches of one projection shape run with different k-slice counts (grid z) from batch to batch?
# usage (gpurun): bash scripts/split_cliff_probe.sh <bench.py args>
: ${GRAFT_REPO_ROOT:?run under gpurun}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/scp
rocprofv3 --kernel-trace --output-format csv -d /tmp/scp -o t -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --steps 200 --warmup 20 --no-second-leg --no-cpu-baseline --no-kernel-timing > /tmp/scp.log 2>&1
python3 - <<PY
import csv, collections
rows=[r for r in csv.DictReader(open("/tmp/scp/t_kernel_trace.csv")) if "gemm" in r["Kernel_Name"]]
c=collections.defaultdict(list)
for r in rows: c[(r["Kernel_Name"].split("(")[0][-44:], int(r["Grid_Size_X"]), int(r.get("Grid_Size_Y") or 1), int(r.get("Grid_Size_Z") or 1))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(c.items()): print("%-46s grid x %8d y %3d z %3d  launches %5d  avg %7.1f us" % (k + (len(v), sum(v)/len(v))))
PY
