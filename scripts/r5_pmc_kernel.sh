#!/bin/bash
# rocprofv3 --pmc <counters...> on a short bench run; per-kernel mean of every counter.  r5_pmc_kernel.sh <tag> "<counters>" <bench args...>
: ${GRAFT_REPO_ROOT:?run under gpurun}
R=$GRAFT_REPO_ROOT
tag=$1; ctr=$2; shift; shift
O=$R/gpurun_out/r5/pmc_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --output-format csv -d $O -o t -- python3 $R/bench.py "$@" --steps 12 --warmup 3 --no-cpu-baseline --no-second-leg --no-module-leg --no-kernel-timing > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
f=$(find $O -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].split('(')[0][-60:]
    acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name, c in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    if 'gemm_b3' in name or 'spmm' in name:
        print(name, {k: (len(v), round(sum(v) / len(v), 1)) for k, v in c.items()})
PY
