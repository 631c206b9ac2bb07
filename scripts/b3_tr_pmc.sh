#!/bin/bash
# Dev tool (GPU box): kernel durations and LDS bank conflicts of gemm_b3_kernel<false> / <true> (scripts/b3_tr_probe.py)
: ${GRAFT_REPO_ROOT:?run under gpurun}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b3tr; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o t -- python3 $R/scripts/b3_tr_probe.py > $O/kt.log 2>&1
python3 $R/scripts/ktrace_summary.py $O/kt/t_kernel_trace.csv gemm_b3_kernel
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc -o t -- python3 $R/scripts/b3_tr_probe.py > $O/pmc.log 2>&1
python3 - <<PY
import csv, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('$O/pmc/t_counter_collection.csv')):
    if 'gemm_b3_kernel' in r['Kernel_Name'] and r['Grid_Size'] in ('262144',):
        d[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in d.items():
    print(k, {c: sum(x) / len(x) for c, x in v.items()})
PY
