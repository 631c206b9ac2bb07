#!/bin/bash
: ${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in bf16x3 f16x3; do
rm -rf /tmp/pk /tmp/pp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pk -o t -- python3 $R/scripts/b3_probe.py $mode > /tmp/pk.log 2>&1
python3 $R/scripts/ktrace_summary.py /tmp/pk/t_kernel_trace.csv gist
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pp -o t -- python3 $R/scripts/b3_probe.py $mode > /tmp/pp.log 2>&1
python3 - <<PY
import csv, collections, glob
f = glob.glob('/tmp/pp/*counter_collection.csv')[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'gemm_' in r['Kernel_Name']:
        d[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for key, c in sorted(d.items()):
    m = {k: sorted(v)[len(v)//2] for k, v in c.items()}
    print(key, {k: '%.3g' % v for k, v in m.items()})
    if 'GRBM_GUI_ACTIVE' in m and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
        cyc = m['GRBM_GUI_ACTIVE'] / 8
        print('    cycles/XCD %.0f  mfma busy per SIMD %.0f  -> mfma util %.3f' % (cyc, m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024, m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc))
PY
done
