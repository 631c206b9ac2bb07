#!/bin/bash
# on the GPU box: PMC counters of the split GEMM main kernels (MFMA busy cycles, active cycles)
: ${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmc1 -o t -- python $R/scripts/h3_bench.py 2 > /tmp/pmc1.log 2>&1
tail -3 /tmp/pmc1.log
python - <<PY
import csv, collections, glob
f = glob.glob('/tmp/pmc1/*counter_collection.csv')[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'gemm_h3' in r['Kernel_Name']:
        key = (r['Kernel_Name'].split('(')[0], int(r['Grid_Size']) // 256)
        d[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, c in sorted(d.items()):
    m = {k: sorted(v)[len(v)//2] for k, v in c.items()}
    print(key, {k: '%.3g' % v for k, v in m.items()})
    if 'GRBM_GUI_ACTIVE' in m and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
        cyc = m['GRBM_GUI_ACTIVE'] / 8
        print('    cycles/XCD %.0f  mfma busy per SIMD %.0f  -> mfma util %.3f' % (cyc, m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024, m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc))
PY
