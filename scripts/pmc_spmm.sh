#!/bin/bash
# Dev tool: PMC counters of the LDS-staged SpMM (scripts/spmm_probe.py) -- run on the GPU box.
: ${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_spmm
mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -o run -- python3 $GRAFT_REPO_ROOT/scripts/spmm_probe.py $1 > $OUT/$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$OUT/*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:50] + ' grid=' + r['Grid_Size']
        if 'spmm' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for k, d in acc.items():
        print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
