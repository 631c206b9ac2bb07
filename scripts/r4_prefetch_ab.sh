#!/bin/bash
# round 4: next-batch extraction beside the optimiser launch (bench.py, GIST_BENCH_PREFETCH=0 / 1)
set -o pipefail
mkdir -p gpurun_out/r4
for cfg in "--n-hidden 512 --steps 300" "--config 2 --steps 300" "--n-hidden 1024 --steps 300" "--config 4 --steps 300" ""; do
  for pf in 0 1; do
    tag=$(echo "pf${pf}_${cfg}" | tr -c 'a-zA-Z0-9_\n' '_')
    GIST_BENCH_PREFETCH=$pf timeout -k 10 300 python bench.py $cfg --no-cpu-baseline --no-second-leg > gpurun_out/r4/$tag.json 2> gpurun_out/r4/$tag.err || { tail -5 gpurun_out/r4/$tag.err; exit 1; }
    python - "$tag" "$cfg" $pf <<'PY'
import json, sys
d=json.loads(open('gpurun_out/r4/%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print('prefetch', sys.argv[3], '| %-28s |' % sys.argv[2], d['value'], d['ms_per_step'], d.get('loss_last'))
PY
  done
done
