#!/bin/bash
# round 4: where the prefetched extraction runs (bench.py GIST_BENCH_PREFETCH=0 none / grid = in the optimiser's grid /
# side = own launch on the library's second stream)
set -o pipefail
mkdir -p gpurun_out/r4side
for rep in 1 2; do
for cfg in "" "--n-hidden 2048 --steps 300" "--n-hidden 1024 --steps 300" "--n-hidden 512 --steps 300" "--config 2 --steps 300" "--config 4 --steps 300" "--config 5"; do
  for pf in 0 grid side; do
    tag=$(echo "pf${pf}_${rep}_${cfg}" | tr -c 'a-zA-Z0-9_\n' '_')
    GIST_BENCH_PREFETCH=$pf timeout -k 10 300 python bench.py $cfg --no-cpu-baseline --no-second-leg > gpurun_out/r4side/$tag.json 2> gpurun_out/r4side/$tag.err || { tail -5 gpurun_out/r4side/$tag.err; exit 1; }
    python - "$tag" "$cfg" $pf <<'PY'
import json, sys
d=json.loads(open('gpurun_out/r4side/%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print('prefetch %-4s' % sys.argv[3], '| %-28s |' % sys.argv[2], d['value'], d['ms_per_step'], d.get('loss_last'))
PY
  done
done
done
