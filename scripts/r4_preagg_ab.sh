#!/bin/bash
# round 4: layer 0's aggregation formed by the extraction (GIST_STEP_PREAGG=0 / 1), bench.py A/B
set -o pipefail
mkdir -p gpurun_out/r4pre
for rep in 1 2; do
for cfg in "--n-hidden 512 --steps 300" "--config 2 --steps 300" "--n-hidden 1024 --steps 300" "--config 4 --steps 300" ""; do
  for pre in 0 1; do
    tag=$(echo "pre${pre}_${rep}_${cfg}" | tr -c 'a-zA-Z0-9_\n' '_')
    GIST_STEP_PREAGG=$pre timeout -k 10 300 python bench.py $cfg --no-cpu-baseline --no-second-leg > gpurun_out/r4pre/$tag.json 2> gpurun_out/r4pre/$tag.err || { tail -5 gpurun_out/r4pre/$tag.err; exit 1; }
    python - "$tag" "$cfg" $pre <<'PY'
import json, sys
d=json.loads(open('gpurun_out/r4pre/%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print('preagg', sys.argv[3], '| %-28s |' % sys.argv[2], d['value'], d['ms_per_step'], d.get('loss_last'))
PY
  done
done
done
