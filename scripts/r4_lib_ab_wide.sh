set -o pipefail
OLD=$1
for rep in 1 2; do
for cfg in "" "--n-hidden 2048 --steps 300"; do
  for lib in old new; do
    if [ $lib = old ]; then export GIST_LIB_PATH=$OLD; else unset GIST_LIB_PATH; fi
    python bench.py $cfg --no-cpu-baseline --no-second-leg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib | $cfg |', d['value'], d['ms_per_step'], d['loss_last'])"
  done
done
done
