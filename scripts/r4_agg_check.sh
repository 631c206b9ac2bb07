set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4agg; mkdir -p $O
timeout -k 10 300 python -m pytest $R/tests/test_preagg_gpu.py $R/tests/test_prefetch_gpu.py $R/tests/test_fused_ops_gpu.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
grep -q " passed" $O/tests.log && ! grep -q failed $O/tests.log || exit 1
cd /tmp && export TMPDIR=/tmp
for cfg in "--n-hidden 512 --steps 300" "--config 2 --steps 300" "--n-hidden 1024 --steps 300" "--config 4 --steps 600" "--n-hidden 2048 --steps 300" ""; do
  python3 $R/bench.py $cfg --no-cpu-baseline --no-second-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('$cfg', d['value'], d['ms_per_step'])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k512 -o step -- python3 $R/bench.py --n-hidden 512 --steps 200 --warmup 20 --no-second-leg --no-cpu-baseline --no-kernel-timing > $O/k512.log 2>&1
grep -h "extract" $O/k512/*/step_kernel_stats.csv $O/k512/step_kernel_stats.csv 2>/dev/null | cut -c1-120
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k4096 -o step -- python3 $R/bench.py --steps 40 --warmup 5 --no-second-leg --no-cpu-baseline --no-kernel-timing > $O/k4096.log 2>&1
grep -h "extract" $O/k4096/*/step_kernel_stats.csv $O/k4096/step_kernel_stats.csv 2>/dev/null | cut -c1-120
find $O -name '*kernel_trace.csv' -delete
