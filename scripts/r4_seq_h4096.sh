set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4seq; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o step -- python3 $R/bench.py --steps 60 --warmup 10 --no-second-leg --no-cpu-baseline --no-kernel-timing > $O/run.log 2>&1 || exit 1
python3 $R/scripts/step_seq.py $(find $O/k -name 'step_kernel_trace.csv' | head -1) > $O/seq_h4096.txt 2>&1
grep '^{"metric"' $O/run.log | tail -1 | cut -c1-120
find $O -name '*kernel_trace.csv' -delete
cat $O/seq_h4096.txt
