#!/bin/bash
# On the GPU box (gpurun): the measurements kept under profiles/ -- bench line, kernel stats of
# the same command, PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy / active cycles: separate
# runs, counters only), full-graph evaluation, per-rank widths of the N = 2/4/8 points.
: ${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_n1.log 2>&1 || exit 1
tail -1 $O/bench_n1.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -o step -- python3 $R/bench.py --no-cpu-baseline --no-f32-rerun > $O/kstats_run.log 2>&1 || exit 1
echo kstats done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o t -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-rerun --no-kernel-timing > $O/pmc_$c.log 2>&1 || exit 1
  echo pmc $c done
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $O/pmc_mfma -o t -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-rerun --no-kernel-timing > $O/pmc_mfma.log 2>&1 || exit 1
echo pmc mfma done
for h in 2048 1024 512; do
  python3 $R/bench.py --n-hidden $h --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_h$h.log 2>&1 || exit 1
done
echo widths done
python3 $R/scripts/eval_bench.py > $O/eval.log 2>&1 || exit 1
python3 $R/scripts/h3_error_probe.py > $O/h3_err.log 2>&1 || exit 1
python3 $R/scripts/h3_bench.py 10 > $O/h3_bench.log 2>&1 || exit 1
tail -1 $O/eval.log | cut -c1-400
# keep the merged output small: drop the raw traces, keep stats and counter tables
find $O -name '*kernel_trace.csv' -size +20M -delete
ls -la $O $O/*/ | head -60
