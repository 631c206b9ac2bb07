#!/bin/bash
# On the GPU box (gpurun): the round-3 measurements kept under profiles/ -- bench line of the default
# workload (bf16x3 headline + fp32 / f16x3 legs + cpu baseline), kernel stats of the same workload, PMC passes
# (FETCH_SIZE, WRITE_SIZE, MFMA busy: separate runs, counters only), the per-rank widths of the N = 2/4/8 points
# with their kernel stats (h = 1024, 512: the fused step), one bench line per BASELINE config (2, 4, 5), the
# convert-on-load GEMM micro-benchmark, full-graph evaluation (+ PMC traffic of its aggregation).
: ${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r3
part=${1:-all}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ $part = all ] || [ $part = a ]; then
python3 $R/bench.py > $O/bench_n1.log 2>&1 || exit 1
grep '^{"metric"' $O/bench_n1.log | tail -1 | cut -c1-160
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -o step -- python3 $R/bench.py --no-cpu-baseline --no-second-leg > $O/kstats_run.log 2>&1 || exit 1
echo kstats done
for mode in bf16x3 f32; do
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_$mode -o t -- python3 $R/bench.py --gemm-mode $mode --steps 20 --warmup 3 --no-cpu-baseline --no-second-leg --no-kernel-timing > $O/pmc_${c}_$mode.log 2>&1 || exit 1
  echo pmc $c $mode done
done
done
for mode in bf16x3 f32; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $O/pmc_mfma_$mode -o t -- python3 $R/bench.py --gemm-mode $mode --steps 20 --warmup 3 --no-cpu-baseline --no-second-leg --no-kernel-timing > $O/pmc_mfma_$mode.log 2>&1 || exit 1
done
echo pmc mfma done
fi
if [ $part = all ] || [ $part = b ]; then
for h in 2048 1024 512; do
  python3 $R/bench.py --n-hidden $h --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_h$h.log 2>&1 || exit 1
done
echo widths done
for h in 1024 512; do
  rm -rf $O/kstats_h$h
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_h$h -o step -- python3 $R/bench.py --n-hidden $h --steps 200 --warmup 20 --no-second-leg --no-cpu-baseline --no-kernel-timing > $O/kstats_h$h.log 2>&1 || exit 1
done
echo width kstats done
python3 $R/bench.py --config 2 --steps 300 --warmup 20 > $O/bench_cfg2.log 2>&1 || exit 1
python3 $R/bench.py --config 4 --steps 600 --warmup 20 > $O/bench_cfg4.log 2>&1 || exit 1
python3 $R/bench.py --config 5 --no-second-leg > $O/bench_cfg5.log 2>&1 || exit 1
echo configs done
for c in 2 4; do
  rm -rf $O/kstats_cfg$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_cfg$c -o step -- python3 $R/bench.py --config $c --steps 200 --warmup 20 --no-second-leg --no-cpu-baseline --no-kernel-timing > $O/kstats_cfg$c.log 2>&1 || exit 1
done
echo config kstats done
fi
if [ $part = all ] || [ $part = c ]; then
python3 $R/scripts/b3c_bench.py 0 0 64 1 64 2 128 1 128128 1 128128 2 > $O/b3c_bench.log 2>&1 || exit 1
python3 $R/scripts/eval_bench.py > $O/eval.log 2>&1 || exit 1
grep '^{' $O/eval.log | tail -1 | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_eval_$c -o t -- python3 $R/scripts/eval_spmm_pmc.py > $O/pmc_eval_$c.log 2>&1 || exit 1
done
echo eval pmc done
fi
find $O -name '*kernel_trace.csv' -size +20M -delete
ls $O
