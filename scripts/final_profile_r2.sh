#!/bin/bash
# On the GPU box (gpurun): the round-2 measurements kept under profiles/ -- bench line (fp32
# headline + bf16x3 / f16x3 legs + cpu baseline), kernel stats of the same workload, PMC passes
# (FETCH_SIZE, WRITE_SIZE, MFMA busy / active cycles: separate runs, counters only), per-rank
# widths of the N = 2/4/8 points, SpMM / GEMM-mode micro-benchmarks, full-graph evaluation.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_n1.log 2>&1 || exit 1
tail -1 $O/bench_n1.log | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -o step -- python3 $R/bench.py --no-cpu-baseline --no-second-leg > $O/kstats_run.log 2>&1 || exit 1
echo kstats done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o t -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-second-leg --no-kernel-timing > $O/pmc_$c.log 2>&1 || exit 1
  echo pmc $c done
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $O/pmc_mfma -o t -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-second-leg --no-kernel-timing > $O/pmc_mfma.log 2>&1 || exit 1
for mode in bf16x3 f16x3; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $O/pmc_mfma_$mode -o t -- python3 $R/bench.py --gemm-mode $mode --steps 20 --warmup 3 --no-cpu-baseline --no-second-leg --no-kernel-timing > $O/pmc_mfma_$mode.log 2>&1 || exit 1
done
echo pmc mfma done
for h in 2048 1024 512; do
  python3 $R/bench.py --n-hidden $h --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_h$h.log 2>&1 || exit 1
done
echo widths done
python3 $R/scripts/spmm_bench.py > $O/spmm_bench.log 2>&1 || exit 1
$R/scripts/_build/lds_gather_probe > $O/lds_gather_probe.log 2>&1
python3 $R/scripts/h3_bench.py 10 > $O/gemm_modes_bench.log 2>&1 || exit 1
python3 $R/scripts/eval_bench.py > $O/eval.log 2>&1 || exit 1
tail -1 $O/eval.log | cut -c1-300
find $O -name '*kernel_trace.csv' -size +20M -delete
ls $O
