#!/bin/bash
# On the GPU box (gpurun): the round-2 measurements kept under profiles/ -- bench line (bf16x3
# headline + fp32 / f16x3 legs + cpu baseline), kernel stats of the same workload, PMC passes
# (FETCH_SIZE, WRITE_SIZE, MFMA busy / active cycles: separate runs, counters only), per-rank
# widths of the N = 2/4/8 points, SpMM / GEMM micro-benchmarks and in-kernel probes, full-graph
# evaluation.  Probe libraries (built here beforehand, they travel with the snapshot):
#   GIST_EXTRA_FLAGS=-DMF_PROBE GIST_LIB_OUT=$PWD/gist_amd/libgist_hip_MFP.so python gist_amd/build.py
#   GIST_EXTRA_FLAGS=-DB3_CLOCK_PROBE GIST_LIB_OUT=$PWD/gist_amd/libgist_hip_CLK.so python gist_amd/build.py
: ${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_n1.log 2>&1 || exit 1
tail -1 $O/bench_n1.log | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -o step -- python3 $R/bench.py --no-cpu-baseline --no-second-leg > $O/kstats_run.log 2>&1 || exit 1
echo kstats done
for mode in bf16x3 f32; do
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_$mode -o t -- python3 $R/bench.py --gemm-mode $mode --steps 20 --warmup 3 --no-cpu-baseline --no-second-leg --no-kernel-timing > $O/pmc_${c}_$mode.log 2>&1 || exit 1
  echo pmc $c $mode done
done
done
for mode in bf16x3 f32 f16x3; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $O/pmc_mfma_$mode -o t -- python3 $R/bench.py --gemm-mode $mode --steps 20 --warmup 3 --no-cpu-baseline --no-second-leg --no-kernel-timing > $O/pmc_mfma_$mode.log 2>&1 || exit 1
done
echo pmc mfma done
for h in 2048 1024 512; do
  python3 $R/bench.py --n-hidden $h --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_h$h.log 2>&1 || exit 1
done
echo widths done
python3 $R/scripts/spmm_bench.py > $O/spmm_bench.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv -d $O/ktrace_spmm -o run -- python3 $R/scripts/spmm_mf_probe.py 4096 2048 1024 512 > $O/ktrace_spmm.log 2>&1 || exit 1
python3 $R/scripts/ktrace_summary.py $O/ktrace_spmm/run_kernel_trace.csv spmm > $O/spmm_kernel_times.txt 2>&1
GIST_LIB_PATH=$R/gist_amd/libgist_hip_MFP.so python3 $R/scripts/spmm_mf_phases.py 4096 2048 > $O/spmm_mf_phases.log 2>&1 || exit 1
GIST_LIB_PATH=$R/gist_amd/libgist_hip_CLK.so python3 $R/scripts/b3_clock_probe.py > $O/b3_clock.log 2>&1 || exit 1
echo probes done
python3 $R/scripts/h3_bench.py 10 > $O/gemm_modes_bench.log 2>&1 || exit 1
python3 $R/scripts/eval_bench.py > $O/eval.log 2>&1 || exit 1
tail -1 $O/eval.log | cut -c1-300
find $O -name '*kernel_trace.csv' -size +20M -delete
ls $O
