#!/bin/bash
# On the GPU box (gpurun): the round-6 measurements kept under profiles/ (scripts/make_profiles_r6.py turns them into
# the committed files).  part a: bench line of the default workload (+ legs + cpu baseline over one full epoch), kernel
# stats of the same workload, PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy: separate runs, counters only).
# part b: per-rank widths 2048/1024/512 and BASELINE configs 2, 4, 5 as bench lines; kernel stats + per-launch step
# sequences of h = 1024, 512, configs 2 and 4.  part c: SQ counters of the narrow steps (what bounds the small kernels).
: ${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r6
part=${1:-all}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ $part = all ] || [ $part = a ]; then
python3 $R/bench.py > $O/bench_n1.log 2>&1 || exit 1
grep '^{"metric"' $O/bench_n1.log | tail -1 | cut -c1-160
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -o step -- python3 $R/bench.py --no-cpu-baseline --no-second-leg --no-module-leg > $O/kstats_run.log 2>&1 || exit 1
echo kstats done
for mode in bf16x3 f32; do
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_$mode -o t -- python3 $R/bench.py --gemm-mode $mode --steps 20 --warmup 3 --no-cpu-baseline --no-second-leg --no-module-leg --no-kernel-timing > $O/pmc_${c}_$mode.log 2>&1 || exit 1
  echo pmc $c $mode done
done
done
for mode in bf16x3 f32; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $O/pmc_mfma_$mode -o t -- python3 $R/bench.py --gemm-mode $mode --steps 20 --warmup 3 --no-cpu-baseline --no-second-leg --no-module-leg --no-kernel-timing > $O/pmc_mfma_$mode.log 2>&1 || exit 1
done
echo pmc mfma done
fi
if [ $part = all ] || [ $part = b ]; then
for h in 2048 1024 512; do
  python3 $R/bench.py --n-hidden $h --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_h$h.log 2>&1 || exit 1
done
echo widths done
python3 $R/bench.py --config 2 --steps 300 --warmup 20 > $O/bench_cfg2.log 2>&1 || exit 1
python3 $R/bench.py --config 4 --steps 600 --warmup 20 > $O/bench_cfg4.log 2>&1 || exit 1
python3 $R/bench.py --config 5 --no-second-leg > $O/bench_cfg5.log 2>&1 || exit 1
echo configs done
for sub in h1024:--n-hidden:1024 h512:--n-hidden:512 cfg2:--config:2 cfg4:--config:4; do
  tag=${sub%%:*}; rest=${sub#*:}; flag=${rest%%:*}; val=${rest#*:}
  rm -rf $O/kstats_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_$tag -o step -- python3 $R/bench.py $flag $val --steps 200 --warmup 20 --no-second-leg --no-module-leg --no-cpu-baseline --no-kernel-timing > $O/kstats_$tag.log 2>&1 || exit 1
  python3 $R/scripts/step_seq.py $(find $O/kstats_$tag -name 'step_kernel_trace.csv' | head -1) > $O/seq_$tag.txt 2>&1
done
echo kstats widths done
fi
if [ $part = all ] || [ $part = c ]; then
for sub in h512:--n-hidden:512 cfg2:--config:2 h4096:--n-hidden:4096; do
  tag=${sub%%:*}; rest=${sub#*:}; flag=${rest%%:*}; val=${rest#*:}
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_$tag -o t -- python3 $R/bench.py $flag $val --steps 20 --warmup 3 --no-cpu-baseline --no-second-leg --no-module-leg --no-kernel-timing > $O/pmc_sq_$tag.log 2>&1 || exit 1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$tag -o t -- python3 $R/bench.py $flag $val --steps 20 --warmup 3 --no-cpu-baseline --no-second-leg --no-module-leg --no-kernel-timing > $O/pmc_fetch_$tag.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_lds_$tag -o t -- python3 $R/bench.py $flag $val --steps 20 --warmup 3 --no-cpu-baseline --no-second-leg --no-module-leg --no-kernel-timing > $O/pmc_lds_$tag.log 2>&1 || exit 1
done
echo sq pmc done
fi
if [ $part = all ] || [ $part = d ]; then
# the unplanted graph beside the planted one (same box), widths 4096 and 512; the partitioner on this box's host
bash $R/scripts/r6_unplanted.sh 4096 512 || exit 1
PYTHONPATH=$R python3 $R/scripts/partition_quality.py all $R/gpurun_out/r6/partitioner_box.json > $O/partitioner.log 2>&1 || true
fi
find $O -name '*kernel_trace.csv' -size +1M -delete
ls $O
