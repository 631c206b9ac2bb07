: ${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=<repo root>)}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/scripts/spmm_bench.py > $O/spmm_bench.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv -d $O/spmm_ktrace -o t -- python3 $R/scripts/spmm_probe.py 4096 1024 512 > $O/spmm_probe.log 2>&1
python3 $R/scripts/ktrace_summary.py $O/spmm_ktrace/t_kernel_trace.csv spmm > $O/spmm_kernel_times.txt
$R/scripts/_build/lds_gather_probe > $O/lds_gather_probe.log 2>&1
python3 $R/scripts/h3_bench.py 10 > $O/gemm_modes_bench.log 2>&1 || exit 1
python3 $R/scripts/eval_bench.py > $O/eval.log 2>&1 || exit 1
rm -rf $O/spmm_ktrace
tail -1 $O/eval.log | cut -c1-200
