"""Turn a rocprofv3 --kernel-trace --stats CSV + the bench line of the same run into the
markdown summary kept under profiles/.

    python scripts/profile_report.py <kernel_stats.csv> <bench_stdout.log> <title> > profiles/xxx.md
"""
import csv, json, sys

rows = list(csv.DictReader(open(sys.argv[1])))
line = [l for l in open(sys.argv[2]) if l.startswith('{"metric"')][-1].strip()
title = sys.argv[3]
print('# rocprofv3 --kernel-trace --stats: %s\n' % title)
print('bench line of the profiled run:\n\n```\n%s\n```\n' % line)
print('| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|')
gemm_calls = gemm_ns = spmm_calls = spmm_ns = h3_calls = h3_ns = pre_ns = 0
main_name = None
MAIN = ('gemm_f32_kernel', 'gemm_h3_kernel', 'gemm_b3_kernel', 'narrow_nn_drop_kernel', 'narrow_nt_kernel',
        'narrow_tn_kernel')
PRE = ('h3_', 'b3_', '_dual_split', 'splitk_reduce')        # split pre-pass kernels + split-K reduce
for r in rows:
    name, calls, tot = r['Name'], int(r['Calls']), float(r['TotalDurationNs'])
    is_main = any(m in name for m in MAIN)
    is_pre = (not is_main) and any(m in name for m in PRE)
    if is_main or is_pre:                  # every kernel a projection call launches, in every GEMM mode
        gemm_ns += tot
        if is_main:
            gemm_calls += calls
    if 'gemm_h3_kernel' in name or 'gemm_b3_kernel' in name:
        h3_calls += calls
        h3_ns += tot
        main_name = 'gist::gemm_b3_kernel' if 'gemm_b3_kernel' in name else 'gist::gemm_h3_kernel'
    elif is_pre and 'splitk_reduce' not in name:
        pre_ns += tot
    if 'spmm_csr' in name:
        spmm_calls += calls
        spmm_ns += tot
    if float(r['Percentage']) >= 0.02:
        short = name if len(name) <= 110 else name[:107] + '...'
        print('| `%s` | %d | %.3f | %.2f | %.2f |' % (short, calls, tot / 1e6, float(r['AverageNs']) / 1e3,
                                                  float(r['Percentage'])))
d = json.loads(line)
print()
print('GEMM (all layouts/tiles + split-K reduce + split pre-pass): %d GEMM launches, %.3f ms total, %.4f ms average per '
      'GEMM call -- compare the bench line\'s per-call figure %s (native HIP-event timer, sampled).'
      % (gemm_calls, gemm_ns / 1e6, gemm_ns / 1e6 / max(gemm_calls, 1),
         d.get('roofline', {}).get('all_projection_calls', {}).get('avg_call_ms',
                                                                   d.get('roofline', {}).get('avg_launch_ms'))))
if h3_calls:
    print('Split GEMM main kernel (%s): %%d launches, %%.3f ms total, %%.4f ms average -- '
          'compare `roofline.avg_launch_ms` = %%s; its split pre-pass kernels (*_dual_split / h3_split_rows / '
          'h3_colmax / h3_split_t): %%.3f ms total.' % main_name % (h3_calls, h3_ns / 1e6, h3_ns / 1e6 / h3_calls,
                                           d.get('roofline', {}).get('avg_launch_ms'), pre_ns / 1e6))
print('SpMM: %d launches, %.3f ms total, %.4f ms average -- compare `roofline_spmm.avg_launch_ms` = %s.'
      % (spmm_calls, spmm_ns / 1e6, spmm_ns / 1e6 / max(spmm_calls, 1),
         d.get('roofline_spmm', {}).get('avg_launch_ms')))
