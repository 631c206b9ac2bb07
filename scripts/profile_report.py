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
gemm_calls = gemm_ns = spmm_calls = spmm_ns = 0
for r in rows:
    name, calls, tot = r['Name'], int(r['Calls']), float(r['TotalDurationNs'])
    if 'gemm_f32' in name or 'splitk_reduce' in name:
        gemm_ns += tot
        if 'gemm_f32' in name:
            gemm_calls += calls
    if 'spmm_csr' in name:
        spmm_calls += calls
        spmm_ns += tot
    if float(r['Percentage']) >= 0.02:
        short = name if len(name) <= 110 else name[:107] + '...'
        print('| `%s` | %d | %.3f | %.2f | %.2f |' % (short, calls, tot / 1e6, float(r['AverageNs']) / 1e3,
                                                  float(r['Percentage'])))
d = json.loads(line)
print()
print('GEMM (all layouts/tiles + split-K reduce): %d GEMM launches, %.3f ms total, %.4f ms average per '
      'GEMM call -- compare `roofline.avg_launch_ms` = %s of the bench line (native HIP-event timer, '
      'sampled).' % (gemm_calls, gemm_ns / 1e6, gemm_ns / 1e6 / max(gemm_calls, 1),
                     d.get('roofline', {}).get('avg_launch_ms')))
print('SpMM: %d launches, %.3f ms total, %.4f ms average -- compare `roofline_spmm.avg_launch_ms` = %s.'
      % (spmm_calls, spmm_ns / 1e6, spmm_ns / 1e6 / max(spmm_calls, 1),
         d.get('roofline_spmm', {}).get('avg_launch_ms')))
