"""Distribution of one kernel's launch durations in a rocprofv3 kernel trace: percentiles and the slowest launches."""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for pat in sys.argv[2:]:
    d = np.array([(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if pat in r['Kernel_Name']])
    idx = [i for i, r in enumerate(rows) if pat in r['Kernel_Name']]
    if d.size == 0:
        continue
    print('%-28s n=%4d  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us; launches over 2x the median: %d; the slowest at positions %s'
          % (pat, d.size, np.percentile(d, 10), np.median(d), np.percentile(d, 90), d.max(), int((d > 2 * np.median(d)).sum()),
             list(np.argsort(-d)[:6])))
