"""One training step launch by launch from a rocprofv3 kernel trace (csv): name, duration, gap to the previous kernel."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# a step starts at an extraction (extract_parts_kernel / adam_extract_kernel carries the NEXT step's) -- use the optimiser as the end
ends = [i for i, n in enumerate(names) if 'adam' in n]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) // 2
a, b = ends[which - 1] + 1, ends[which] + 1
prev_end = int(rows[a - 1]['End_Timestamp'])
tot = 0.0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-64s %8.1f us  gap %6.1f' % (r['Kernel_Name'][:64], (e - s) / 1e3, (s - prev_end) / 1e3))
    tot += (e - s) / 1e3
    prev_end = e
print('kernels %.1f us, span %.1f us, %d launches' % (tot, (int(rows[b - 1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3, b - a))
