#!/usr/bin/env python3
"""Dev tool: the kernel SEQUENCE of one training step from a rocprofv3 --kernel-trace CSV of bench.py: per launch
(in stream order) name, grid, duration and the gap to the previous kernel's end -- medians over the traced steps.
    python scripts/step_seq.py <kernel_trace.csv> [last_kernel_substring]
A step ENDS with every launch whose name contains the substring (default: `adam_`, the optimiser launch -- with the next
batch prefetched, `adam_extract_kernel`, there is no extraction launch at a step's start to look for)."""
import csv
import sys
import collections

rows = list(csv.DictReader(open(sys.argv[1])))
last = sys.argv[2] if len(sys.argv) > 2 else 'adam_'
rows.sort(key=lambda r: int(r['Start_Timestamp']))
steps, cur = [], []
for r in rows:
    cur.append(r)
    if last in r['Kernel_Name']:
        steps.append(cur)
        cur = []
steps = steps[1:]                                       # (the first "step" carries the set-up launches)
if not steps:
    sys.exit('no step found')
lens = collections.Counter(len(s) for s in steps)
L = lens.most_common(1)[0][0]
steps = [s for s in steps if len(s) == L]
steps = steps[5:] if len(steps) > 10 else steps        # (skip the first few: warm-up)
print('%d steps of %d launches' % (len(steps), L))
tot_k = tot_g = 0.0
for i in range(L):
    durs = sorted((int(s[i]['End_Timestamp']) - int(s[i]['Start_Timestamp'])) / 1e3 for s in steps)
    gaps = sorted((int(s[i]['Start_Timestamp']) - int(s[i - 1]['End_Timestamp'])) / 1e3 for s in steps) if i else [0.0]
    r = steps[0][i]
    name = r['Kernel_Name'].split('(')[0].replace('gist::', '').replace('void ', '')[:52]
    d, g = durs[len(durs) // 2], gaps[len(gaps) // 2]
    tot_k += d
    tot_g += g
    print('%2d %-52s grid %7s x%-3s wg %4s  %7.2f us  gap %5.2f' % (i, name, r['Grid_Size_X'], r.get('Grid_Size_Y', ''),
                                                                     r['Workgroup_Size_X'], d, g))
span = sorted((int(s[-1]['End_Timestamp']) - int(s[0]['Start_Timestamp'])) / 1e3 for s in steps)
print('kernel time %.1f us + gaps %.1f us; first start -> last end median %.1f us' % (tot_k, tot_g, span[len(span) // 2]))
