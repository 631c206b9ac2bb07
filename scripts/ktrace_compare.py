"""Per-kernel time per step of two rocprofv3 kernel traces side by side (A = planted, B = unplanted, ...):
ktrace_compare.py <trace_a.csv> <steps_a> <trace_b.csv> <steps_b>.  Only kernels launched at least steps/2 times count
(the set-up's one-off launches are left out); us per step = total duration / steps."""
import csv, sys, re
from collections import defaultdict


def load(path, steps):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('gist::', '')
        tot[name] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        cnt[name] += 1
    return {k: (tot[k] / steps, cnt[k] / steps, tot[k] / cnt[k]) for k in tot if cnt[k] >= steps / 2}


a = load(sys.argv[1], float(sys.argv[2]))
b = load(sys.argv[3], float(sys.argv[4]))
names = sorted(set(a) | set(b), key=lambda k: -(b.get(k, (0,))[0] - a.get(k, (0,))[0]))
print('%-44s %9s %9s %8s | launches/step A, B | avg us A, B' % ('kernel', 'A us/step', 'B us/step', 'B - A'))
sa = sb = 0.0
for k in names:
    ua, ca, ma = a.get(k, (0.0, 0.0, 0.0))
    ub, cb, mb = b.get(k, (0.0, 0.0, 0.0))
    sa += ua; sb += ub
    print('%-44s %9.2f %9.2f %+8.2f | %5.2f %5.2f | %7.2f %7.2f' % (k[:44], ua, ub, ub - ua, ca, cb, ma, mb))
print('%-44s %9.2f %9.2f %+8.2f' % ('total', sa, sb, sb - sa))
