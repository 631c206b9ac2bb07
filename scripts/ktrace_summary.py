import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r['Kernel_Name']: continue
    k = r['Kernel_Name'].split('(')[0][-40:] + ' grid=%7s wg=%4s' % (r['Grid_Size_X'], r['Workgroup_Size_X'])
    d[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
if len(sys.argv) > 3 and sys.argv[3] == 'seq':
    # launches in order, grouped in runs of identical configuration
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    run = []
    for r in rows + [None]:
        key = None if r is None else (r['Grid_Size_X'],)
        if run and (r is None or key != run[0][0]):
            ts = sorted(t for _, t in run)
            print('  grid=%8s n=%3d median %7.1f us min %7.1f' % (run[0][0][0], len(ts), ts[len(ts) // 2], ts[0]))
            run = []
        if r is not None:
            run.append((key, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
    sys.exit(0)
for k, v in d.items():
    v.sort()
    print('  %-70s n=%4d median %8.1f us  min %8.1f' % (k, len(v), v[len(v) // 2], v[0]))
