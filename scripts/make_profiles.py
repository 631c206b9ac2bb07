"""Turn gpurun_out/final (scripts/final_profile.sh) into the files kept under profiles/."""
import collections, csv, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, 'gpurun_out', 'final')
P = os.path.join(ROOT, 'profiles')


def last_json(path):
    return json.loads([l for l in open(path) if l.startswith('{')][-1])


def short(name):
    name = re.sub(r'^void ', '', name)
    return name.split('(')[0]


# 1. bench lines
lines = {'bench_n1': last_json(os.path.join(F, 'bench_n1.log')),
         'per_rank_width_emulation': {('n_hidden_%d' % h): last_json(os.path.join(F, 'bench_h%d.log' % h))
                                      for h in (2048, 1024, 512)},
         'note': 'bench.py defaults on 1x MI355X (GEMM mode f16x3; `f32_mfma` = the same workload '
                 're-timed with every projection on the fp32 matrix cores, same process). '
                 'per_rank_width_emulation: `bench.py --n-hidden H/N` on one GPU = the per-rank work of '
                 'the N = 2/4/8 points (no collective); the multi-GPU numbers are the driver\'s.'}
json.dump(lines, open(os.path.join(P, 'r01_bench_lines.json'), 'w'), indent=1)

# 2. kernel stats of the same command
ks = os.path.join(F, 'kstats', 'step_kernel_stats.csv')
out = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'profile_report.py'), ks,
                      os.path.join(F, 'kstats_run.log'),
                      '`python3 bench.py --no-cpu-baseline --no-f32-rerun` (150 steps + 10 warm-up, GEMM mode f16x3, 1x MI355X)'],
                     capture_output=True, text=True, check=True).stdout
open(os.path.join(P, 'r01_bench_n1_kernel_stats.md'), 'w').write(out)
open(os.path.join(P, 'r01_bench_n1_kernel_stats.csv'), 'w').write(open(ks).read())


# 3. PMC tables
def table(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        d[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return d


fetch = table(os.path.join(F, 'pmc_FETCH_SIZE', 't_counter_collection.csv'))
write = table(os.path.join(F, 'pmc_WRITE_SIZE', 't_counter_collection.csv'))
rows = []
for k in fetch:
    f = fetch[k]['FETCH_SIZE']
    w = write.get(k, {}).get('WRITE_SIZE', [0.0])
    rows.append((sum(f), k, len(f), sum(f) / len(f), sum(w) / max(len(w), 1)))
rows.sort(reverse=True)
md = ['# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), `bench.py --steps 20 --warmup 3 '
      '--no-cpu-baseline --no-f32-rerun --no-kernel-timing`, GEMM mode f16x3, round 1', '',
      'Raw counter averages per dispatch (KB). gfx950 correction: fabric read bytes = 2 x FETCH_SIZE for wide '
      'coalesced streams (MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact.', '',
      '| kernel | dispatches | FETCH_SIZE avg KB (raw) | WRITE_SIZE avg KB |', '|---|---|---|---|']
for tot, k, n, fa, wa in rows[:24]:
    md.append('| `%s` | %d | %.1f | %.1f |' % (k, n, fa, wa))
open(os.path.join(P, 'r01_pmc_fetch_write.md'), 'w').write('\n'.join(md) + '\n')


def traffic(kname):          # every instantiation whose name starts with kname
    f = [x for k in fetch if k.startswith(kname) for x in fetch[k]['FETCH_SIZE']]
    w = [x for k in write if k.startswith(kname) for x in write[k]['WRITE_SIZE']]
    return dict(launches=len(f), FETCH_SIZE_KB_raw=round(sum(f) / len(f), 1),
                WRITE_SIZE_KB=round(sum(w) / len(w), 1),
                hbm_bytes_corrected=int((2 * sum(f) / len(f) + sum(w) / len(w)) * 1024))


sp = {k: traffic(k) for k in fetch if k.startswith('gist::spmm_csr_rowsplit_kernel')}   # per VEC/accumulate instantiation
tot_l = sum(v['launches'] for v in sp.values())
json.dump({'hbm_bytes_per_launch': int(sum(v['hbm_bytes_corrected'] * v['launches'] for v in sp.values()) / tot_l),
           'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 20 '
                     '--warmup 3`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 128-B requests '
                     'at 64 B, MI355X_MICROARCH.md section HBM; WRITE_SIZE exact), averaged over all SpMM launches of '
                     'the step (3 forward + 2 backward)',
           'kernels': sp, 'round': 1}, open(os.path.join(P, 'spmm_traffic.json'), 'w'), indent=1)
h3 = traffic('gist::gemm_h3_kernel')
json.dump({'hbm_bytes_per_launch': h3['hbm_bytes_corrected'], 'method': 'as profiles/spmm_traffic.json; average over '
           'the 5 split-GEMM launches of a step (3 x 137 GFLOP, 2 x 20 GFLOP)', 'kernel': h3, 'round': 1},
          open(os.path.join(P, 'gemm_h3_traffic.json'), 'w'), indent=1)

# 4. MFMA occupancy / clock of the split GEMM
mf = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(os.path.join(F, 'pmc_mfma', 't_counter_collection.csv'))):
    if 'gemm_h3_kernel' in r['Kernel_Name'] or 'gemm_f32_kernel' in r['Kernel_Name']:
        key = (short(r['Kernel_Name']), int(r['Grid_Size']) // 256)
        mf[key][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            dur[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
md = ['# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES, `bench.py '
      '--steps 20 --warmup 3 ...`, GEMM mode f16x3, round 1', '',
      'Per dispatch medians. cycles = GRBM_GUI_ACTIVE / 8 XCDs; MFMA pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / '
      '(1024 SIMDs x cycles); clock = cycles / duration -- indicative only: the quotient reads high on dispatches '
      'shorter than ~0.3 ms (MI355X_MICROARCH.md, DVFS give-back), see the sub-100-us rows.', '',
      '| kernel | workgroups | dispatches | duration us | cycles | MFMA occupancy | clock GHz | LDS bank conflicts |',
      '|---|---|---|---|---|---|---|---|']
for key in sorted(mf, key=lambda k: -sum(dur[k])):
    c = {k: sorted(v)[len(v) // 2] for k, v in mf[key].items()}
    d_us = sorted(dur[key])[len(dur[key]) // 2]
    cyc = c['GRBM_GUI_ACTIVE'] / 8
    md.append('| `%s` | %d | %d | %.1f | %.0f | %.3f | %.2f | %.0f |' % (
        key[0], key[1], len(dur[key]), d_us, cyc, c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc,
        cyc / d_us / 1e3, c.get('SQ_LDS_BANK_CONFLICT', 0)))
open(os.path.join(P, 'r01_pmc_mfma.md'), 'w').write('\n'.join(md) + '\n')

# 5. split GEMM: error against float64 and time against the fp32 kernel
for src, dst in (('h3_err.log', 'r01_gemm_h3_error_vs_float64.txt'), ('h3_bench.log', 'r01_gemm_h3_vs_f32_bench.txt')):
    txt = ''.join(l for l in open(os.path.join(F, src)) if 'amdgpu.ids' not in l)
    open(os.path.join(P, dst), 'w').write(txt)

# 6. full-graph evaluation
ev = last_json(os.path.join(F, 'eval.log'))
old = json.load(open(os.path.join(P, 'r01_eval_fullgraph.json')))
old.update(ev)
old['note_gemm_mode'] = ('eval_forward_H4096_s with the projections in GEMM mode f16x3 (0.318 s in mode f32: '
                         'the H-wide projections of 233 k rows are 18 TFLOP)')
json.dump(old, open(os.path.join(P, 'r01_eval_fullgraph.json'), 'w'))
print(open(os.path.join(P, 'r01_pmc_mfma.md')).read())
print(json.dumps(h3))
