"""Turn gpurun_out/final_r2 (scripts/final_profile_r2.sh) into the round-2 files under profiles/."""
import collections, csv, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, 'gpurun_out', 'final_r2')
P = os.path.join(ROOT, 'profiles')
TAG = 'r02'


def last_json(path):
    return json.loads([l for l in open(path) if l.startswith('{')][-1])


def short(name):
    return re.sub(r'^void ', '', name).split('(')[0]


def clean(path):
    return ''.join(l for l in open(path) if 'amdgpu.ids' not in l)


# 1. bench lines
lines = {'bench_n1': last_json(os.path.join(F, 'bench_n1.log')),
         'per_rank_width_emulation': {('n_hidden_%d' % h): last_json(os.path.join(F, 'bench_h%d.log' % h))
                                      for h in (2048, 1024, 512)},
         'note': 'bench.py defaults on 1x MI355X: headline = large projections as three bf16 pieces per operand '
                 '(all 24 bits, six cross terms: GEMM mode bf16x3, the default); `f32_mfma` / `f16x3_split` = the same '
                 'workload re-timed with every projection on the fp32 matrix cores / in the 22-bit f16 split mode, same '
                 'process. per_rank_width_emulation: `bench.py --n-hidden H/N` on one GPU = the per-rank work '
                 'of the N = 2/4/8 points (no collective); the multi-GPU numbers are the driver\'s.'}
json.dump(lines, open(os.path.join(P, TAG + '_bench_lines.json'), 'w'), indent=1)

# 2. kernel stats of the same workload
ks = os.path.join(F, 'kstats', 'step_kernel_stats.csv')
out = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'profile_report.py'), ks,
                      os.path.join(F, 'kstats_run.log'),
                      '`python3 bench.py --no-cpu-baseline --no-second-leg` (150 steps + 10 warm-up, GEMM mode '
                      'bf16x3 = the default, 1x MI355X), round 2'], capture_output=True, text=True, check=True).stdout
open(os.path.join(P, TAG + '_bench_n1_kernel_stats.md'), 'w').write(out)
open(os.path.join(P, TAG + '_bench_n1_kernel_stats.csv'), 'w').write(open(ks).read())


# 3. PMC tables
def table(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        d[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return d


def traffic(fetch, write, kname):          # every instantiation whose name starts with kname
    f = [x for k in fetch if k.startswith(kname) for x in fetch[k]['FETCH_SIZE']]
    w = [x for k in write if k.startswith(kname) for x in write[k]['WRITE_SIZE']]
    return dict(launches=len(f), FETCH_SIZE_KB_raw=round(sum(f) / len(f), 1),
                WRITE_SIZE_KB=round(sum(w) / len(w), 1),
                hbm_bytes_corrected=int((2 * sum(f) / len(f) + sum(w) / len(w)) * 1024))


md = ['# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), `bench.py --gemm-mode M --steps 20 '
      '--warmup 3 --no-cpu-baseline --no-second-leg --no-kernel-timing`, round 2', '',
      'Raw counter averages per dispatch (KB). gfx950 correction: fabric read bytes = 2 x FETCH_SIZE for wide '
      'coalesced streams (MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact.', '']
tables = {}
for mode in ('bf16x3', 'f32'):
    fetch = table(os.path.join(F, 'pmc_FETCH_SIZE_' + mode, 't_counter_collection.csv'))
    write = table(os.path.join(F, 'pmc_WRITE_SIZE_' + mode, 't_counter_collection.csv'))
    tables[mode] = (fetch, write)
    rows = []
    for k in fetch:
        f = fetch[k]['FETCH_SIZE']
        w = write.get(k, {}).get('WRITE_SIZE', [0.0])
        rows.append((sum(f), k, len(f), sum(f) / len(f), sum(w) / max(len(w), 1)))
    rows.sort(reverse=True)
    md += ['## GEMM mode ' + mode + (' (the default)' if mode == 'bf16x3' else ''), '',
           '| kernel | dispatches | FETCH_SIZE avg KB (raw) | WRITE_SIZE avg KB |', '|---|---|---|---|']
    for tot, k, n, fa, wa in rows[:20]:
        md.append('| `%s` | %d | %.1f | %.1f |' % (k, n, fa, wa))
    md.append('')
open(os.path.join(P, TAG + '_pmc_fetch_write.md'), 'w').write('\n'.join(md) + '\n')

method = ('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --gemm-mode %s --steps 20 '
          '--warmup 3`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 128-B requests at '
          '64 B, MI355X_MICROARCH.md section HBM; WRITE_SIZE exact)')
fetch, write = tables['bf16x3']
sp = {k: traffic(fetch, write, k) for k in fetch if k.startswith('gist::spmm_csr')}
tot_l = sum(v['launches'] for v in sp.values())
json.dump({'hbm_bytes_per_launch': int(sum(v['hbm_bytes_corrected'] * v['launches'] for v in sp.values()) / tot_l),
           'method': method % 'bf16x3' + ', averaged over all SpMM launches of the step (4 x D=4096 on the block-dense '
                                         'matrix-core kernel, 1 x D=602 on the row-split kernel)',
           'kernels': sp, 'round': 2}, open(os.path.join(P, 'spmm_traffic.json'), 'w'), indent=1)
gb = {k: traffic(fetch, write, k) for k in fetch if k.startswith('gist::gemm_b3_kernel')}
tot_l = sum(v['launches'] for v in gb.values())
json.dump({'hbm_bytes_per_launch': int(sum(v['hbm_bytes_corrected'] * v['launches'] for v in gb.values()) / tot_l),
           'method': method % 'bf16x3' + ', averaged over the bf16x3 main-kernel launches of the step (5 per step)',
           'kernels': gb, 'round': 2}, open(os.path.join(P, 'gemm_b3_traffic.json'), 'w'), indent=1)
fetch, write = tables['f32']
gf = {k: traffic(fetch, write, k) for k in fetch if k.startswith('gist::gemm_f32_kernel')}
tot_l = sum(v['launches'] for v in gf.values())
json.dump({'hbm_bytes_per_launch': int(sum(v['hbm_bytes_corrected'] * v['launches'] for v in gf.values()) / tot_l),
           'method': method % 'f32' + ', averaged over the fp32 GEMM launches of the step (all layouts / tiles)',
           'kernels': gf, 'round': 2}, open(os.path.join(P, 'gemm_f32_traffic.json'), 'w'), indent=1)

# 4. MFMA occupancy / clock of the GEMM kernels in the three modes
md = ['# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES, `bench.py '
      '--gemm-mode M --steps 20 --warmup 3 ...`, round 2', '',
      'Per dispatch medians. cycles = GRBM_GUI_ACTIVE / 8 XCDs; MFMA pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / '
      '(1024 SIMDs x cycles); clock = cycles / duration (reads high on dispatches shorter than ~0.3 ms).', '',
      '| mode | kernel | workgroups | dispatches | duration us | cycles | MFMA occupancy | clock GHz | LDS bank conflicts |',
      '|---|---|---|---|---|---|---|---|---|']
for mode, sub in (('bf16x3', 'pmc_mfma_bf16x3'), ('f32', 'pmc_mfma_f32'), ('f16x3', 'pmc_mfma_f16x3')):
    mf = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(F, sub, 't_counter_collection.csv'))):
        if any(s in r['Kernel_Name'] for s in ('gemm_h3_kernel', 'gemm_b3_kernel', 'gemm_f32_kernel')):
            key = (short(r['Kernel_Name']), int(r['Grid_Size']) // int(r['Workgroup_Size']))
            mf[key][r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
                dur[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for key in sorted(mf, key=lambda k: -sum(dur[k]))[:6]:
        c = {k: sorted(v)[len(v) // 2] for k, v in mf[key].items()}
        d_us = sorted(dur[key])[len(dur[key]) // 2]
        cyc = c['GRBM_GUI_ACTIVE'] / 8
        md.append('| %s | `%s` | %d | %d | %.1f | %.0f | %.3f | %.2f | %.0f |' % (
            mode, key[0], key[1], len(dur[key]), d_us, cyc, c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc,
            cyc / d_us / 1e3, c.get('SQ_LDS_BANK_CONFLICT', 0)))
open(os.path.join(P, TAG + '_pmc_mfma.md'), 'w').write('\n'.join(md) + '\n')

# 5. micro-benchmarks
for src, dst in (('spmm_bench.log', TAG + '_spmm_kernels_bench.txt'),
                 ('spmm_kernel_times.txt', TAG + '_spmm_kernel_times.txt'),
                 ('spmm_mf_phases.log', TAG + '_spmm_mfma_phases.txt'),
                 ('b3_clock.log', TAG + '_gemm_b3_clock.txt'),
                 ('gemm_modes_bench.log', TAG + '_gemm_modes_bench.txt')):
    open(os.path.join(P, dst), 'w').write(clean(os.path.join(F, src)))

# 6. full-graph evaluation
ev = last_json(os.path.join(F, 'eval.log'))
ev['note'] = ('round 2: the evaluator works block of rows by block of rows (2 x [N, H] + one row block instead of '
              '[N, 2H] per layer); projections in the library default GEMM mode (bf16x3 for the large ones); the '
              'blocked SpMM kernels do not pay on the full graph (mean degree 490: every row is a long row)')
json.dump(ev, open(os.path.join(P, TAG + '_eval_fullgraph.json'), 'w'), indent=1)
print(open(os.path.join(P, TAG + '_pmc_mfma.md')).read())
print(json.dumps(json.load(open(os.path.join(P, 'spmm_traffic.json')))['hbm_bytes_per_launch']))
