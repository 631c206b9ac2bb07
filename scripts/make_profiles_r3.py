"""Turn gpurun_out/final_r3 (scripts/final_profile_r3.sh) into the round-3 files under profiles/."""
import collections, csv, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, 'gpurun_out', 'final_r3')
P = os.path.join(ROOT, 'profiles')
TAG = 'r03'


def last_json(path, key='{"metric"'):
    return json.loads([l for l in open(path) if l.startswith(key)][-1])


def short(name):
    return re.sub(r'^void ', '', name).split('(')[0]


def clean(path):
    return ''.join(l for l in open(path) if 'amdgpu.ids' not in l)


# 1. bench lines: the default workload, the per-rank widths, one line per BASELINE config
other = []
for i in (1, 2):
    f = os.path.join(F, 'bench_n1_box%d.log' % i)
    if os.path.exists(f):
        d = last_json(f)
        other.append({k: d[k] for k in ('value', 'ms_per_step')} | {'gemm_b3_avg_launch_ms': d['roofline']['avg_launch_ms'],
                                                                    'roofline_frac': d['roofline']['frac'],
                                                                    'f32_mfma_ms_per_step': d['f32_mfma']['ms_per_step'],
                                                                    'f16x3_split_ms_per_step': d['f16x3_split']['ms_per_step']})
lines = {'bench_n1': last_json(os.path.join(F, 'bench_n1.log')),
         'bench_n1_same_command_other_boxes': other,
         'per_rank_width_emulation': {('n_hidden_%d' % h): last_json(os.path.join(F, 'bench_h%d.log' % h))
                                      for h in (2048, 1024, 512)},
         'baseline_configs': {'config_2': last_json(os.path.join(F, 'bench_cfg2.log')),
                              'config_4_one_rank_of_8': last_json(os.path.join(F, 'bench_cfg4.log')),
                              'config_5_one_rank_of_8_plus_exchange': last_json(os.path.join(F, 'bench_cfg5.log'))},
         'note': 'bench.py on 1x MI355X, round 3, final code.  Every gpurun call lands on another box and the bf16-MFMA-bound kernels follow the '
                 'box (power / clock): the default invocation measured 2.63-2.73 ms/step over six boxes this round; bench_n1 is the run the '
                 'kernel statistics and PMC passes of profiles/r03_* were taken with (the slowest of them), bench_n1_same_command_other_boxes two '
                 'more.  bench_n1 = the default invocation (BASELINE config 3 at N = 1: '
                 'Reddit-like, H = 4096, L = 2, GEMM mode bf16x3); `f32_mfma` / `f16x3_split` = the same workload '
                 're-timed in the other GEMM modes, same process.  per_rank_width_emulation: `bench.py --n-hidden H/N '
                 '--steps 300 --warmup 20` = the per-rank work of the N = 2/4/8 points of config 3 on one GPU (no '
                 'collective; HIP-event kernel timing on every 8th step).  baseline_configs: `bench.py --config 2 '
                 '--steps 300`, `--config 4 --steps 600` (ONE rank of the 8-GPU run: its sub-GCN of width 512 on the '
                 'Amazon-like graph), `--config 5` (one rank of H = 32768 / 8 plus the H = 32768, S = 8 weight '
                 'exchange measured with 8 base replicas on this GPU).  Multi-GPU numbers are the driver\'s.'}
json.dump(lines, open(os.path.join(P, TAG + '_bench_lines.json'), 'w'), indent=1)

# 2. kernel stats: default workload + the fused small-width steps + configs 2 and 4
title = ('`python3 bench.py --no-cpu-baseline --no-second-leg` (150 steps + 10 warm-up, GEMM mode bf16x3 = the '
         'default, 1x MI355X), round 3')
ks = os.path.join(F, 'kstats', 'step_kernel_stats.csv')
out = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'profile_report.py'), ks,
                      os.path.join(F, 'kstats_run.log'), title], capture_output=True, text=True, check=True).stdout
open(os.path.join(P, TAG + '_bench_n1_kernel_stats.md'), 'w').write(out)
open(os.path.join(P, TAG + '_bench_n1_kernel_stats.csv'), 'w').write(open(ks).read())
for sub, what in (('h1024', '--n-hidden 1024'), ('h512', '--n-hidden 512'), ('cfg2', '--config 2'),
                  ('cfg4', '--config 4')):
    ks = os.path.join(F, 'kstats_' + sub, 'step_kernel_stats.csv')
    d = last_json(os.path.join(F, 'kstats_%s.log' % sub))
    rows = list(csv.DictReader(open(ks)))
    steps = 220.0
    step_rows = [r for r in rows if int(r['Calls']) >= 200]          # kernels of the training step
    tot = sum(float(r['TotalDurationNs']) for r in step_rows)
    txt = ['# rocprofv3 --kernel-trace --stats: `python3 bench.py %s --steps 200 --warmup 20 --no-second-leg '
           '--no-cpu-baseline --no-kernel-timing`, 1x MI355X, round 3' % what, '',
           'bench line of the profiled run: ms_per_step %.4f, value %.4f epochs/s' % (d['ms_per_step'], d['value']),
           '', 'Kernels of the training step (called at least once per step): %.1f launches and %.1f us of kernel '
           'time per step.' % (sum(int(r['Calls']) for r in step_rows) / steps, tot / steps / 1e3), '',
           '| kernel | launches / step | avg us | us / step | % of step kernels |', '|---|---|---|---|---|']
    for r in step_rows:
        txt.append('| `%s` | %.2f | %.2f | %.2f | %.1f |' % (
            short(r['Name'])[:90], int(r['Calls']) / steps, float(r['AverageNs']) / 1e3,
            float(r['TotalDurationNs']) / steps / 1e3, 100.0 * float(r['TotalDurationNs']) / tot))
    open(os.path.join(P, '%s_step_kernels_%s.md' % (TAG, sub)), 'w').write('\n'.join(txt) + '\n')


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
      '--warmup 3 --no-cpu-baseline --no-second-leg --no-kernel-timing`, round 3', '',
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
           'kernels': sp, 'round': 3}, open(os.path.join(P, 'spmm_traffic.json'), 'w'), indent=1)
gb = {k: traffic(fetch, write, k) for k in fetch if k.startswith('gist::gemm_b3_kernel')}
tot_l = sum(v['launches'] for v in gb.values())
json.dump({'hbm_bytes_per_launch': int(sum(v['hbm_bytes_corrected'] * v['launches'] for v in gb.values()) / tot_l),
           'method': method % 'bf16x3' + ', averaged over the bf16x3 main-kernel launches of the step (5 per step)',
           'kernels': gb, 'round': 3}, open(os.path.join(P, 'gemm_b3_traffic.json'), 'w'), indent=1)
fetch, write = tables['f32']
gf = {k: traffic(fetch, write, k) for k in fetch if k.startswith('gist::gemm_f32_kernel')}
tot_l = sum(v['launches'] for v in gf.values())
json.dump({'hbm_bytes_per_launch': int(sum(v['hbm_bytes_corrected'] * v['launches'] for v in gf.values()) / tot_l),
           'method': method % 'f32' + ', averaged over the fp32 GEMM launches of the step (all layouts / tiles)',
           'kernels': gf, 'round': 3}, open(os.path.join(P, 'gemm_f32_traffic.json'), 'w'), indent=1)

# 4. MFMA occupancy / clock of the GEMM kernels
md = ['# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES, `bench.py '
      '--gemm-mode M --steps 20 --warmup 3 ...`, round 3', '',
      'Per dispatch medians. cycles = GRBM_GUI_ACTIVE / 8 XCDs; MFMA pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / '
      '(1024 SIMDs x cycles); clock = cycles / duration (reads high on dispatches shorter than ~0.3 ms).', '',
      '| mode | kernel | workgroups | dispatches | duration us | cycles | MFMA occupancy | clock GHz | LDS bank conflicts |',
      '|---|---|---|---|---|---|---|---|---|']
for mode, sub in (('bf16x3', 'pmc_mfma_bf16x3'), ('f32', 'pmc_mfma_f32')):
    mf = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(F, sub, 't_counter_collection.csv'))):
        if any(s in r['Kernel_Name'] for s in ('gemm_b3_kernel', 'gemm_b3c_kernel', 'gemm_f32_kernel')):
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
hdr = ('# scripts/b3c_bench.py 0 0 64 1 64 2 128 1 128128 1 128128 2 on MI355X: the convert-on-load bf16x3 GEMM (gemm_b3c.hip) vs the\n'
       '# fp32 kernel, standalone calls (us; a call that splits k includes its reduce pass).  "default/default" = the library\'s dispatch\n'
       '# (NT / NN from 2 GFLOP with >= 256 tiles of 64 x 64 on gemm_b3c_kernel -- 128 x 128 tiles where the output has 256 of them --, TN on the\n'
       '# fp32 kernel, >= 9-16 GFLOP on the pre-split kernel gemm_b3 incl. its per-call pre-pass); tile/splits columns force gemm_b3c_kernel (tuning\n'
       '# hook b3c = 2) with that tile (64 = 64 x 64, 128 = 128 x 64, 128128 = 128 x 128) and k-slice count; the last three rows are above the\n'
       '# pre-split threshold: every column is gemm_b3.\n')
open(os.path.join(P, TAG + '_b3c_bench.txt'), 'w').write(hdr + clean(os.path.join(F, 'b3c_bench.log')))

# 6. full-graph evaluation: timings + PMC traffic of the aggregation
ev = last_json(os.path.join(F, 'eval.log'), '{')
info = last_json(os.path.join(F, 'pmc_eval_FETCH_SIZE.log'), '{')
fetch = table(os.path.join(F, 'pmc_eval_FETCH_SIZE', 't_counter_collection.csv'))
write = table(os.path.join(F, 'pmc_eval_WRITE_SIZE', 't_counter_collection.csv'))
pm = {}
for k in fetch:
    if not k.startswith('gist::spmm_csr'):
        continue
    f, w = fetch[k]['FETCH_SIZE'], write.get(k, {}).get('WRITE_SIZE', [0.0])
    pm[k] = dict(dispatches=len(f), FETCH_SIZE_KB_raw_total=round(sum(f), 1), WRITE_SIZE_KB_total=round(sum(w), 1),
                 fabric_bytes_total=int((2 * sum(f) + sum(w)) * 1024))
ev['aggregation_traffic_D4096'] = {
    'method': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over scripts/eval_spmm_pmc.py: per form a '
              'warm-up call and a measured call, so every total below covers TWO calls of (a) one gather pass over A '
              '[spmm_csr_rowsplit_kernel, also used by (c)], (b) A_diag on the matrix cores [spmm_csr_mfma_kernel], (c) '
              'A_rest in 8 column tiles of 512 floats; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction)',
    'counters': pm, 'timings_and_algorithmic_bytes': info}
ev['note'] = ('round 3: with node_blocks the aggregation is A = A_diag (inside the parts: counts x features on the matrix '
              'cores, 1.9 ms at D = 4096 for 55 % of the edges) + A_rest (random inter-part edges: gathered in 512-float '
              'column tiles, 7.4 TB/s of gather bytes -- without the intra-part neighbours there is no L2 reuse left, so '
              'the remainder is bound by the fabric / Infinity Cache, not by HBM)')
json.dump(ev, open(os.path.join(P, TAG + '_eval_fullgraph.json'), 'w'), indent=1)
print(open(os.path.join(P, TAG + '_pmc_mfma.md')).read())
print(json.dumps(json.load(open(os.path.join(P, 'spmm_traffic.json')))['hbm_bytes_per_launch']))
print(json.dumps(pm, indent=1))
