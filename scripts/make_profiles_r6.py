"""Turn gpurun_out/final_r6 (scripts/final_profile_r6.sh) into the round-6 files under profiles/."""
import collections, csv, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, 'gpurun_out', 'final_r6')
P = os.path.join(ROOT, 'profiles')
TAG = 'r06'


def last_json(path, key='{"metric"'):
    return json.loads([l for l in open(path) if l.startswith(key)][-1])


def short(name):
    return re.sub(r'^void ', '', name.replace('(anonymous namespace)::', '')).split('(')[0]


def clean(path):
    return ''.join(l for l in open(path) if 'amdgpu.ids' not in l)


# 1. bench lines: the default workload, the per-rank widths, one line per BASELINE config
lines = {'bench_n1': last_json(os.path.join(F, 'bench_n1.log')),
         'per_rank_width_emulation': {('n_hidden_%d' % h): last_json(os.path.join(F, 'bench_h%d.log' % h))
                                      for h in (2048, 1024, 512)},
         'baseline_configs': {'config_2': last_json(os.path.join(F, 'bench_cfg2.log')),
                              'config_4_one_rank_of_8': last_json(os.path.join(F, 'bench_cfg4.log')),
                              'config_5_one_rank_of_8_plus_exchange': last_json(os.path.join(F, 'bench_cfg5.log'))},
         'note': 'bench.py on 1x MI355X, round 6, final code (scripts/final_profile_r6.sh).  bench_n1 = the default invocation '
                 '(BASELINE config 3 at N = 1: Reddit-like, H = 4096, L = 2, GEMM mode bf16x3) incl. the CPU baseline over one full '
                 'epoch; `f32_mfma` / `f16x3_split` = the same workload re-timed in the other GEMM modes, same process.  '
                 'per_rank_width_emulation: `bench.py --n-hidden H/N --steps 300 --warmup 20` = the per-rank work of the N = 2/4/8 '
                 'points of config 3 on one GPU (no collective).  baseline_configs: `bench.py --config 2 --steps 300`, `--config 4 '
                 '--steps 600` (ONE rank of the 8-GPU run), `--config 5` (one rank of H = 32768 / 8 plus the H = 32768, S = 8 weight '
                 'exchange measured with 8 base replicas on this GPU).  Multi-GPU numbers are the driver\'s.'}
json.dump(lines, open(os.path.join(P, TAG + '_bench_lines.json'), 'w'), indent=1)

# 2. kernel stats: default workload + the fused small-width steps + configs 2 and 4
title = ('`python3 bench.py --no-cpu-baseline --no-second-leg` (150 steps + 10 warm-up, GEMM mode bf16x3 = the '
         'default, 1x MI355X), round 6')
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
           '--no-cpu-baseline --no-kernel-timing`, 1x MI355X, round 6' % what, '',
           'bench line of the profiled run: ms_per_step %.4f, value %.4f epochs/s' % (d['ms_per_step'], d['value']),
           '', 'Kernels of the training step (called at least once per step): %.1f launches and %.1f us of kernel '
           'time per step.' % (sum(int(r['Calls']) for r in step_rows) / steps, tot / steps / 1e3), '',
           '| kernel | launches / step | avg us | us / step | % of step kernels |', '|---|---|---|---|---|']
    for r in step_rows:
        txt.append('| `%s` | %.2f | %.2f | %.2f | %.1f |' % (
            short(r['Name'])[:90], int(r['Calls']) / steps, float(r['AverageNs']) / 1e3,
            float(r['TotalDurationNs']) / steps / 1e3, 100.0 * float(r['TotalDurationNs']) / tot))
    seq = os.path.join(F, 'seq_%s.txt' % sub)
    if os.path.exists(seq):
        txt += ['', 'The step launch by launch (scripts/step_seq.py on the same trace: median duration of every launch in stream '
                'order; blank names = kernels in an anonymous namespace: the fused class layer `class_layer_kernel` [grid n/16 x 256] '
                'and its `class_dw_kernel` [grid K/64 x n/128]; `gemm_f32_dual_kernel` = dZ and dW of a hidden layer in one launch):', '', '```'] + open(seq).read().rstrip().split('\n') + ['```']
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
      '--warmup 3 --no-cpu-baseline --no-second-leg --no-kernel-timing`, round 6', '',
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
           'kernels': sp, 'round': 6}, open(os.path.join(P, 'spmm_traffic.json'), 'w'), indent=1)
gb = {k: traffic(fetch, write, k) for k in fetch if k.startswith('gist::gemm_b3_kernel')}
tot_l = sum(v['launches'] for v in gb.values())
json.dump({'hbm_bytes_per_launch': int(sum(v['hbm_bytes_corrected'] * v['launches'] for v in gb.values()) / tot_l),
           'method': method % 'bf16x3' + ', averaged over the bf16x3 main-kernel launches of the step (5 per step)',
           'kernels': gb, 'round': 6}, open(os.path.join(P, 'gemm_b3_traffic.json'), 'w'), indent=1)
fetch, write = tables['f32']
gf = {k: traffic(fetch, write, k) for k in fetch if k.startswith('gist::gemm_f32_kernel')}
tot_l = sum(v['launches'] for v in gf.values())
json.dump({'hbm_bytes_per_launch': int(sum(v['hbm_bytes_corrected'] * v['launches'] for v in gf.values()) / tot_l),
           'method': method % 'f32' + ', averaged over the fp32 GEMM launches of the step (all layouts / tiles)',
           'kernels': gf, 'round': 6}, open(os.path.join(P, 'gemm_f32_traffic.json'), 'w'), indent=1)

# 4. MFMA occupancy / clock of the GEMM kernels
md = ['# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES, `bench.py '
      '--gemm-mode M --steps 20 --warmup 3 ...`, round 6', '',
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

# 5. SQ counters of the narrow steps: what the small kernels wait for
md = ['# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (one pass) and '
      '--pmc FETCH_SIZE and --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE (passes of their own), `bench.py <config> --steps 20 --warmup 3 --no-cpu-baseline --no-second-leg --no-kernel-timing`, round 6', '',
      'Per kernel of the step, medians over its dispatches.  wait share = SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave-cycles parked at s_waitcnt / '
      's_barrier: memory latency and barriers), issue-stall share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, active share = SQ_ACTIVE_INST_ANY / '
      'SQ_WAVE_CYCLES; MFMA occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); fabric read = 2 x FETCH_SIZE (gfx950 '
      'correction).  Beside these rows the bench lines of the same widths carry `launch_floor`: the step\'s launch count and the cost of as many '
      'EMPTY launches on the same box (1.5 us each: 8 % of the h = 512 step, 12 % of config 2\'s) -- the steps are not bound by launch overhead but by '
      'kernels of 5-50 us whose waves spend most of their cycles parked at s_waitcnt / s_barrier on chains of dependent memory round trips, the projections '
      'issue-stalled on a matrix pipe that one 32 x 32 tile per SIMD keeps a third to a half busy.', '']
for sub, what in (('h512', '--n-hidden 512'), ('cfg2', '--config 2'), ('h4096', '(the default workload: H = 4096)')):
    f = os.path.join(F, 'pmc_sq_' + sub, 't_counter_collection.csv')
    if not os.path.exists(f):
        continue
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        key = (short(r['Kernel_Name'])[:70], int(r['Grid_Size']) // int(r['Workgroup_Size']))
        d[key][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
            dur[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    fe = collections.defaultdict(list)
    ff = os.path.join(F, 'pmc_fetch_' + sub, 't_counter_collection.csv')
    if os.path.exists(ff):
        for r in csv.DictReader(open(ff)):
            fe[(short(r['Kernel_Name'])[:70], int(r['Grid_Size']) // int(r['Workgroup_Size']))].append(float(r['Counter_Value']))
    ld = collections.defaultdict(lambda: collections.defaultdict(list))
    fl = os.path.join(F, 'pmc_lds_' + sub, 't_counter_collection.csv')
    if os.path.exists(fl):
        for r in csv.DictReader(open(fl)):
            ld[(short(r['Kernel_Name'])[:70], int(r['Grid_Size']) // int(r['Workgroup_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
    md += ['## `bench.py %s`' % what, '', '| kernel | workgroups | dispatches | duration us (profiled) | wait share | issue-stall share | active share | MFMA occupancy | fabric read MB | LDS bank-conflict cycles / LDS active cycles |',
           '|---|---|---|---|---|---|---|---|---|---|']
    med = lambda v: sorted(v)[len(v) // 2] if v else 0.0
    for key in sorted(d, key=lambda k: -sum(dur[k])):
        if len(dur[key]) < 15:
            continue
        c = {k: med(v) for k, v in d[key].items()}
        wc = max(c.get('SQ_WAVE_CYCLES', 0.0), 1.0)
        cyc = max(c.get('GRBM_GUI_ACTIVE', 0.0) / 8, 1.0)
        la = med(ld[key].get('SQ_LDS_IDX_ACTIVE', []))
        lc = med(ld[key].get('SQ_LDS_BANK_CONFLICT', []))
        md.append('| `%s` | %d | %d | %.1f | %.2f | %.2f | %.2f | %.3f | %.2f | %s |' % (
            key[0] or '(anonymous namespace)', key[1], len(dur[key]), med(dur[key]), c.get('SQ_WAIT_ANY', 0) / wc,
            c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('SQ_ACTIVE_INST_ANY', 0) / wc,
            c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / cyc, 2 * med(fe.get(key, [0.0])) / 1024,
            ('%.3f' % (lc / la)) if la > 0 else '-'))
    md.append('')
open(os.path.join(P, TAG + '_pmc_narrow_steps.md'), 'w').write('\n'.join(md) + '\n')
print(open(os.path.join(P, TAG + '_pmc_mfma.md')).read())

# 6. round 6: the module path beside the engine path, the unplanted graph, the partitioner on the GPU box's host
mp = {}
for tag, path in (('default_H4096', 'bench_n1.log'), ('n_hidden_2048', 'bench_h2048.log'), ('n_hidden_1024', 'bench_h1024.log'),
                  ('n_hidden_512', 'bench_h512.log'), ('config_2', 'bench_cfg2.log'), ('config_4_one_rank_of_8', 'bench_cfg4.log')):
    d = last_json(os.path.join(F, path))
    m = d.get('module_path') or {}
    mp[tag] = {'engine_path_ms_per_step': d['ms_per_step'], 'module_path_ms_per_step': m.get('ms_per_step'),
               'module_over_engine': m.get('vs_engine_path_ms_per_step'),
               'host_issue_ms_per_step_engine': (d.get('host_issue_ms_per_step') or {}).get('median'),
               'host_issue_ms_per_step_module': (m.get('host_issue_ms_per_step') or {}).get('median'),
               'cfs_throttled_periods_engine': (d.get('host_counters_delta') or {}).get('cgroup_nr_throttled'),
               'cfs_throttled_periods_module': (m.get('host_counters_delta') or {}).get('cgroup_nr_throttled')}
json.dump({'note': 'bench.py lines of scripts/final_profile_r6.sh: the headline (engine path: one gist_sage_step per iteration) and the '
                   'module_path leg of the same process (the reference loop body on gist_amd.modules.GCN / nn.CrossEntropyLoss / '
                   'optim.Adam / sampler.ClusterIter: three phase calls per iteration); host_issue = median host time between the '
                   'starts of two iterations (no synchronisation in the loop)', 'lines': mp},
          open(os.path.join(P, TAG + '_module_path.json'), 'w'), indent=1)
U = os.path.join(ROOT, 'gpurun_out', 'r6', 'unplanted')
if os.path.isdir(U):
    un = {}
    for ds in ('reddit-synth', 'reddit-communities'):
        for h in (4096, 512):
            f = os.path.join(U, '%s_h%d.json' % (ds, h))
            if not os.path.exists(f):
                continue
            d = json.load(open(f))
            un['%s_h%d' % (ds, h)] = {
                'workload': d['config']['workload'], 'ms_per_step': d['ms_per_step'], 'epochs_per_s': d['value'],
                'module_path_ms_per_step': (d.get('module_path') or {}).get('ms_per_step'),
                'roofline_spmm': {k: d['roofline_spmm'][k] for k in ('kernel', 'achieved', 'frac', 'avg_launch_ms', 'launches',
                                                                      'mean_batch_rows', 'mean_batch_nnz')},
                'gemm_avg_launch_ms': d['roofline']['avg_launch_ms'], 'batch_locality': d.get('batch_locality'),
                'partition': d.get('partition')}
    json.dump({'note': 'scripts/r6_unplanted.sh on one box: the planted block model (parts = its blocks) beside the power-law community '
                       'graph (communities of 30-400 nodes, mixing 0.3, random node ids) cut into 1500 parts by gist_partition_graph, at '
                       'the metric\'s width and at config 3\'s 8-GPU per-rank width.  Round 6: the projections\' tiles past the last full round '
                       'run as k slices (gemm_b3.hip tail units: every second batch of this graph has a ninth 256-row tile), two parts of one '
                       'community in a batch are multiplied as a dense pair (spmm_mfma.hip).  Round 5 on this graph: 2.89 ms / 0.287 ms against '
                       '2.52 / 0.264 planted', 'runs': un},
              open(os.path.join(P, TAG + '_unplanted_graph.json'), 'w'), indent=1)
pb = os.path.join(ROOT, 'gpurun_out', 'r6', 'partitioner_box.json')
if os.path.exists(pb):
    open(os.path.join(P, TAG + '_partitioner.json'), 'w').write(open(pb).read())
