"""Dev tool: sweep (tile, splits) per GEMM shape in SUBPROCESSES (the override is read once
per process) and print the best configuration -- input for gemm.hip:choose_cfg."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [('nt', 2046, 4096, 8192), ('nt', 2046, 4096, 1204), ('tn', 4096, 1204, 2046),
          ('tn', 4096, 8192, 2046), ('nn', 2046, 8192, 4096),
          ('nt', 2046, 41, 8192), ('tn', 41, 8192, 2046), ('nn', 2046, 8192, 41),
          ('nt', 2046, 2048, 1204), ('nt', 2046, 2048, 4096), ('tn', 2048, 4096, 2046), ('nn', 2046, 4096, 2048),
          ('tn', 2048, 1204, 2046), ('nt', 2046, 41, 4096), ('tn', 41, 4096, 2046),
          ('nt', 2046, 1024, 1204), ('nt', 2046, 1024, 2048), ('tn', 1024, 2048, 2046), ('nn', 2046, 2048, 1024),
          ('tn', 1024, 1204, 2046), ('nt', 2046, 41, 2048), ('tn', 41, 2048, 2046),
          ('nt', 2046, 512, 1204), ('nt', 2046, 512, 1024), ('tn', 512, 1024, 2046), ('nn', 2046, 1024, 512),
          ('tn', 512, 1204, 2046), ('nt', 2046, 41, 1024), ('tn', 41, 1024, 2046), ('nn', 2046, 1024, 41),
          ('nt', 2046, 256, 1204), ('nt', 2046, 256, 512), ('tn', 256, 512, 2046), ('nn', 2046, 512, 256)]
if os.environ.get('GIST_SWEEP_SHAPES'):      # e.g. 'nt,1140,512,1024;nn,1140,1024,512'
    SHAPES = [(t.split(',')[0],) + tuple(int(v) for v in t.split(',')[1:]) for t in os.environ['GIST_SWEEP_SHAPES'].split(';')]
CHILD = r'''
import sys, json, os, torch
sys.path.insert(0, %r)
from gist_amd import hip
dev = torch.device('cuda', 0)
if os.environ.get('GIST_B3C'):
    hip.tuning('b3c', int(os.environ['GIST_B3C']))
if os.environ.get('GIST_GEMM_TILE') and os.environ.get('GIST_GEMM_SPLITS'):
    hip.tuning('gemm_tile', int(os.environ['GIST_GEMM_TILE']))
    hip.tuning('gemm_splits', int(os.environ['GIST_GEMM_SPLITS']))
shapes = json.loads(sys.argv[1])
out = []
for (lay, m, n, k) in shapes:
    if lay == 'nt':
        a, w, y = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), torch.empty(m, n, device=dev)
        f = lambda: hip.gemm_nt(a, w, None, y)
    elif lay == 'nn':
        a, w, y = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev), torch.empty(m, n, device=dev)
        f = lambda: hip.gemm_nn(a, w, y)
    else:
        a, w, y = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev), torch.empty(m, n, device=dev)
        f = lambda: hip.gemm_tn(a, w, y)
    hip.workspace(32 * m * n * 4, dev)
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); out.append(ts[len(ts)//2])
print(json.dumps(out))
''' % ROOT
res = {}
import itertools
CONFIGS = [(128, 1), (128, 2), (64, 1), (64, 2), (64, 4), (64, 8), (64, 16)] if '--quick' in sys.argv else list(itertools.product((128, 64), (1, 2, 4, 8, 16, 32)))
for tile, sp in CONFIGS:
    if True:
        env = dict(os.environ, GIST_GEMM_TILE=str(tile), GIST_GEMM_SPLITS=str(sp))
        o = subprocess.run([sys.executable, '-c', CHILD, json.dumps(SHAPES)], env=env, capture_output=True, text=True)
        line = [l for l in o.stdout.strip().split('\n') if l.startswith('[')]
        if not line:
            print('FAILED', tile, sp, o.stderr[-400:]); continue
        res[(tile, sp)] = json.loads(line[-1])
env = {k: v for k, v in os.environ.items() if not k.startswith('GIST_GEMM')}
o = subprocess.run([sys.executable, '-c', CHILD, json.dumps(SHAPES)], env=env, capture_output=True, text=True)
auto = json.loads([l for l in o.stdout.strip().split('\n') if l.startswith('[')][-1])
for i, sh in enumerate(SHAPES):
    items = sorted(((v[i], k) for k, v in res.items()))
    best_t, best_k = items[0]
    fl = 2.0 * sh[1] * sh[2] * sh[3]
    print('%s m=%5d n=%5d k=%5d  best %s %.3f ms %.1f TF | auto %.3f ms %.1f TF | t128s1 %.3f  t64s1 %.3f | top3 %s' % (
        sh[0], sh[1], sh[2], sh[3], best_k, best_t, fl / best_t / 1e9, auto[i], fl / auto[i] / 1e9,
        res[(128, 1)][i], res[(64, 1)][i], [(k, round(t, 3)) for t, k in items[:3]]), flush=True)
