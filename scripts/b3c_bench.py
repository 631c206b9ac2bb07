"""Dev tool: the convert-on-load bf16x3 GEMM (gemm_b3c.hip) against the fp32 kernel on the per-rank shapes of
the N = 4 / 8 points, with tile / split-K overrides.  python scripts/b3c_bench.py [tile splits]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gist_amd import hip

dev = 'cuda:0'
SH = [('nt', 2046, 1024, 1204), ('nt', 2046, 1024, 2048), ('nn', 2046, 2048, 1024), ('tn', 1024, 2048, 2046),
      ('tn', 1024, 1204, 2046), ('nt', 2046, 512, 1204), ('nt', 2046, 512, 1024), ('nn', 2046, 1024, 512),
      ('tn', 512, 1024, 2046), ('tn', 512, 1204, 2046), ('nt', 2046, 2048, 4096), ('nn', 2046, 4096, 2048),
      ('nt', 2046, 4096, 8192)]


def run(form, a, b, c):
    if form == 'nt':
        hip.gemm_nt(a, b, None, c)
    elif form == 'nn':
        hip.gemm_nn(a, b, c)
    else:
        hip.gemm_tn(a, b, c)


def t(form, a, b, c, it=30):
    for _ in range(3):
        run(form, a, b, c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        run(form, a, b, c)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


cfgs = [(0, 0)]
if len(sys.argv) > 2:
    cfgs = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
for form, m, n, k in SH:
    sa = (m, k) if form != 'tn' else (k, m)
    sb = (n, k) if form == 'nt' else (k, n)
    a, b = torch.randn(*sa, device=dev), torch.randn(*sb, device=dev)
    c = torch.empty(m, n, device=dev)
    hip.gemm_mode('f32')
    f32 = t(form, a, b, c)
    hip.gemm_mode('bf16x3')
    out = []
    for tile, sp in cfgs:
        # tile/splits given: every shape forced onto gemm_b3c_kernel with them (tuning hook b3c = 2);
        # 0 0 = the library's default dispatch (most layouts stay on the fp32 kernel)
        hip.tuning('b3c', 2 if (tile or sp) else 0)
        hip.tuning('gemm_tile', tile)
        hip.tuning('gemm_splits', sp)
        out.append('%s/%s: %6.1f' % (tile or 'default', sp or 'default', t(form, a, b, c)))
    hip.tuning('gemm_tile', 0)
    hip.tuning('gemm_splits', 0)
    hip.tuning('b3c', 0)
    gf = 2.0 * m * n * k / 1e9
    print('%s %5d %5d %5d  %5.2f GF  f32 %6.1f us (%5.1f TF) | b3c %s' % (form, m, n, k, gf, f32, gf / f32 * 1e3,
                                                                     '  '.join(out)), flush=True)
