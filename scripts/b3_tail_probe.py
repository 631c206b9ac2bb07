"""Times the bf16x3 GEMM on the H = 4096 step's shapes at 2046 rows (256 tiles) and at 2049-2304 rows (a ninth
row tile), with and without the tail units (tuning hook b3_tail)."""
import sys
import torch
from gist_amd import hip

DEV = 'cuda:0'


def bench(form, m, n, k, reps=20):
    sa, sb = {'nt': ((m, k), (n, k)), 'nn': ((m, k), (k, n)), 'tn': ((k, m), (k, n))}[form]
    a = torch.randn(*sa, device=DEV)
    b = torch.randn(*sb, device=DEV)
    y = torch.empty(m, n, device=DEV)
    f = {'nt': lambda: hip.gemm_nt(a, b, None, y), 'nn': lambda: hip.gemm_nn(a, b, y), 'tn': lambda: hip.gemm_tn(a, b, y)}[form]
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


hip.gemm_mode('bf16x3')
rows = [int(x) for x in sys.argv[1:]] or [2046, 2049, 2100, 2200, 2304]
for form, n, k in [('nt', 4096, 8192), ('nn', 8192, 4096), ('nt', 4096, 1204), ('nt', 2048, 4096), ('nn', 4096, 2048)]:
    for m in rows:
        hip.tuning('b3_tail', 1)
        t_off = bench(form, m, n, k)
        hip.tuning('b3_tail', 0)
        t_on = bench(form, m, n, k)
        print(f'{form} m={m} n={n} k={k}: whole tiles {t_off:8.1f} us   tail units {t_on:8.1f} us  (incl. the per-call operand split)', flush=True)
