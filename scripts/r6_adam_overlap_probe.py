"""Can the optimiser's HBM pass run BESIDE a bf16x3 projection (another stream, the CUs' free wave slots)?
Times, with HIP events on the main stream: the step's last projections alone, Adam over W_1's 33.5 M parameters alone,
and both issued together (Adam on a side stream that waits for an event of the main one and is joined afterwards)."""
import torch
from gist_amd import hip

DEV = 'cuda:0'
hip.gemm_mode('bf16x3')
n = 2046
P = 4096 * 8192
p, g, m, v = (torch.randn(P, device=DEV) for _ in range(4))
v.abs_()
shapes = [('tn', 4096, 1204, n), ('tn', 4096, 8192, n), ('nn', n, 8192, 4096)]
side = torch.cuda.Stream()


def timed(f, reps=10):
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


def adam():
    hip.adam_(p, g, m, v, 3, 0.01)


t_adam = timed(adam)
print('adam alone (33.5 M parameters): %.1f us' % t_adam)
for form, mm, nn, kk in shapes:
    sa, sb = {'nn': ((mm, kk), (kk, nn)), 'tn': ((kk, mm), (kk, nn))}[form]
    a = torch.randn(*sa, device=DEV)
    b = torch.randn(*sb, device=DEV)
    y = torch.empty(mm, nn, device=DEV)
    f = {'nn': lambda: hip.gemm_nn(a, b, y), 'tn': lambda: hip.gemm_tn(a, b, y)}[form]
    t_g = timed(f)

    def both():
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            adam()
        f()
        ev2 = torch.cuda.Event()
        ev2.record(side)
        torch.cuda.current_stream().wait_event(ev2)
    t_b = timed(both)
    print('%s %d x %d x %d: projection alone %.1f us, with Adam beside it %.1f us (serial %.1f): hidden %.1f us'
          % (form, mm, nn, kk, t_g, t_b, t_g + t_adam, t_g + t_adam - t_b), flush=True)
