"""Dev tool: per-workgroup phase timeline of the fp32 GEMM (prologue / main loop / epilogue).

Builds nothing itself: expects scripts/_build/libgemm_trace.so =
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DGIST_GEMM_TRACE=1 \
        -Iinclude -shared -o scripts/_build/libgemm_trace.so gist_amd/csrc/gemm.hip gist_amd/csrc/capi.hip
Timestamps are s_memrealtime ticks (100 MHz)."""
import ctypes, os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(HERE, '_build', 'libgemm_trace.so'))
p, i64 = ctypes.c_void_p, ctypes.c_int64
L.gist_gemm_nt_f32.argtypes = [p, i64, p, i64, p, p, i64, i64, i64, i64, p, i64, p]
L.gist_gemm_nn_f32.argtypes = [p, i64, p, i64, p, i64, i64, i64, i64, p, i64, p]
L.gist_gemm_tn_f32.argtypes = [p, i64, p, i64, p, i64, i64, i64, i64, p, i64, p]
L.gist_gemm_trace_read.argtypes = [p, i64]
dev = torch.device('cuda', 0)


def run(kind, m, n, k):
    st = torch.cuda.current_stream().cuda_stream
    if kind == 'nt':
        a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); c = torch.empty(m, n, device=dev)
        f = lambda: L.gist_gemm_nt_f32(a.data_ptr(), k, w.data_ptr(), k, None, c.data_ptr(), n, m, n, k, None, 0, st)
    elif kind == 'nn':
        a = torch.randn(m, k, device=dev); w = torch.randn(k, n, device=dev); c = torch.empty(m, n, device=dev)
        f = lambda: L.gist_gemm_nn_f32(a.data_ptr(), k, w.data_ptr(), n, c.data_ptr(), n, m, n, k, None, 0, st)
    else:
        a = torch.randn(k, m, device=dev); w = torch.randn(k, n, device=dev); c = torch.empty(m, n, device=dev)
        f = lambda: L.gist_gemm_tn_f32(a.data_ptr(), m, w.data_ptr(), n, c.data_ptr(), n, m, n, k, None, 0, st)
    for _ in range(3):
        assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); assert f() == 0; e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    nb = ((m + 127) // 128) * ((n + 127) // 128)
    buf = np.zeros(nb * 12, np.uint64)
    assert L.gist_gemm_trace_read(buf.ctypes.data, nb) == 0
    t = buf.reshape(nb, 12)
    t0 = t[:, 0].min()
    us = (t[:, :4].astype(np.int64) - np.int64(t0)) / 100.0
    hw = t[:, 4]
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
    cu = ((hw & np.uint64(0xffffffff)).astype(np.int64) >> 8) & 0xf
    se = ((hw & np.uint64(0xffffffff)).astype(np.int64) >> 13) & 0x7
    cuid = xcc * 64 + se * 16 + cu
    print('%s %dx%dx%d: %.1f us by events, %.1f TF; blocks %d, distinct CUs %d' %
          (kind, m, n, k, ms * 1e3, 2.0 * m * n * k / ms / 1e9, nb, len(set(cuid.tolist()))))
    def q(x):
        return 'min %.1f  p10 %.1f  med %.1f  p90 %.1f  max %.1f' % (
            x.min(), np.percentile(x, 10), np.median(x), np.percentile(x, 90), x.max())
    print('  start            ', q(us[:, 0]))
    print('  prologue (0->1)  ', q(us[:, 1] - us[:, 0]))
    print('  main loop (1->2) ', q(us[:, 2] - us[:, 1]))
    print('  epilogue (2->3)  ', q(us[:, 3] - us[:, 2]))
    print('  end              ', q(us[:, 3]))
    ph = t[:, 5:10].astype(np.float64)
    nkt = t[:, 10].astype(np.float64)
    loop_ticks = ph.sum(1)
    loop_us = us[:, 2] - us[:, 1]
    tick_per_us = np.median(loop_ticks / loop_us)
    names = ['issue loads', 'mfma block', 'wait vmcnt(0)', 'lds store', 'barrier']
    print('  wave-0 phases per k step (s_memtime ticks, %.0f ticks/us):' % tick_per_us)
    for i, nm in enumerate(names):
        v = ph[:, i] / nkt
        print('    %-14s med %7.1f  p90 %7.1f   (%.1f%% of the loop)' %
              (nm, np.median(v), np.percentile(v, 90), 100 * ph[:, i].sum() / loop_ticks.sum()))
    per_cu = np.bincount(cuid)
    per_cu = per_cu[per_cu > 0]
    print('  blocks per CU: min %d max %d ; second-round starts (start > 20us): %d' %
          (per_cu.min(), per_cu.max(), int((us[:, 0] > 20).sum())))


if __name__ == '__main__':
    shapes = [('nt', 2046, 4096, 8192), ('nn', 2046, 8192, 4096), ('tn', 4096, 8192, 2046),
              ('nt', 2046, 4096, 1204), ('nt', 2048, 4096, 1024), ('nt', 2048, 4096, 16384)]
    for s in shapes:
        run(*s)
