"""In-kernel clock and matrix-pipe share of the bf16x3 GEMM's k loop (dev tool): a diagnostic build
(-DB3_CLOCK_PROBE) stamps s_memtime / s_memrealtime around the k loop of every workgroup; clock =
d(memtime) / d(memrealtime) x 100 MHz after >= 2 s of back-to-back launches; pipe share = the
loop's MFMA cycles (k tiles x 96 MFMAs x 16 cycles x 2 waves per SIMD) / its core cycles.

    GIST_EXTRA_FLAGS=-DB3_CLOCK_PROBE GIST_LIB_OUT=$PWD/gist_amd/libgist_hip_CLK.so python gist_amd/build.py
    GIST_LIB_PATH=$PWD/gist_amd/libgist_hip_CLK.so python scripts/b3_clock_probe.py
"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gist_amd import hip, _lib

dev = torch.device('cuda', 0)
L = _lib.load()
L.gist_b3_clock_read.argtypes = [ctypes.c_void_p, ctypes.c_int64]
L.gist_b3_clock_read.restype = ctypes.c_int
hip.gemm_mode('bf16x3')
for (m, n, k) in [(2046, 4096, 8192), (4096, 8192, 2046), (2046, 4096, 4096)]:
    a, w = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev)
    y = torch.empty(m, n, device=dev)
    t0 = time.time()
    while time.time() - t0 < 2.5:
        for _ in range(50):
            hip.gemm_nt(a, w, None, y)
        torch.cuda.synchronize()
    nb = min(4096, -(-m // 256) * -(-n // 128))
    buf = np.zeros(2 * nb, np.uint64)
    assert L.gist_b3_clock_read(buf.ctypes.data, nb) == 0
    cyc, rt = buf[0::2].astype(np.float64), buf[1::2].astype(np.float64)
    ok = rt > 0
    clk = cyc[ok] / rt[ok] * 100e6 / 1e9
    n_kt = -(-k // 64) * 2
    print('m=%d n=%d k=%d: in-kernel clock median %.3f GHz (p10 %.3f, p90 %.3f) over %d workgroups; '
          'k loop %.1f us = %.0f cycles, MFMA cycles %d -> pipe share %.3f'
          % (m, n, k, np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90), ok.sum(),
             np.median(rt[ok]) / 100.0, np.median(cyc[ok]), n_kt * 3072, n_kt * 3072 / np.median(cyc[ok])),
          flush=True)
