"""Where a k step of the convert-on-load bf16x3 GEMM goes (dev tool).  A diagnostic build (-DC3_PROBE) stamps
s_memtime in one consumer wave (wave 0) and one producer wave (wave 4) of every workgroup: cycles at the
barriers, in the work between them, and (producers) in convert + LDS write vs issuing the global loads.

    GIST_EXTRA_FLAGS=-DC3_PROBE GIST_LIB_OUT=$PWD/gist_amd/libgist_c3probe.so python gist_amd/build.py
    GIST_LIB_PATH=$PWD/gist_amd/libgist_c3probe.so python scripts/b3c_probe.py
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gist_amd import hip, _lib

dev = torch.device('cuda', 0)
L = _lib.load()
L.gist_c3_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int64]
L.gist_c3_probe_read.restype = ctypes.c_int
hip.gemm_mode('bf16x3')
hip.tuning('b3c', 2)
for tile in (64, 128, 128128):
    hip.tuning('gemm_tile', tile)
    hip.tuning('gemm_splits', 1)
    for (m, n, k) in [(2046, 512, 1024), (2046, 1024, 2048), (2046, 2048, 4096)]:
        a, w = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev)
        y = torch.empty(m, n, device=dev)
        for _ in range(20):
            hip.gemm_nt(a, w, None, y)
        torch.cuda.synchronize()
        nb = min(4096, -(-m // min(tile, 128)) * -(-n // (128 if tile > 128 else 64)))
        buf = np.zeros(16 * nb, np.uint64)
        assert L.gist_c3_probe_read(buf.ctypes.data, nb) == 0
        b = buf.reshape(nb, 16).astype(np.float64)
        steps = k // 32
        med = np.median(b, axis=0) / steps
        print('tile %3d x 64, %d x %d x %d (%d workgroups, %d k steps): cycles per k step -- consumer loop %.0f = barrier '
              '%.0f + reads issue & MFMAs %.0f + fragment wait %.0f | producer loop %.0f = LDS writes %.0f + wait for the '
              'loads %.0f + convert %.0f + load issue %.0f + barrier %.0f' % (
                  tile, m, n, k, nb, steps, med[0], med[1], med[2], med[3], med[4], med[5], med[6], med[8], med[9], med[7]),
              flush=True)
