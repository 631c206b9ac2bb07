"""CPU: the C-ABI library loads and exports every symbol include/gist_hip.h declares,
the ctypes binding covers exactly that set, and host-only entry points behave."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'gist_hip.h')


def _declared():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(gist_[a-z0-9_]+)\s*\(', src)))


def test_library_built():
    from gist_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'run `python gist_amd/build.py`'


def test_exports_match_header():
    from gist_amd import _lib
    names = _declared()
    assert len(names) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), 'libgist_hip.so does not export %s' % n
    assert sorted(_lib.SIGNATURES) == names, 'ctypes binding and header disagree'


def test_host_only_entry_points():
    from gist_amd import _lib
    L = _lib.load()
    assert L.gist_abi_version() == _lib.ABI_VERSION
    assert L.gist_gemm_workspace_bytes(0, 5, 5) == 0
    prev = L.gist_gemm_get_mode()
    assert L.gist_gemm_set_mode(0) == 0                                # fp32 MFMA everywhere
    assert L.gist_gemm_workspace_bytes(2046, 4096, 8192) == 0          # enough tiles
    assert L.gist_gemm_workspace_bytes(2046, 41, 8192) > 0             # split-K
    assert L.gist_gemm_set_mode(1) == 0                                # f16x3 split on large shapes
    kpad = 8192
    assert L.gist_gemm_workspace_bytes(2046, 4096, 8192) == 49152 + (2046 + 4096) * kpad * 4
    assert L.gist_gemm_workspace_bytes(2046, 4096, 1204) == 49152 + (2046 + 4096) * 1216 * 4
    assert L.gist_gemm_workspace_bytes(2046, 41, 8192) > 0             # skinny: fp32 split-K
    assert L.gist_gemm_workspace_bytes(2046, 41, 8192) < 64 << 20
    assert L.gist_gemm_set_mode(7) == -1 and b'mode' in L.gist_last_error()
    # the step's own split workspace: sized from the plan's shapes, host side only
    P = _lib.StepPlan()
    P.n_layers, P.use_layernorm, P.p_drop = 3, 1, 0.2
    for k, (i, o) in enumerate([(602, 4096), (4096, 4096), (4096, 41)]):
        P.layer[k].n_in, P.layer[k].n_out = i, o
        P.layer[k].ldz, P.layer[k].ldy = 2 * i, o if o % 4 == 0 else 44
    P.n_max, P.feat_absmax = 2200, 5.0
    assert L.gist_gemm_set_mode(1) == 0
    need = L.gist_step_h3_workspace_bytes(ctypes.byref(P))
    # layers 0 and 1 qualify: Zs + ZsT + Ws (+ WsT) + the shared gradient splits
    kp = lambda k: -(-k // 32) * 32
    lower = 4 * (2200 * kp(1204) + 1204 * kp(2200) + 4096 * kp(1204)
                 + 2200 * kp(8192) + 8192 * kp(2200) + 2 * 4096 * kp(8192)
                 + 2200 * kp(4096) + 4096 * kp(2200))
    assert lower <= need < lower + (8 << 20), (need, lower)
    P.feat_absmax = 0.0                                                # unknown bound: layer 0 opts out
    assert 0 < L.gist_step_h3_workspace_bytes(ctypes.byref(P)) < need
    P.use_layernorm = 0                                                # no LayerNorm bound: layer 1 too
    assert L.gist_step_h3_workspace_bytes(ctypes.byref(P)) == 0
    assert L.gist_gemm_set_mode(0) == 0
    P.use_layernorm, P.feat_absmax = 1, 5.0
    assert L.gist_step_h3_workspace_bytes(ctypes.byref(P)) == 0        # mode f32: nothing kept
    assert L.gist_step_h3_workspace_bytes(None) == 0
    assert L.gist_gemm_set_mode(prev) == 0
    assert L.gist_colsum_partials(0) == 0 and L.gist_colsum_partials(129) == 3
    # argument validation happens before any device work
    assert L.gist_spmm_csr_f32(None, None, None, 4, None, 4, 3, 4, None, None, 0, None) == -1
    assert b'null pointer' in L.gist_last_error()
    assert L.gist_adam_f32(None, None, None, None, -1, 0.1, 0.9, 0.999, 1e-8, 0.0, 1, None) == -1


def test_cpu_tensors_are_rejected():
    import torch
    from gist_amd import hip
    x = torch.zeros(4, 4)
    rp = torch.zeros(5, dtype=torch.int32)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        hip.spmm(rp, rp, x, x.clone())


def test_missing_library_is_loud(monkeypatch):
    from gist_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libgist_hip.so')
    with pytest.raises(_lib.GistLibraryError, match='no CPU fallback'):
        _lib.load()


def test_plain_c_consumer_compiles_links_and_runs(tmp_path):
    """include/gist_hip.h is valid C99 and libgist_hip.so links from a C program (no Python,
    no torch): the boundary really is a C ABI."""
    import subprocess
    from gist_amd import _lib
    src = os.path.join(ROOT, 'tests', 'cabi_consumer.c')
    exe = str(tmp_path / 'cabi_consumer')
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = ['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), src,
           '-o', exe, '-L', libdir, '-l:libgist_hip.so', '-Wl,-rpath,' + libdir,
           '-Wl,-rpath,/opt/rocm/lib']
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-500:])
    assert 'cabi consumer ok' in r.stdout
