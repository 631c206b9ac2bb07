"""GPU: a plain-C program -- no Python, no torch in the process -- allocates with hipMalloc, calls the C ABI
(gist_in_degree_norm_f32, gist_spmm_csr_f32 in its forward and reversed-accumulate forms, gist_gemm_nt_f32) on a 300-node
multigraph and checks the results against C loops (SURVEY.md section 8b "C ABI": raw pointers + sizes + a stream, error
codes, no allocation inside).  Built here with gcc against include/gist_hip.h and the HIP runtime's C API."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_consumer_runs_device_work(tmp_path):
    from gist_amd import _lib
    src = os.path.join(ROOT, 'tests', 'cabi_device_consumer.c')
    exe = str(tmp_path / 'cabi_device_consumer')
    libdir = os.path.dirname(_lib.LIB_PATH)
    # the process must use ONE HIP runtime: the library resolves libamdhip64 by soname, so link /opt/rocm's first
    cmd = ['gcc', '-std=c99', '-O1', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), '-I', '/opt/rocm/include',
           src, '-o', exe, '-L', libdir, '-l:libgist_hip.so', '-L', '/opt/rocm/lib', '-lamdhip64', '-lm',
           '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib']
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-1000:], r.stderr[-1000:])
    assert 'cabi device consumer ok' in r.stdout
