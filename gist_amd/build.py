"""Build libgist_hip.so (the C-ABI HIP library, include/gist_hip.h) for gfx950.

hipcc cross-compiles without a GPU.  The .so is built IN-TREE
(gist_amd/libgist_hip.so) so it travels with the repo snapshot to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
# dev knobs for A/B builds of kernel variants: extra -D flags and a different output name
OUT = os.environ.get('GIST_LIB_OUT', os.path.join(HERE, 'libgist_hip.so'))
EXTRA = os.environ.get('GIST_EXTRA_FLAGS', '').split()
SOURCES = ['capi.hip', 'spmm.hip', 'spmm_mfma.hip', 'spmm_dense32.hip', 'gemm.hip', 'gemm_h3.hip', 'gemm_b3.hip', 'gemm_b3c.hip', 'rowops.hip', 'classlayer.hip', 'subgraph.hip', 'step.hip',
           'partition.hip', 'prep.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall',
         '-Wno-unused-function', '-fno-gpu-rdc']


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False):
    objdir = os.path.join(HERE, 'build' if not EXTRA else 'build_' + '_'.join(
        f.strip('-').replace('=', '_') for f in EXTRA))
    os.makedirs(objdir, exist_ok=True)
    deps = [os.path.join(CSRC, 'common.h'), os.path.join(HERE, '..', 'include', 'gist_hip.h')]
    objs, procs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or _newer(src, obj) or any(_newer(d, obj) for d in deps):
            cmd = [HIPCC] + FLAGS + EXTRA + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % s)
    if procs or not os.path.exists(OUT):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
