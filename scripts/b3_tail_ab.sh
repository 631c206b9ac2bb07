#!/bin/bash
# A/B of gemm_b3's tail units: prints the probe under each variant library in ab/
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in gist_amd/libgist_hip.so ab/*.so; do
  echo "== $lib"
  GIST_LIB_PATH=$PWD/$lib PYTHONPATH=. python scripts/b3_tail_probe.py 2046 2100 2>&1 | grep -v amdgpu.ids
done
