#!/bin/bash
# dev tool: builds the split GEMM without its DMA / without its MFMAs and times the big shapes
set -e
cd "$(dirname "$0")/.."
for v in NO_DMA NO_MFMA; do
  GIST_EXTRA_FLAGS="-DH3_PROBE_$v" GIST_LIB_OUT=$PWD/gist_amd/libgist_hip_$v.so python gist_amd/build.py > /dev/null
done
