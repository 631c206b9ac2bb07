#!/bin/bash
# Dev tool (gpurun): same-box A/B of the hidden layers' forward projections left as split-K slabs to their LayerNorm
# (the default) against the reduce pass of their own.  Build the B side first:
#   GIST_EXTRA_FLAGS=-DSTEP_NO_YSLABS GIST_LIB_OUT=$PWD/gist_amd/libgist_noys.so python gist_amd/build.py
: ${GRAFT_REPO_ROOT:?run under gpurun}
cd $GRAFT_REPO_ROOT
for a in "--n-hidden 2048" "--n-hidden 1024" "--n-hidden 512" "--config 2" "--config 4 --steps 600"; do
 for lib in hip noys hip noys; do
  GIST_LIB_PATH=$GRAFT_REPO_ROOT/gist_amd/libgist_$lib.so timeout -k 10 200 python bench.py $a --steps 300 --warmup 20 --no-cpu-baseline --no-second-leg --no-kernel-timing 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"$a\", \"$lib\", d[\"ms_per_step\"])" || exit 1
 done
done
