#!/bin/bash
# GPU box: A/B of two builds of libcwlt.so on the scan micro-benchmark (tools/bench_kernels.py), same box, alternating.
# usage: tools/ab_scan.sh BASE.so  (the in-tree library is the candidate)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=$R/reinforcement-learning-in-music-generation_amd/libcwlt.so
cp $L /tmp/cand.so
for rep in 1 2; do
  cp $1 $L && echo "== base" && python3 $R/tools/bench_kernels.py 512 1024 2>&1 | grep bfloat16
  cp /tmp/cand.so $L && echo "== candidate" && python3 $R/tools/bench_kernels.py 512 1024 2>&1 | grep bfloat16
done
cp /tmp/cand.so $L
