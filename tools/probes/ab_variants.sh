#!/bin/bash
# GPU box: the one-sweep backward under each ablation build tools/probes/libcwlt_abl<n>.so (bit 1 = no global stores,
# 2 = no loads after the prologue, 4 = no MFMA phases, ...: tools/probes/build_sweep_ablation.sh builds them); results are
# wrong by construction, only the time is read.
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=$R/reinforcement-learning-in-music-generation_amd/libcwlt.so
cp $L /tmp/keep.so
for n in ${ABLS:-0 1 2 3 11 19 35 67 75 51 123 0}; do
  cp $R/tools/probes/libcwlt_abl$n.so $L
  echo "ABL=$n $(python3 $R/tools/bench_kernels.py 512 1024 2>&1 | grep bfloat16 | grep 'one sweep')"
done
cp /tmp/keep.so $L
