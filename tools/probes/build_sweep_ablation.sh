#!/bin/bash
# Build container: libcwlt variants whose one-sweep attention backward (csrc/cla_bf16.hip) has parts switched off at
# compile time (-DABL=n; tools/probes/sweep_ablation.patch applied to a COPY of the source):
#   1 no global stores | 2 no loads after the prologue | 4 no MFMA phases | 8 no staging | 16 no phase 2 |
#   32 no phase-1 state MFMAs | 64 no output-tile read-back / column sums
# -> tools/probes/libcwlt_abl<n>.so (git-ignored; they travel to the GPU box), timed there by tools/probes/ab_variants.sh.
# The variants compute wrong results by construction; only their run time is read (profiles/r03_sweep_ablation.txt).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
PKG=$R/reinforcement-learning-in-music-generation_amd
python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.build()" | tail -1
T=$(mktemp -d)
cp $PKG/csrc/*.h $PKG/csrc/cla_bf16.hip $T/
(cd $T && patch -p4 cla_bf16.hip < $R/tools/probes/sweep_ablation.patch)
for n in ${ABLS:-0 1 2 3 11 19 35 67 75 51 123}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form \
      -I $R/include -DABL=$n -c $T/cla_bf16.hip -o $T/cla_abl$n.o &
  if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
done
wait
for n in ${ABLS:-0 1 2 3 11 19 35 67 75 51 123}; do
  objs=$(ls $PKG/csrc/.obj/*.o | grep -v cla_bf16.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -no-hip-rt -o $R/tools/probes/libcwlt_abl$n.so $objs $T/cla_abl$n.o
done
rm -rf $T
