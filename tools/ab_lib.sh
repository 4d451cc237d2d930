#!/bin/bash
# GPU box: A/B of two builds of libcwlt.so on one command, same box, alternating.
# usage: tools/ab_lib.sh BASE.so 'command printing the figure of interest'   (the in-tree library is the candidate)
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=$R/reinforcement-learning-in-music-generation_amd/libcwlt.so
cp $L /tmp/cand.so
for rep in 1 2 3; do
  cp $1 $L && echo "== base" && bash -c "$2"
  cp /tmp/cand.so $L && echo "== candidate" && bash -c "$2"
done
cp /tmp/cand.so $L
