#!/bin/bash
# The RRT legs with other wave-growth rules (SFFGPU_RRT_GROW: the next wave speculates that many % of what survived a cut wave)
# and with the other-trees query outside the chain.  Run from the repo root on the GPU box.
out=gpurun_out/r6_rrt_grow.txt
: > $out
for g in 150 200 300 400 600; do
  for m in rrt star multi; do
    echo "== SFFGPU_RRT_GROW=$g $m" >> $out
    SFFGPU_RRT_GROW=$g python profiles/rrt_probe.py 150000 $m >> $out 2>&1
  done
done
echo "== other-trees query outside the chain, multi" >> $out
SFFGPU_RRT_NO_CHAIN_CONN=1 SFFGPU_PROFILE=1 python profiles/rrt_probe.py 150000 multi >> $out 2>&1
echo "== inside (default), multi" >> $out
SFFGPU_PROFILE=1 python profiles/rrt_probe.py 150000 multi >> $out 2>&1
