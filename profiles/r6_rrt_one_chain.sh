#!/bin/bash
# The RRT legs with the repaired rows inside the wave's one chain (the device lists them; default) and as a second chain
out=gpurun_out/r6_rrt_one_chain.txt
: > $out
for oc in 1 0; do
  for m in rrt star multi; do
    echo "== SFFGPU_RRT_ONE_CHAIN=$oc $m" >> $out
    SFFGPU_PROFILE=1 SFFGPU_RRT_ONE_CHAIN=$oc python profiles/rrt_probe.py 150000 $m 2>&1 | grep -v clearance | tail -3 | cut -c1-330 >> $out
  done
done
