#!/bin/bash
# The RRT legs with and without the repaired slots (SFFGPU_RRT_REPAIR).  Run from the repo root on the GPU box.
out=gpurun_out/r6_rrt_repair.txt
: > $out
for cfg in "1 150 48" "0 150 1"; do
  set -- $cfg
  for m in rrt star multi; do
    echo "== SFFGPU_RRT_REPAIR=$1 SFFGPU_RRT_GROW=$2 SFFGPU_RRT_SMALL=$3 $m" >> $out
    SFFGPU_PROFILE=1 SFFGPU_RRT_REPAIR=$1 SFFGPU_RRT_GROW=$2 SFFGPU_RRT_SMALL=$3 python profiles/rrt_probe.py 150000 $m 2>&1 | grep -v clearance | tail -3 >> $out
  done
done
