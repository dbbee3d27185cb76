#!/bin/bash
# The RRT legs with the replay's walk done ahead (SFFGPU_RRT_DRY: edges only for the rows the replay takes) and other growth rules
out=gpurun_out/r6_rrt_dry.txt
: > $out
for cfg in "1 150" "1 200" "1 300" "1 400" "0 150"; do
  set -- $cfg
  for m in rrt star multi; do
    echo "== SFFGPU_RRT_DRY=$1 SFFGPU_RRT_GROW=$2 $m" >> $out
    SFFGPU_PROFILE=1 SFFGPU_RRT_DRY=$1 SFFGPU_RRT_GROW=$2 python profiles/rrt_probe.py 150000 $m 2>&1 | grep -v clearance | tail -3 | cut -c1-330 >> $out
  done
done
