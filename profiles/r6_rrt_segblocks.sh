#!/bin/bash
# RRT*: launch width of the exact edge kernel (SFFGPU_SEG_BLOCKS, 512 by default) now that a wave brings ~26 k edges per launch
out=gpurun_out/r6_rrt_segblocks.txt
: > $out
for b in 256 512 768 1024 2048; do
  echo "== SFFGPU_SEG_BLOCKS=$b" >> $out
  SFFGPU_PROFILE=1 SFFGPU_SEG_BLOCKS=$b python profiles/rrt_probe.py 150000 star 2>&1 | grep -E "run_wave|iterations_per_s" | tail -2 | cut -c1-330 >> $out
done
