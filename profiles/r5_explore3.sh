#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; out=gpurun_out; mkdir -p $out; rm -f $out/r5_explore3.txt
for rep in 1 2; do
for cells in 134217728 268435456 536870912; do
echo "== dense shipped, cells $cells" >> $out/r5_explore3.txt
bash profiles/bench_lean.sh r5e3 SFFGPU_CLEAR_CELLS=$cells >> $out/r5_explore3.txt 2>&1
done
for hdiv in 2 4 8 16; do
echo "== c5 shipped, hdiv $hdiv" >> $out/r5_explore3.txt
SFFGPU_PROFILE=1 SFFGPU_CLEAR_HDIV=$hdiv timeout 300 python3 profiles/c5_probe.py 2000000 8192 2>&1 | grep -E "clearance bits|^\{" | cut -c1-330 >> $out/r5_explore3.txt
done
done
cat $out/r5_explore3.txt
