#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; out=gpurun_out; mkdir -p $out; rm -f $out/r5_explore2.txt
for cells in 134217728 536870912; do
echo "== dense_3D bench job dbg, cells $cells" >> $out/r5_explore2.txt
SFFGPU_CLEAR_CELLS=$cells SFFGPU_LIB=libsffgpu_dbg.so SFFGPU_PROFILE=1 timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs 2>&1 | grep -E "exact kernel|clearance bits" | tail -3 | cut -c1-400 >> $out/r5_explore2.txt
echo "== building C5 dbg, cells $cells" >> $out/r5_explore2.txt
SFFGPU_CLEAR_CELLS=$cells SFFGPU_LIB=libsffgpu_dbg.so SFFGPU_PROFILE=1 timeout 300 python3 profiles/c5_probe.py 2000000 8192 2>&1 | grep -E "exact kernel|clearance bits" | tail -3| cut -c1-400 >> $out/r5_explore2.txt
echo "== shipped, cells $cells" >> $out/r5_explore2.txt
bash profiles/bench_lean.sh r5e2 SFFGPU_CLEAR_CELLS=$cells >> $out/r5_explore2.txt 2>&1
SFFGPU_CLEAR_CELLS=$cells timeout 300 python3 profiles/c5_probe.py 2000000 8192 2>&1 | tail -1 | cut -c1-330 >> $out/r5_explore2.txt
done
cat $out/r5_explore2.txt
