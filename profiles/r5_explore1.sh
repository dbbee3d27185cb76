#!/bin/bash
# round-5 exploration 1: exact-kernel items with / without candidate triangles at several triangle-grid resolutions
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; out=gpurun_out; mkdir -p $out
for div in 3 6 12; do
  echo "== dense_3D bench job, SFFGPU_TG_DIV=$div" >> $out/r5_explore1.txt
  SFFGPU_LIB=libsffgpu_dbg.so SFFGPU_PROFILE=1 SFFGPU_TG_DIV=$div timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs 2>&1 | grep -E "exact kernel|^\{" | cut -c1-700 >> $out/r5_explore1.txt
  echo "== building C5, SFFGPU_TG_DIV=$div" >> $out/r5_explore1.txt
  SFFGPU_LIB=libsffgpu_dbg.so SFFGPU_PROFILE=1 SFFGPU_TG_DIV=$div timeout 300 python3 profiles/c5_probe.py 2000000 8192 2>&1 | grep -E "exact kernel|^\{" | cut -c1-700 >> $out/r5_explore1.txt
done
for div in 3 6; do
  echo "== shipped build, dense_3D bench, SFFGPU_TG_DIV=$div" >> $out/r5_explore1.txt
  bash profiles/bench_lean.sh r5e1_$div SFFGPU_TG_DIV=$div >> $out/r5_explore1.txt 2>&1
  echo "== shipped build, C5, SFFGPU_TG_DIV=$div" >> $out/r5_explore1.txt
  SFFGPU_TG_DIV=$div timeout 300 python3 profiles/c5_probe.py 2000000 8192 2>&1 | tail -1 | cut -c1-400 >> $out/r5_explore1.txt
done
cat $out/r5_explore1.txt
