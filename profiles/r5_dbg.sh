#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; out=gpurun_out; mkdir -p $out; rm -f $out/r5_dbg.txt
echo "== dense_3D bench job dbg" >> $out/r5_dbg.txt
SFFGPU_LIB=libsffgpu_dbg.so SFFGPU_PROFILE=1 timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs 2>&1 | grep -E "exact kernel|block query" | tail -4 | cut -c1-400 >> $out/r5_dbg.txt
echo "== building C5 dbg" >> $out/r5_dbg.txt
SFFGPU_LIB=libsffgpu_dbg.so SFFGPU_PROFILE=1 timeout 300 python3 profiles/c5_probe.py 2000000 8192 2>&1 | grep -E "exact kernel" | tail -2| cut -c1-400 >> $out/r5_dbg.txt
cat $out/r5_dbg.txt
