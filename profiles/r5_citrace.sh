#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; out=gpurun_out; mkdir -p $out; rm -f $out/r5_citrace.txt
for L in 300 500 700; do
echo "== k_collide_items launch $L of the bench job" >> $out/r5_citrace.txt
SFFGPU_LIB=libsffgpu_ci$L.so SFFGPU_PROFILE=1 timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 0 --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs 2>&1 | grep -A18 "k_collide_items trace" | tail -19 >> $out/r5_citrace.txt
done
cat $out/r5_citrace.txt
