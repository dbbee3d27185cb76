#!/bin/bash
# A/B of two builds on the bench job and configs[4]: bash profiles/r5_ab.sh <variant lib name> [reps]
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; out=gpurun_out; mkdir -p $out
v=$1; reps=${2:-2}
for rep in $(seq 1 $reps); do
  for lib in libsffgpu.so $v; do
    echo "== $lib"
    bash profiles/bench_lean.sh ab_$rep SFFGPU_LIB=$lib
    SFFGPU_LIB=$lib timeout 300 python3 profiles/c5_probe.py 2000000 8192 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5', round(d['accepted_nodes_per_s']/1e6,3), 'M nodes/s', d['nodes'], 'nodes', {k: round(d[k],1) for k in ('total_ms','host_ms','sweep_ms','collide_ms')})"
  done
done
