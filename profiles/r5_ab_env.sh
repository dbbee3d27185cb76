#!/bin/bash
# A/B of run-time toggles on the bench job and configs[4]: bash profiles/r5_ab_env.sh <reps> "<ENV=1 ...>" "<ENV=1 ...>" ...   ("-" = no toggle)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; out=gpurun_out; mkdir -p $out
reps=$1; shift
for rep in $(seq 1 $reps); do
  for tog in "$@"; do
    [ "$tog" = "-" ] && tog="SFFGPU_DUMMY=1"
    echo "== $tog"
    bash profiles/bench_lean.sh ab_$rep $tog
    env $tog timeout 300 python3 profiles/c5_probe.py 2000000 8192 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5', round(d['accepted_nodes_per_s']/1e6,3), 'M nodes/s', d['nodes'], 'nodes', {k: round(d[k],1) for k in ('total_ms','host_ms','sweep_ms','collide_ms')})"
  done
done
