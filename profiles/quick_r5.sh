#!/bin/bash
# quick GPU check of a build (round 5): selected parity tests, the lean bench line, configs[4]; bash profiles/quick_r5.sh <tag> [pytest -k expression]
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
tag=${1:-q5}
kexpr=${2:-"clearance or collide or segment or pose or full_size or c5_building"}
cd $root
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$kexpr" > $out/${tag}_tests.log 2>&1
tail -3 $out/${tag}_tests.log
bash profiles/bench_lean.sh $tag
timeout 300 python3 profiles/c5_probe.py 2000000 8192 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5', round(d['accepted_nodes_per_s']/1e6,3), 'M nodes/s', d['nodes'], 'nodes', {k: round(d[k],1) for k in ('total_ms','host_ms','sweep_ms','collide_ms')})"
