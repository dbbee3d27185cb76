#!/bin/bash
# configs[4] with heavy survivors emitted as 16-sample items (variants built with -DSFFK_SPLIT_MIN=<n>)
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
for rep in 1 2; do
for lib in libsffgpu.so libsffgpu_split16.so libsffgpu_split24.so; do
  SFFGPU_LIB=$lib timeout 300 python3 profiles/c5_probe.py 2000000 8192 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['accepted_nodes_per_s']/1e6,3), 'M nodes/s', {k: round(d[k],1) for k in ('total_ms','collide_ms')})"
done; done
