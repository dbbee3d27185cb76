#!/bin/bash
# configs[4] with the SFF* passes as one launch (k_star_tail) and as the fixed chain, same box, alternating
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
for i in 1 2 3; do for T in 1 0; do
  echo -n "tail=$T ${EXTRA_ENV:-} : "
  env SFFGPU_STAR_TAIL=$T ${EXTRA_ENV:-} timeout 120 python3 profiles/c5_probe.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f M nodes/s  %.2f ms  fallbacks %d passes %d' % (d['accepted_nodes_per_s']/1e6, d['seconds']*1e3, d['host_fallback_waves'], d['star_passes']))"
done; done
