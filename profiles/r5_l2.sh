#!/bin/bash
# L2 counters + FETCH_SIZE of the query kernel with / without the spatial order
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
bash profiles/collect_counters.sh r5_ord "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" > /dev/null 2>&1
export SFFGPU_NO_ORDER=1
bash profiles/collect_counters.sh r5_noord "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" > /dev/null 2>&1
python3 - <<'PY'
import json
for t in ("r5_ord","r5_noord"):
    d=json.load(open("gpurun_out/%s_counters.json"%t))
    for k,v in d.items():
        if "query_block" in k or "collide_items" in k or "k_commit" in k: print(t,k[:40],{a:round(b) for a,b in v.items()})
PY
