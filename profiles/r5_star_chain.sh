#!/bin/bash
# SFF* as one launch per pass (SFFGPU_STAR_TAIL=0), configs[4], launch by launch (eager): duration of k_star_pass / k_star_exact by
# their position in the round (the grid of an idle launch is as large as a working one's: ~4.4 us under the profiler)
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
SFFGPU_STAR_TAIL=0 SFFGPU_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/star_chain -o t -- python3 $root/profiles/c5_probe.py 2000000 8192 > $out/star_chain.log 2>&1
python3 - $out/star_chain/t_kernel_trace.csv > $out/r5_star_chain.txt <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
pos={"pass":0,"exact":0}
acc=collections.defaultdict(list)
gaps=[]
prev_end=None
inchain=False
for r in rows:
    n=r["Kernel_Name"]; s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    if "k_star_knn" in n: pos={"pass":0,"exact":0}
    for k in ("pass","exact"):
        if "k_star_"+k in n:
            acc[(k,pos[k])].append((e-s)/1e3); pos[k]+=1
            if prev_end is not None: gaps.append((s-prev_end)/1e3)
    prev_end=e
for k in ("pass","exact"):
    for p in range(8):
        v=acc.get((k,p))
        if not v: continue
        print("%-5s %d: n %4d avg %7.2f us  max %7.2f  (longer than 6 us: %4d)" % (k,p,len(v),sum(v)/len(v),max(v),sum(1 for x in v if x>6.0)))
gaps.sort()
print("gap before a chain launch (eager): median %.2f us, mean %.2f" % (gaps[len(gaps)//2], sum(gaps)/len(gaps)))
PY
rm -rf $out/star_chain
cat $out/r5_star_chain.txt
