#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; out=gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_multiprocess.py -x -q -m gpu > $out/r5_mp.log 2>&1; grep -E "passed|failed|Error|assert" $out/r5_mp.log | tail -8
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --force-dist --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs 2>/dev/null | tail -1 > $out/r5_force_dist_line.json
python3 -c "
import json; d=json.load(open('$out/r5_force_dist_line.json')); print(d['value'], d.get('dist_budget_us_per_round'))"
bash profiles/bench_lean.sh r5mp
SFFGPU_PROFILE=1 python3 -c "
import sys; sys.path.insert(0,'tests')
import common, space_filling_forest_star_amd as S
sc=common.scenario('dense3d'); ctx=S.Context(0); ctx.upload_env(sc['env']); ctx.upload_robot(sc['robot'])
" 2>&1 | grep clearance
