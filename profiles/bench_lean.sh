#!/bin/bash
# lean bench line of the current build (no CPU baseline / extra legs): bash profiles/bench_lean.sh <tag> [env assignments...]
tag=${1:-b}; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
env "$@" timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs > gpurun_out/${tag}_bench_lean.log 2>&1
tail -1 gpurun_out/${tag}_bench_lean.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', 'query us', round(d['roofline']['avg_launch_us'],2), 'device clock', round(d['roofline'].get('avg_launch_us_by_device_clock_every_workgroup') or 0,2), 'Mnodes/s', round(d['value']/1e6,3), 'nodes', d['config']['nodes_at_end'], {k: round(v,1) for k,v in d['time_split_ms'].items()})"
