#!/bin/bash
# priority-frontier mode: parity tests, throughput probe, kernel trace (run through gpurun)
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
tag=${1:-p}
cd $root
[ -z "${SKIP_TESTS:-}" ] && timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_device_engine.py -x -q -m gpu -k "priority" > $out/${tag}_prio_tests.log 2>&1
grep -E "passed|failed|error" $out/${tag}_prio_tests.log | tail -3
timeout 300 python3 profiles/priority_probe.py 300000 1024 8192 16384 > $out/${tag}_prio_probe.jsonl 2> $out/${tag}_prio_probe.err
cut -c1-330 $out/${tag}_prio_probe.jsonl
bash profiles/trace_cmd.sh ${tag}_prio profiles/priority_probe.py 300000 8192
SFFGPU_PROFILE=1 timeout 300 python3 profiles/priority_probe.py 300000 16384 2>&1 | grep -i -E "fault|fallback" | head -8
