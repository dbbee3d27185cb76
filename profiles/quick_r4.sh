#!/bin/bash
# quick GPU check of a build: smoke, the device-engine parity tests, the lean bench line (run through gpurun)
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
tag=${1:-q}
cd $root
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1; echo "smoke rc $?" >> $out/${tag}_smoke.log
timeout 900 python3 -m pytest tests/test_gpu_device_engine.py -x -q -m gpu > $out/${tag}_dev_tests.log 2>&1
tail -3 $out/${tag}_dev_tests.log
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs > $out/${tag}_bench_lean.log 2>&1
tail -1 $out/${tag}_bench_lean.log | cut -c1-1500
tail -5 $out/${tag}_smoke.log
