#!/bin/bash
# the exchange of the sharded forest with ONE rank (pack -> all-gather -> unpack in every round), three runs each of
#   torch:  the caller (Python, torch.distributed) issues the collective between dev_round_eval and dev_round_commit
#   native: the library's own RCCL communicator; whole waves are enqueued by sffgpu_forest_run, one ahead
# -> gpurun_out/r6_force_dist.jsonl (one line per run: driver, value, the per-round budget)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
: > gpurun_out/r6_force_dist.jsonl
for rep in 1 2 3; do
  for drv in torch native; do
    extra="--torch-exchange"; [ $drv = native ] && extra=""
    timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --force-dist $extra --cpu-iters 0 --no-sweep-micro --no-wave-sweep --no-extra-legs 2>/dev/null | tail -1 |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'driver':'$drv','rep':$rep,'value':d['value'],'ms_per_step':d['ms_per_step'],'parallelism':d['config']['parallelism'],'dist_budget_us_per_round':d.get('dist_budget_us_per_round')}))" >> gpurun_out/r6_force_dist.jsonl
  done
done
cat gpurun_out/r6_force_dist.jsonl
