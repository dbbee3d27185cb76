#!/bin/bash
# copies what collect_r5.sh left in gpurun_out/ into profiles/ under the names the documents use (run in the build container)
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
o=$root/gpurun_out; p=$root/profiles
cpn() { [ -s "$o/$1" ] && cp "$o/$1" "$p/$2" || echo "missing $1"; }
cpn r5_bench_kernel_stats.csv r5_bench_kernel_stats.csv
cpn r5_bench_pmc_summary.json r5_bench_pmc_summary.json
cpn r5_bench_bench_line.json r5_bench_line.json
cpn r5_bench_bench_line_under_trace.json r5_bench_line_under_trace.json
cpn r5_sq_summary.json r5_sq_summary.json
cpn r5_query_counters.json r5_query_counters.json
cpn r5_query_no_order_counters.json r5_query_counters_no_order.json
cpn r5_sweep_kernel_stats.csv r5_sweep_kernel_stats.csv
cpn r5_sweep_pmc_summary.json r5_sweep_pmc_summary.json
cpn r5_sweep_line.json r5_sweep_line.json
cpn r5_c5_kernel_stats.csv r5_c5_kernel_stats.csv
cpn r5_c5_probe.json r5_c5_probe.json
cpn r5_tail_ab.txt r5_tail_ab.txt
cpn r5_c5_star_clocks.txt r5_c5_star_clocks.txt
cpn r5_star_chain.txt r5_star_chain.txt
cpn r5_force_dist_line.json r5_force_dist_line.json
cpn r5_phase_clocks.txt r5_phase_clocks.txt
cpn r5_small_waves.txt r5_small_waves.txt
cpn r5_rrt_probe.jsonl r5_rrt_probe.jsonl
cpn r5_priority_probe.jsonl r5_priority_probe.jsonl
cpn r5_bench_lean_no_order.txt r5_bench_lean_no_order.txt
cpn r5_bench_lean_order.txt r5_bench_lean_order.txt
cpn r5_full_gpu_tests.log r5_full_gpu_tests.log
