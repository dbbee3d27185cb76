#!/bin/bash
# copies what collect_r4.sh left in gpurun_out/ into profiles/ under the names the documents use (run in the build container)
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
o=$root/gpurun_out; p=$root/profiles
cpn() { [ -s "$o/$1" ] && cp "$o/$1" "$p/$2" || echo "missing $1"; }
cpn r4_bench_kernel_stats.csv r4_bench_kernel_stats.csv
cpn r4_bench_pmc_summary.json r4_bench_pmc_summary.json
cpn r4_bench_bench_line.json r4_bench_line.json
cpn r4_bench_bench_line_under_trace.json r4_bench_line_under_trace.json
cpn r4_sq_summary.json r4_sq_summary.json
cpn r4_query_counters.json r4_query_counters.json
cpn r4_c5_kernel_stats.csv r4_c5_kernel_stats.csv
cpn r4_c5_probe.json r4_c5_probe.json
cpn r4_priority_kernel_stats.csv r4_priority_kernel_stats.csv
cpn r4_priority_probe.jsonl r4_priority_probe.jsonl
cpn r4_priority_probe_host_engine.jsonl r4_priority_probe_host_engine.jsonl
cpn r4_small_waves.txt r4_small_waves.txt
cpn r4_force_dist_line.json r4_force_dist_line.json
cpn r4_phase_clocks.txt r4_phase_clocks.txt
cpn r4_rrt_star_kernel_stats.csv r4_rrt_star_kernel_stats.csv
cpn r4_rrt_probe.jsonl r4_rrt_probe.jsonl
cpn r4_rrt_knn_split.txt r4_rrt_knn_split.txt
cpn r4_knn_microbench.jsonl r4_knn_microbench.jsonl
cpn r4_heap_microbench.txt r4_heap_microbench.txt
cpn r4_full_gpu_tests.log r4_full_gpu_tests.log
