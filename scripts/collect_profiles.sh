#!/bin/bash
# Copy what scripts/make_counters.py left under gpurun_out/counters/ into profiles/ (tracked), named per round.
round=${1:-r03}
cd "$(dirname "$0")/.."
src=gpurun_out/counters
for f in $src/*_pmc.csv $src/*_kernel_stats.csv $src/*_kernel_medians.csv $src/*_bench_under_rocprof.json $src/*_tcc_summary.json; do
  [ -f "$f" ] && cp "$f" "profiles/${round}_$(basename "$f")"
done
cp $src/counters.json profiles/counters.json
ls profiles/${round}_* | wc -l
