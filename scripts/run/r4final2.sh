#!/bin/bash
# Round-4, final sources: MSD counters, MSD kernel stats at every row kernel, the bench lines for profiles/ (run through gpurun).
mkdir -p gpurun_out/counters gpurun_out/r4final
cp profiles/counters.json gpurun_out/counters/counters.json
MDX_ROUND=r04 python scripts/make_counters.py msd_c4 msd_c4_b8 msd_c4_b2 msd_c4_b4 msd_c4_b16 msd_c4_b32 msd_tcc stats_msd > gpurun_out/r4final/make_counters.log 2>&1 || { tail -20 gpurun_out/r4final/make_counters.log; exit 1; }
cp gpurun_out/counters/counters.json profiles/counters.json
bash scripts/run/r4final.sh
