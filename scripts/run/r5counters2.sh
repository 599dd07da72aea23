#!/bin/bash
# Round 5 counters, part 2: MSD entries (PMC + TCC passes) and the kernel statistics of every bench line.
mkdir -p gpurun_out/counters gpurun_out/r5k
timeout -k 10 200 python scripts/diag/onsager_profile.py > gpurun_out/r5k/onsager_profile.txt 2>&1; echo "profile rc=$?"; head -3 gpurun_out/r5k/onsager_profile.txt
MDX_ROUND=r05 timeout -k 10 900 python scripts/make_counters.py msd_c4 msd_c4_b8 msd_tcc stats > gpurun_out/counters/make_counters_2.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/counters/make_counters_2.log | cut -c1-300
ls gpurun_out/counters | wc -l
