#!/bin/bash
# kernel statistics (rocprofv3 --kernel-trace --stats) of the default command and every workload on the final sources
mkdir -p gpurun_out/counters
MDX_ROUND=r05 timeout -k 10 1000 python scripts/make_counters.py stats > gpurun_out/counters/make_counters_stats.log 2>&1; echo "rc=$?"
ls gpurun_out/counters/*kernel_stats.csv | wc -l
