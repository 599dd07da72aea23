#!/bin/bash
# Round-5 final pass, part 3: traffic entries for C4 in 2, 4, 16, 32 blocks, then the bench lines again with every entry in place.
mkdir -p gpurun_out/counters
MDX_ROUND=r05 timeout -k 10 600 python scripts/make_counters.py msd_c4_b2 msd_c4_b4 msd_c4_b16 msd_c4_b32 > gpurun_out/counters/make_counters_4.log 2>&1; echo "counters rc=$?"
cp gpurun_out/counters/counters.json profiles/counters.json
bash scripts/run/r5final2.sh
