#!/bin/bash
# Round-5 final pass, part 1 (run through gpurun): the GPU suite once, smoke, randomised parity, MSD counters on the final
# sources.
out=gpurun_out/r5final; mkdir -p $out gpurun_out/counters
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log | cut -c1-300
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
{ timeout -k 10 150 python scripts/rdf_fuzz.py 100 5; timeout -k 10 150 python scripts/sq_fuzz.py 100 5; timeout -k 10 150 python scripts/msd_fuzz.py 100 5; } > $out/parity_fuzz.txt 2>&1; grep -i "done\|mismatch" $out/parity_fuzz.txt | tail -6
cp profiles/counters.json gpurun_out/counters/counters.json
MDX_ROUND=r05 timeout -k 10 600 python scripts/make_counters.py msd_c4 msd_c4_b8 stats_msd > gpurun_out/counters/make_counters_3.log 2>&1; echo "counters rc=$?"
