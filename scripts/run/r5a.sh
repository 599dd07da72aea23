#!/bin/bash
# Round 5, first GPU pass: the suite once, smoke, the default bench line, the MSD lines.
out=gpurun_out/r5a; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -5 $out/pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
tail -c 1500 $out/bench_default.json
