#!/bin/bash
# The suite and the default line once more on the tree as it is committed.
out=gpurun_out/r5last; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log | cut -c1-300
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
( time timeout -k 10 500 python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2>&1 | grep real; echo "bench rc=$?"
tail -c 900 $out/bench_default.json
