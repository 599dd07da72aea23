#!/bin/bash
out=gpurun_out/r5u; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_engines.py -m gpu -q -k "msd or transform" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log | cut -c1-200
timeout -k 10 120 python scripts/msd_fuzz.py 80 19 > $out/fuzz.log 2>&1; tail -2 $out/fuzz.log; grep -c MISMATCH $out/fuzz.log
