#!/bin/bash
out=gpurun_out/r5fuzz; mkdir -p $out
timeout -k 10 330 python scripts/msd_fuzz.py 300 13 > $out/msd_fuzz_long_shapes.txt 2>&1; tail -2 $out/msd_fuzz_long_shapes.txt
