#!/bin/bash
out=gpurun_out/r5fuzz; mkdir -p $out
{ timeout -k 10 330 python scripts/rdf_fuzz.py 300 11; timeout -k 10 330 python scripts/sq_fuzz.py 300 11; timeout -k 10 330 python scripts/msd_fuzz.py 300 11; } > $out/parity_fuzz_long.txt 2>&1
grep -i "done" $out/parity_fuzz_long.txt
