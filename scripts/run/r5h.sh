#!/bin/bash
out=gpurun_out/r5h; mkdir -p $out
timeout -k 10 400 python scripts/diag/register_slices.py > $out/register_slices.json 2> $out/err.log; echo "rc=$?"; cat $out/register_slices.json; tail -5 $out/err.log
