#!/bin/bash
out=gpurun_out/r5f; mkdir -p $out
timeout -k 5 70 python scripts/diag/mmap_truncate_probe.py > $out/probe.log 2>&1; echo "probe rc=$?"; cat $out/probe.log | tail -30
