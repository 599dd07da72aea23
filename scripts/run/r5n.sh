#!/bin/bash
out=gpurun_out/r5n; mkdir -p $out
timeout -k 5 200 python scripts/diag/mapped_route_probe.py > $out/probe.log 2>&1; echo "probe rc=$?"; tail -20 $out/probe.log | cut -c1-400
