#!/bin/bash
# long randomised parity runs of the round's final kernels
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/fuzz
timeout -k 10 500 python scripts/rdf_fuzz.py 400 101 > gpurun_out/fuzz/rdf.log 2>&1; rc=$?; tail -n 1 gpurun_out/fuzz/rdf.log; [ $rc -ne 0 ] && exit $rc
MDX_RDF_LDS_FLUSH_UNITS=3 timeout -k 10 300 python scripts/rdf_fuzz.py 150 102 > gpurun_out/fuzz/rdf_flush.log 2>&1; rc=$?; tail -n 1 gpurun_out/fuzz/rdf_flush.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/sq_fuzz.py 200 103 > gpurun_out/fuzz/sq.log 2>&1; rc=$?; tail -n 1 gpurun_out/fuzz/sq.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python scripts/msd_fuzz.py 250 104 > gpurun_out/fuzz/msd.log 2>&1; rc=$?; tail -n 1 gpurun_out/fuzz/msd.log; exit $rc
