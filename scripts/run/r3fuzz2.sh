#!/bin/bash
# long randomised parity runs on the final sources of round 3 (new seeds; the RDF fuzz also with the two-stream form
# and with many small slabs)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/fuzz2
timeout -k 10 500 python scripts/rdf_fuzz.py 300 211 > gpurun_out/fuzz2/rdf.log 2>&1; rc=$?; tail -n 1 gpurun_out/fuzz2/rdf.log; [ $rc -ne 0 ] && exit $rc
MDX_RDF_OVERLAP=1 MDX_RDF_SLAB_BYTES=400000 timeout -k 10 300 python scripts/rdf_fuzz.py 150 212 > gpurun_out/fuzz2/rdf_overlap_small_slabs.log 2>&1; rc=$?; tail -n 1 gpurun_out/fuzz2/rdf_overlap_small_slabs.log; [ $rc -ne 0 ] && exit $rc
MDX_RDF_SLAB_BYTES=400000 MDX_RDF_LDS_FLUSH_UNITS=3 timeout -k 10 300 python scripts/rdf_fuzz.py 150 213 > gpurun_out/fuzz2/rdf_small_slabs_flush.log 2>&1; rc=$?; tail -n 1 gpurun_out/fuzz2/rdf_small_slabs_flush.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python scripts/sq_fuzz.py 300 214 > gpurun_out/fuzz2/sq.log 2>&1; rc=$?; tail -n 1 gpurun_out/fuzz2/sq.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python scripts/msd_fuzz.py 250 215 > gpurun_out/fuzz2/msd.log 2>&1; rc=$?; tail -n 1 gpurun_out/fuzz2/msd.log; exit $rc
