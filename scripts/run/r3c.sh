#!/bin/bash
# counters of the small / large RDF sizes and the S(q) kernel (debug of scripts/make_counters.py), MSD TCC passes
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python scripts/make_counters.py "$@" > gpurun_out/r3c_counters.log 2>&1
rc=$?; tail -60 gpurun_out/r3c_counters.log; exit $rc
