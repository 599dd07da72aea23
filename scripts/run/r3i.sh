#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "rdf or radial or c2 or c5 or beyond or traj or smoke" > gpurun_out/r3i_pytest.log 2>&1
rc=$?; tail -n 3 gpurun_out/r3i_pytest.log; if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python scripts/rdf_fuzz.py 120 201 > gpurun_out/r3i_fuzz.log 2>&1; rc=$?; tail -n 1 gpurun_out/r3i_fuzz.log; if [ $rc -ne 0 ]; then exit $rc; fi
python scripts/run/diag_single.py
