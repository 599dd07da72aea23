#!/bin/bash
# regular quad items for q_max-filtered sets: parity (tests, fuzz), then rates by set shape
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "sq or structure or isf or scatter or ssf" > gpurun_out/r3k_pytest.log 2>&1
rc=$?; tail -n 3 gpurun_out/r3k_pytest.log; if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python scripts/sq_fuzz.py 150 77 > gpurun_out/r3k_fuzz.log 2>&1; rc=$?; tail -n 3 gpurun_out/r3k_fuzz.log; if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python scripts/run/diag_sq_forms.py > gpurun_out/r3k_forms.txt 2>&1 || { tail -5 gpurun_out/r3k_forms.txt; exit 1; }
cat gpurun_out/r3k_forms.txt
MDX_SQ_NO_REGULAR=1 timeout -k 10 300 python scripts/run/diag_sq_forms.py > gpurun_out/r3k_forms_general.txt 2>&1 || exit 1
cat gpurun_out/r3k_forms_general.txt
