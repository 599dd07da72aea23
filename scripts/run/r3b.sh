#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r3b_pytest.log 2>&1
rc=$?; tail -5 gpurun_out/r3b_pytest.log; echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 200 python scripts/run/diag_class.py > gpurun_out/r3b_diag_class.txt 2>&1; rc=$?; cat gpurun_out/r3b_diag_class.txt | tail -40
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 200 python bench.py --workload ingest --steps 2 > gpurun_out/r3b_ingest.json 2> gpurun_out/r3b_ingest.err || exit $?
python -c "
import json;d=json.load(open('gpurun_out/r3b_ingest.json'));print('resident',d['frames_per_sec']);d=d['extra']['ingest']
print({k:(round(v['frames_per_sec']),round(v['ratio_to_resident'],3)) for k,v in d.items() if isinstance(v,dict) and 'frames_per_sec' in v}, d.get('error'))"
