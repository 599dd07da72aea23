#!/bin/bash
# MSD after a kernel change: parity tests, fuzz, bench with / without the aligned chunk heads
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "msd or onsager or correl or polymer or EndToEnd or c4" > gpurun_out/r3e_pytest.log 2>&1
rc=$?; tail -4 gpurun_out/r3e_pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python scripts/msd_fuzz.py 60 77 > gpurun_out/r3e_msd_fuzz.log 2>&1; rc=$?; tail -n 2 gpurun_out/r3e_msd_fuzz.log
if [ $rc -ne 0 ]; then exit $rc; fi
for v in head nohead head nohead; do
  if [ $v = nohead ]; then export MDX_MSD_NO_HEAD=1; else unset MDX_MSD_NO_HEAD; fi
  timeout -k 10 200 python bench.py --workload msd --steps 20 --no-cpu-baseline > gpurun_out/r3e_msd_$v.json 2>gpurun_out/r3e_msd_$v.err || exit $?
  python -c "
import json
d=json.load(open('gpurun_out/r3e_msd_$v.json')); print('$v', round(d['ms_per_step'],3),'ms/step', d['roofline']['kernel_ms_per_step']/20, d['result_digest'][:3])"
done
