#!/bin/bash
# S(q) / ISF kernels after a change: their parity tests, the fuzz, the bench lines (with the general form beside)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "sq or structure_factor or isf or fourier or c3 or Intermediate or smoke" > gpurun_out/r3d_pytest.log 2>&1
rc=$?; tail -4 gpurun_out/r3d_pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python scripts/sq_fuzz.py 40 5 > gpurun_out/r3d_sq_fuzz.log 2>&1; rc=$?; tail -2 gpurun_out/r3d_sq_fuzz.log
if [ $rc -ne 0 ]; then exit $rc; fi
for v in regular general; do
  if [ $v = general ]; then export MDX_SQ_NO_REGULAR=1; else unset MDX_SQ_NO_REGULAR; fi
  timeout -k 10 200 python bench.py --workload sq --steps 5 --no-cpu-baseline > gpurun_out/r3d_sq_$v.json 2>gpurun_out/r3d_sq_$v.err || exit $?
  timeout -k 10 200 python bench.py --workload sq --n-points 32 --frames 200 --steps 3 --no-cpu-baseline > gpurun_out/r3d_sq32_$v.json 2>gpurun_out/r3d_sq32_$v.err || exit $?
  timeout -k 10 200 python bench.py --workload isf --steps 3 --no-cpu-baseline > gpurun_out/r3d_isf_$v.json 2>gpurun_out/r3d_isf_$v.err || exit $?
  python -c "
import json
for n in ('sq','sq32','isf'):
    d=json.load(open('gpurun_out/r3d_%s_$v.json'%n)); print('$v',n,round(d['frames_per_sec'],1),'frames/s', round(d['ms_per_step'],3),'ms/step')"
done
