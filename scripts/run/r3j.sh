#!/bin/bash
# S(q) / ISF regular form with coordinates fetched a tile ahead: parity, then A/B against the general fill
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "sq or structure or isf or scatter or ssf" > gpurun_out/r3j_pytest.log 2>&1
rc=$?; tail -n 3 gpurun_out/r3j_pytest.log; if [ $rc -ne 0 ]; then exit $rc; fi
for i in 1 2; do
  timeout -k 10 300 python bench.py --workload sq --steps 10 --warmup 2 > gpurun_out/r3j_sq_new_$i.json 2> gpurun_out/r3j_sq_new_$i.err || exit 1
  MDX_SQ_NO_REGULAR=1 timeout -k 10 300 python bench.py --workload sq --steps 10 --warmup 2 > gpurun_out/r3j_sq_general_$i.json 2> gpurun_out/r3j_sq_general_$i.err || exit 1
done
timeout -k 10 300 python bench.py --workload sq --n-points 32 --steps 3 --warmup 1 > gpurun_out/r3j_sq32.json 2> gpurun_out/r3j_sq32.err || exit 1
timeout -k 10 300 python bench.py --workload isf --steps 3 --warmup 1 > gpurun_out/r3j_isf.json 2> gpurun_out/r3j_isf.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3j_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d.get('value'), d.get('ms_per_step'), (d.get('extra') or {}).get('frames_per_sec'), d.get('roofline',{}).get('frac'))
PY
