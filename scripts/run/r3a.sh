#!/bin/bash
# round 3, first GPU pass: full parity suite, default bench (with the ingest legs), fuzz, ingest thread sweep
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r3a_pytest.log 2>&1
rc=$?; tail -5 gpurun_out/r3a_pytest.log; echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python bench.py > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err
rc=$?; echo "bench rc=$rc"; tail -3 gpurun_out/r3a_bench.err
if [ $rc -ge 124 ]; then exit $rc; fi
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3a_bench.json"))
print("C2(i) frames/s", d["frames_per_sec"], "value", d["value"], "ms/step", d["ms_per_step"])
print("roofline", {k:d["roofline"].get(k) for k in ("frac","frac_evaluations","frac_binned","kernel_ms_per_launch")})
print("cpu_c", d.get("cpu_baseline_c"))
for k,v in d.get("extra",{}).items():
    if k=="ingest": print("ingest", json.dumps(v)[:3000])
    else: print(k, v.get("frames_per_sec"), v.get("ms_per_step"), v.get("error"))
PY
timeout -k 10 200 python scripts/rdf_fuzz.py 60 31 > gpurun_out/r3a_fuzz.log 2>&1; rc=$?; tail -2 gpurun_out/r3a_fuzz.log
if [ $rc -ge 124 ]; then exit $rc; fi
for t in 4 16; do
  MDX_IO_THREADS=$t timeout -k 10 200 python bench.py --workload ingest --steps 2 > gpurun_out/r3a_ingest_t$t.json 2> gpurun_out/r3a_ingest_t$t.err || exit $?
  python -c "
import json;d=json.load(open('gpurun_out/r3a_ingest_t$t.json'))['extra']['ingest']
print('threads $t', {k:(round(v['frames_per_sec']),round(v['ratio_to_resident'],3)) for k,v in d.items() if isinstance(v,dict) and 'frames_per_sec' in v}, d.get('error'))"
done
rocprofv3 -L > gpurun_out/r3a_counters_avail.txt 2>&1 || true
grep -c . gpurun_out/r3a_counters_avail.txt
