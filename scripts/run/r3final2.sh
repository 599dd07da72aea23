#!/bin/bash
# round 3, evidence pass after the S(q) / ISF changes: parity suite, S(q) counters and kernel statistics of the final
# sources (the RDF / MSD entries of profiles/counters.json stay: their sources did not change), bench lines, fuzz
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/final2
export TMPDIR=/tmp
O=gpurun_out/final2
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
rc=$?; tail -3 $O/pytest_gpu.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 1100 python scripts/make_counters.py sq_c3 stats > $O/make_counters.log 2>&1 || { tail -20 $O/make_counters.log; exit 1; }
cp gpurun_out/counters/counters.json profiles/counters.json
echo counters done
timeout -k 10 300 python bench.py > $O/bench_final.json 2> $O/bench_final.err || { tail -5 $O/bench_final.err; exit 1; }
timeout -k 10 200 python bench.py --workload sq --steps 10 --warmup 2 > $O/bench_sq_c3.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --workload isf > $O/bench_isf.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --workload sq --n-points 32 --frames 200 --steps 3 > $O/bench_sq_default_grid.json 2>/dev/null || exit 1
timeout -k 10 300 python scripts/run/diag_sq_forms.py > $O/sq_forms.txt 2>&1 || { tail -5 $O/sq_forms.txt; exit 1; }
timeout -k 10 300 python scripts/sq_fuzz.py 120 43 > $O/fuzz_sq.log 2>&1 || { tail -3 $O/fuzz_sq.log; exit 1; }
tail -1 $O/fuzz_sq.log
cat $O/sq_forms.txt
python - <<'PY'
import json
O="gpurun_out/final2/"
d=json.load(open(O+"bench_final.json"))
print("C2(i)", round(d["frames_per_sec"]), "frames/s", d["value"], "frac", d["roofline"]["frac"], d["roofline"].get("frac_evaluations"), d["roofline"].get("frac_binned"), "traffic", d["roofline"]["traffic"])
for k,v in d["extra"].items():
    if k=="ingest": print({a:(round(b["frames_per_sec"]),round(b["ratio_to_resident"],3)) for a,b in v.items() if isinstance(b,dict) and "frames_per_sec" in b})
    else: print(k, v.get("frames_per_sec"), v.get("ms_per_step"), v.get("roofline",{}).get("frac"), v.get("error"))
for n in ("bench_sq_c3","bench_isf","bench_sq_default_grid"):
    e=json.load(open(O+n+".json")); print(n, e.get("frames_per_sec"), e.get("value"), e.get("ms_per_step"), e.get("roofline",{}).get("frac"), e.get("roofline",{}).get("valu"))
print(json.dumps(json.load(open("profiles/counters.json"))["sq_c3"], indent=1))
PY
