#!/bin/bash
# round 3, final evidence pass: parity suite, counters of the final sources, the bench lines, fuzz
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/final
export TMPDIR=/tmp
O=gpurun_out/final
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
rc=$?; tail -3 $O/pytest_gpu.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 1150 python scripts/make_counters.py > $O/make_counters.log 2>&1 || { tail -20 $O/make_counters.log; exit 1; }
cp gpurun_out/counters/counters.json profiles/counters.json
echo counters done
timeout -k 10 300 python bench.py > $O/bench_final.json 2> $O/bench_final.err || { tail -5 $O/bench_final.err; exit 1; }
timeout -k 10 200 python bench.py --workload msd --steps 20 --no-cpu-baseline > $O/bench_msd_20steps.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --workload msd --blocks 8 --steps 10 --no-cpu-baseline > $O/bench_msd8.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --workload isf > $O/bench_isf.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --workload sq --steps 10 --warmup 2 > $O/bench_sq_c3.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --workload sq --n-points 32 --frames 200 --steps 3 > $O/bench_sq_default_grid.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --atoms 131072 --frames 1000 --steps 2 --no-extras > $O/bench_c5size.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --workload rdf_wide --atoms 1000 --frames 20000 --steps 3 --no-extras > $O/bench_c1like.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --gpus 2 --share-devices --shard-fixed --frames 64 --atoms 8192 --no-cpu-baseline --no-extras > $O/bench_2ranks_shared.json 2>/dev/null || exit 1
timeout -k 10 300 python scripts/rdf_fuzz.py 100 41 > $O/fuzz_rdf.log 2>&1 || { tail -3 $O/fuzz_rdf.log; exit 1; }
timeout -k 10 200 python scripts/sq_fuzz.py 50 41 > $O/fuzz_sq.log 2>&1 || { tail -3 $O/fuzz_sq.log; exit 1; }
timeout -k 10 200 python scripts/msd_fuzz.py 50 41 > $O/fuzz_msd.log 2>&1 || { tail -3 $O/fuzz_msd.log; exit 1; }
tail -1 $O/fuzz_rdf.log $O/fuzz_sq.log $O/fuzz_msd.log
python - <<'PY'
import json
O="gpurun_out/final/"
d=json.load(open(O+"bench_final.json"))
print("C2(i)", round(d["frames_per_sec"]), "frames/s", d["value"], "frac", d["roofline"]["frac"], d["roofline"].get("frac_evaluations"), d["roofline"].get("frac_binned"), "traffic", d["roofline"]["traffic"], "alg", d["roofline"]["hbm"]["algorithmic_bytes_per_launch"])
for k,v in d["extra"].items():
    if k=="ingest": print({a:(round(b["frames_per_sec"]),round(b["ratio_to_resident"],3)) for a,b in v.items() if isinstance(b,dict) and "frames_per_sec" in b})
    else: print(k, v.get("frames_per_sec"), v.get("ms_per_step"), v.get("roofline",{}).get("frac"), v.get("error"))
for n in ("bench_msd_20steps","bench_msd8","bench_isf","bench_sq_default_grid","bench_c5size","bench_c1like","bench_2ranks_shared"):
    e=json.load(open(O+n+".json")); print(n, e.get("frames_per_sec"), e.get("ms_per_step"), e.get("roofline",{}).get("frac"))
PY
