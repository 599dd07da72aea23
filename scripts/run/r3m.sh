#!/bin/bash
# RDF: kernel durations with the gather form of the sort (default) and with the scatter form (MDX_RDF_SORT_SCATTER=1)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3m
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "rdf or radial or c2 or c5 or beyond or smoke" > gpurun_out/r3m/pytest.log 2>&1
rc=$?; tail -n 2 gpurun_out/r3m/pytest.log; if [ $rc -ne 0 ]; then exit $rc; fi
for mode in gather scatter; do
  if [ $mode = scatter ]; then export MDX_RDF_SORT_SCATTER=1; else unset MDX_RDF_SORT_SCATTER; fi
  timeout -k 10 300 python bench.py --frames 4000 --steps 3 --no-cpu-baseline --no-extras > gpurun_out/r3m/bench_$mode.json 2>/dev/null || exit 1
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3m/prof_$mode -o p -- python3 $GRAFT_REPO_ROOT/bench.py --frames 4000 --steps 2 --no-cpu-baseline --no-extras > /dev/null 2>&1) || exit 1
done
unset MDX_RDF_SORT_SCATTER
timeout -k 10 300 python scripts/rdf_fuzz.py 60 77 > gpurun_out/r3m/fuzz.log 2>&1; rc=$?; tail -n 1 gpurun_out/r3m/fuzz.log; if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r3m/bench_default_10000.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --atoms 131072 --frames 1000 --steps 2 --no-extras --no-cpu-baseline > gpurun_out/r3m/c5.json 2>/dev/null || exit 1
python - <<'PY'
import json,glob,csv
for mode in ("gather","scatter"):
    d=json.load(open(f"gpurun_out/r3m/bench_{mode}.json"))
    print(mode, round(d["frames_per_sec"]), d["ms_per_step"])
    for f in glob.glob(f"gpurun_out/r3m/prof_{mode}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rdf_" in r["Name"]: print("   ", r["Name"][:40], r["Calls"], r["AverageNs"], r["Percentage"])
print("default 10000 frames", round(json.load(open("gpurun_out/r3m/bench_default_10000.json"))["frames_per_sec"]), "C5", round(json.load(open("gpurun_out/r3m/c5.json"))["frames_per_sec"]))
PY
