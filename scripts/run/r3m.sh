#!/bin/bash
# RDF: the sort kernel's standalone cost (one stream: MDX_RDF_NO_OVERLAP=1) against the overlapped default
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3m
export TMPDIR=/tmp
for mode in overlap serial; do
  if [ $mode = serial ]; then export MDX_RDF_NO_OVERLAP=1; else unset MDX_RDF_NO_OVERLAP; fi
  timeout -k 10 300 python bench.py --frames 4000 --steps 3 --no-cpu-baseline --no-extras > gpurun_out/r3m/bench_$mode.json 2>/dev/null || exit 1
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3m/prof_$mode -o p -- python3 $GRAFT_REPO_ROOT/bench.py --frames 4000 --steps 2 --no-cpu-baseline --no-extras > /dev/null 2>&1) || exit 1
done
python - <<'PY'
import json,glob,csv
for mode in ("overlap","serial"):
    d=json.load(open(f"gpurun_out/r3m/bench_{mode}.json"))
    print(mode, round(d["frames_per_sec"]), d["ms_per_step"])
    for f in glob.glob(f"gpurun_out/r3m/prof_{mode}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rdf_" in r["Name"]: print("   ", r["Name"][:40], r["Calls"], r["AverageNs"], r["Percentage"])
PY
