#!/bin/bash
# ISF and MSD: kernel time against wall time of a step (launch-level gaps)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3p
export TMPDIR=/tmp
for w in isf msd; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3p/prof_$w -o p -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3p/bench_$w.json 2>/dev/null) || exit 1
done
python - <<'PY'
import json,glob,csv
for w in ("isf","msd"):
    d=json.loads(open(f"gpurun_out/r3p/bench_{w}.json").read().strip().splitlines()[-1])
    print(w, "ms_per_step", d["ms_per_step"], "steps", d["steps"], "warmup", d["warmup"])
    tot=0
    for f in glob.glob(f"gpurun_out/r3p/prof_{w}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            print("   ", r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
            if "synth" not in r["Name"]: tot+=float(r["TotalDurationNs"])
    print("   kernel total per step (ms):", tot/1e6/(d["steps"]+d["warmup"]))
PY
