#!/bin/bash
# Run on the GPU box (through gpurun): one rocprofv3 PMC pass (counters + kernel trace only)
# over a bench.py invocation; prints per-kernel sums of each counter.
# usage: scripts/profile_counters.sh <tag> "<CTR1 CTR2 ...>" <bench args...>
set -o pipefail
tag=$1; ctrs=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/ctr_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$out" -o "$tag" -- python3 "$root/bench.py" "$@" > "$out/bench.json" 2> "$out/bench.err" || { tail -5 "$out/bench.err"; exit 1; }
python3 - "$out" "$tag" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: collections.defaultdict(float))
for path in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            res[row.get("Kernel_Name", "")[:60]][row["Counter_Name"]] += float(row["Counter_Value"])
with open(f"{out}/{tag}_counters.csv", "w") as fh:
    for name, d in res.items():
        for k, v in sorted(d.items()):
            fh.write(f"\"{name}\",{k},{v:.0f}\n")
print(open(f"{out}/{tag}_counters.csv").read())
PY
find "$out" -name "*counter_collection.csv" -size +2M -delete 2>/dev/null
find "$out" -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
exit 0
