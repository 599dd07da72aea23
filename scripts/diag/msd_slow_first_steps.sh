#!/bin/bash
# Which host calls run while a pass A launch of the first steps takes 4 x its time?  (kernel + HIP API trace, no counters)
out=$PWD/gpurun_out/msdtrace; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --hip-trace --output-format csv -d $out -o t -- python3 $GRAFT_REPO_ROOT/bench.py --workload msd --steps 8 --warmup 0 --no-cpu-baseline --no-onsager > $out/bench.json 2> $out/bench.err
python3 - <<'PY'
import csv, glob, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/msdtrace"
k = [r for r in csv.DictReader(open(glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]))]
api = [r for r in csv.DictReader(open(glob.glob(out + "/**/*hip_api_trace.csv", recursive=True)[0]))]
t0 = float(k[0]["Start_Timestamp"])
for r in k:
    if "cols400" not in r["Kernel_Name"]:
        continue
    s, e = float(r["Start_Timestamp"]), float(r["End_Timestamp"])
    line = "passA start %8.1f ms dur %6.2f ms" % ((s - t0) / 1e6, (e - s) / 1e6)
    if e - s > 15e6:
        calls = [(a["Function"], (float(a["Start_Timestamp"]) - t0) / 1e6, (float(a["End_Timestamp"]) - float(a["Start_Timestamp"])) / 1e6)
                 for a in api if float(a["End_Timestamp"]) > s and float(a["Start_Timestamp"]) < e]
        calls = [c for c in calls if c[2] > 0.05]
        line += "  host calls > 50 us meanwhile: " + "; ".join("%s @%.1f %.2f ms" % c for c in calls[:12])
    print(line)
PY
