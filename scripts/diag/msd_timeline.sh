#!/bin/bash
# Timeline of one MSD step (C4, one block) from rocprofv3 --kernel-trace: where do the milliseconds between the
# kernels' own time (26 ms) and the step (28 - 31 ms) go?
out=$GRAFT_REPO_ROOT/gpurun_out/msd_timeline; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o tl -- python3 $GRAFT_REPO_ROOT/bench.py --workload msd --steps 6 --warmup 2 --no-onsager --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python3 - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for p in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
for p in glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", "")))
rows.sort()
# the last two steps: print every event with the gap before it
last = [i for i, r in enumerate(rows) if "cols400" in r[2]]
i0 = last[-4]
prev_end = rows[i0 - 1][1] if i0 else rows[i0][0]
for s, e, name in rows[i0 - 6:]:
    print(f"gap {max(0, s - prev_end) / 1e3:9.1f} us   dur {(e - s) / 1e3:9.1f} us   {name}")
    prev_end = max(prev_end, e)
PY
