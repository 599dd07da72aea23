#!/bin/bash
out=gpurun_out/r5c; mkdir -p $out
timeout -k 10 400 python scripts/diag/mmap_feed_rates.py > $out/mmap_feed_rates.json 2> $out/mmap.err; echo "rc=$?"; cat $out/mmap_feed_rates.json; tail -3 $out/mmap.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$out/trace_b1 -o b1 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload msd --steps 3 --warmup 3 --no-onsager --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/trace_b1.json 2> $GRAFT_REPO_ROOT/$out/trace_b1.err; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
fs = glob.glob("gpurun_out/r5c/trace_b1/**/*kernel_trace.csv", recursive=True)
print(fs)
rows = list(csv.DictReader(open(fs[0])))
rows = [r for r in rows if "msd_fft" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[-8]["Start_Timestamp"])
for r in rows[-8:]:
    print(r["Kernel_Name"][:60], "stream", r.get("Stream_Id"), "start %.3f end %.3f ms" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6))
PY
