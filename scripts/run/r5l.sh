#!/bin/bash
# kernel split of the shapes whose pass B has 1024-point rows (409 600 = 400 x 1024, 2^16 = 64 x 1024, 2^20 = 1024 x 1024)
out=$GRAFT_REPO_ROOT/gpurun_out/r5l; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/$name -o $name --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload msd --steps 4 --warmup 3 --no-onsager --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$?"
}
run n409600 --atoms 5000 --frames 200000
run n65536 --atoms 30000 --frames 32768
run n1048576 --atoms 2000 --frames 500000
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json
for name in ("n409600", "n65536", "n1048576"):
    try:
        d = json.load(open(f"gpurun_out/r5l/{name}.json"))
        print(name, d["config"]["workload"], "ms/step %.2f kernel %.2f" % (d["ms_per_step"], d["roofline"]["kernel_ms_per_step"]))
        f = glob.glob(f"gpurun_out/r5l/{name}/**/*kernel_stats.csv", recursive=True)[0]
        for r in list(csv.DictReader(open(f)))[:5]:
            print("   ", r["Name"][:70], r["Calls"], "avg %.3f ms" % (float(r["AverageNs"]) / 1e6), r["Percentage"])
    except Exception as e:
        print(name, "ERR", e)
PY
