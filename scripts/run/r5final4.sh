#!/bin/bash
# Round-5 final pass, part 4: after the last edit of the MSD sources — its tests, its counters, its bench lines.
out=gpurun_out/r5final; mkdir -p $out gpurun_out/counters
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "msd or onsager or Onsager or correl or cross or EndToEnd or end_to_end or polymer or acf" > $out/pytest_msd_last.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest_msd_last.log | cut -c1-200
MDX_ROUND=r05 timeout -k 10 600 python scripts/make_counters.py msd_c4 msd_c4_b8 stats_msd > gpurun_out/counters/make_counters_5.log 2>&1; echo "counters rc=$?"
cp gpurun_out/counters/counters.json profiles/counters.json
run() { name=$1; shift; timeout -k 10 500 python bench.py "$@" > $out/$name.json 2> $out/$name.err; echo "$name: rc=$?"; }
run bench_final
run bench_msd_20steps --workload msd --steps 20 --warmup 6
run bench_msd8 --workload msd --blocks 8 --steps 20 --warmup 6
for b in 2 4 16 32 64 250; do run bench_msd$b --workload msd --blocks $b --steps 12 --warmup 6 --no-onsager --no-cpu-baseline; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5final/bench_*.json")):
    try:
        d = json.load(open(f)); r = d.get("roofline", {})
        print(f.split("/")[-1], "%.4g" % d.get("value", 0), "ms/step %.2f" % d.get("ms_per_step", 0), "frac", r.get("frac"), "traffic", r.get("traffic"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -c 1400 $out/bench_final.json
