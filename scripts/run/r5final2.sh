#!/bin/bash
# Round-5 final pass, part 2: the bench lines for profiles/ (after profiles/counters.json holds the final MSD entries).
out=gpurun_out/r5final; mkdir -p $out
run() { name=$1; shift; t0=$SECONDS; timeout -k 10 500 python bench.py "$@" > $out/$name.json 2> $out/$name.err; echo "$name: rc=$? $((SECONDS - t0)) s wall"; }
run bench_final
run bench_c5size --atoms 131072 --frames 500 --steps 3 --no-extras --cpu-seconds 3
run bench_c1like --workload rdf_wide --atoms 1000 --frames 20000 --steps 3 --no-extras --cpu-seconds 3
run bench_sq_c3 --workload sq --steps 10 --warmup 2
run bench_isf --workload isf --steps 3 --warmup 1
run bench_msd_20steps --workload msd --steps 20 --warmup 6
run bench_msd8 --workload msd --blocks 8 --steps 20 --warmup 6
for b in 2 4 16 32 64 250; do run bench_msd$b --workload msd --blocks $b --steps 12 --warmup 6 --no-onsager --no-cpu-baseline; done
run bench_msd_409600 --workload msd --atoms 5000 --frames 200000 --steps 8 --warmup 4 --no-onsager --no-cpu-baseline
run bench_msd_65536 --workload msd --atoms 30000 --frames 32768 --steps 8 --warmup 4 --no-onsager --no-cpu-baseline
run bench_msd_1048576 --workload msd --atoms 2000 --frames 500000 --steps 8 --warmup 4 --no-onsager --no-cpu-baseline
run bench_2ranks_shared --gpus 2 --share-devices --shard-fixed --frames 2000 --steps 2 --no-cpu-baseline
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5final/bench_*.json")):
    try:
        d = json.load(open(f))
        r = d.get("roofline", {})
        print(f.split("/")[-1], d.get("metric"), "%.4g" % d.get("value", 0), "ms/step %.2f" % d.get("ms_per_step", 0), "frac", r.get("frac"), "traffic", r.get("traffic"))
    except Exception as e:
        print(f, "ERR", e)
PY
