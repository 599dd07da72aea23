#!/bin/bash
# Round 5: pass B on a second stream (A/B against MDX_MSD_NO_OVERLAP=1, alternating on one box), the MSD / Onsager
# tests on it, host-feed rates by copy-thread count.
out=gpurun_out/r5b; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "msd or onsager or Onsager or correl or cross or runtime" > $out/pytest_msd.log 2>&1; echo "pytest rc=$?"
tail -3 $out/pytest_msd.log
ab() { name=$1; shift
  for rep in 1 2; do
    MDX_MSD_NO_OVERLAP=1 timeout -k 10 200 python bench.py --workload msd --steps 12 --warmup 6 --no-onsager --no-cpu-baseline "$@" > $out/${name}_serial_$rep.json 2>> $out/err.log
    timeout -k 10 200 python bench.py --workload msd --steps 12 --warmup 6 --no-onsager --no-cpu-baseline "$@" > $out/${name}_overlap_$rep.json 2>> $out/err.log
  done
}
ab b1
ab b8 --blocks 8
ab b2 --blocks 2
ab b250 --blocks 250
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5b/b*.json")):
    try:
        d = json.load(open(f)); r = d.get("roofline", {})
        print(f.split("/")[-1], "ms/step %.2f" % d["ms_per_step"], "kernel ms", r.get("kernel_ms_per_step"), "frac", r.get("frac"), "median", r.get("frac_median_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
timeout -k 10 300 python scripts/diag/pageable_vs_ring.py > $out/pageable_vs_ring.json 2>> $out/err.log; cat $out/pageable_vs_ring.json
timeout -k 10 500 python scripts/diag/host_feed_rates.py 8 12 16 > $out/host_feed_rates.txt 2>> $out/err.log; cat $out/host_feed_rates.txt
