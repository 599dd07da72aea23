#!/bin/bash
out=gpurun_out/r5g; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout -k 10 400 python bench.py --workload msd --steps 5 --warmup 5 --no-cpu-baseline > $out/msd.json 2> $out/msd.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5g/msd.json"))
o = d["onsager"]
for k in ("class_hbm_f64", "class_hbm_f32", "class_host_f32", "class_host_f32_pinned", "class_file"):
    v = o[k]
    print(k, "%.1f ms" % v["ms_per_analysis"], [round(x, 1) for x in v["ms_each"]], "first %.1f" % v["first_analysis_ms"], "link", v.get("link_bound_ms"), {a: round(b, 1) for a, b in v["phases_ms"].items()}, "dev", v.get("max_rel_deviation_from_hbm_f64"))
print("h2d", o["h2d_page_locked_GB_per_sec"], o["h2d_pageable_ring_GB_per_sec"])
PY
