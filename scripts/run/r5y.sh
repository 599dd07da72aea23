#!/bin/bash
out=gpurun_out/r5y; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "onsager or Onsager or c4" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $out/pytest.log | cut -c1-200
timeout -k 10 400 python bench.py --workload msd --steps 5 --warmup 5 --no-cpu-baseline > $out/msd.json 2> $out/msd.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5y/msd.json"))
o = d["onsager"]
for k in ("class_hbm_f64", "class_hbm_f32", "class_host_f32", "class_host_f32_pinned", "class_file"):
    v = o[k]
    print(k, "%.1f ms" % v["ms_per_analysis"], [round(x, 1) for x in v["ms_each"]], "link", v.get("link_bound_ms"), "dev", v.get("max_rel_deviation_from_hbm_f64"))
PY
