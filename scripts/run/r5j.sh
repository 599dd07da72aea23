#!/bin/bash
out=gpurun_out/r5j; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "msd or onsager or Onsager or correl or cross or runtime or c4" > $out/pytest_msd.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_msd.log
timeout -k 10 300 python bench.py --workload msd --steps 10 --warmup 6 --no-cpu-baseline --no-onsager > $out/msd.json 2> $out/msd.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5j/msd.json"))
print("ms/step", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms_per_step"], "result_ms", d["result_ms"], "frac", d["roofline"]["frac"])
PY
timeout -k 10 200 python scripts/msd_fuzz.py 60 > $out/msd_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 $out/msd_fuzz.log
