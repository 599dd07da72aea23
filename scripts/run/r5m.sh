#!/bin/bash
out=gpurun_out/r5m; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_engines.py tests/test_gpu_analysis.py -m gpu -x -q -k "msd or onsager or Onsager or transform" > $out/pytest_msd.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_msd.log | cut -c1-200
for cfg in "5000 200000" "30000 32768" "2000 500000" "7142 140000"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --workload msd --atoms $1 --frames $2 --steps 6 --warmup 4 --no-onsager --no-cpu-baseline > $out/msd_$1x$2.json 2>> $out/err.log
  python - <<PY
import json
d = json.load(open("gpurun_out/r5m/msd_$1x$2.json"))
print(d["config"]["workload"][:110], "| ms/step %.2f kernel %.2f" % (d["ms_per_step"], d["roofline"]["kernel_ms_per_step"]))
PY
done
