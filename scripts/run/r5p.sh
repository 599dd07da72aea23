#!/bin/bash
out=gpurun_out/r5p; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "msd or onsager or Onsager or correl or cross or EndToEnd or polymer" > $out/pytest_msd.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest_msd.log | cut -c1-200
timeout -k 10 150 python scripts/msd_fuzz.py 90 7 > $out/msd_fuzz.log 2>&1; tail -1 $out/msd_fuzz.log
for cfg in "2000 500000" "3814 262144" "7629 131072"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --workload msd --atoms $1 --frames $2 --steps 6 --warmup 4 --no-onsager --no-cpu-baseline > $out/msd_$1x$2.json 2>> $out/err.log
  python - <<PY
import json
d = json.load(open("gpurun_out/r5p/msd_$1x$2.json"))
print(d["config"]["workload"][:110], "| ms/step %.2f kernel %.2f" % (d["ms_per_step"], d["roofline"]["kernel_ms_per_step"]))
PY
done
MDX_MSD_ROCFFT=1 timeout -k 10 200 python bench.py --workload msd --atoms 2000 --frames 100000 --steps 4 --warmup 3 --no-onsager --no-cpu-baseline > $out/msd_rocfft.json 2>> $out/err.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5p/msd_rocfft.json"))
print(d["config"]["workload"][:110], "| ms/step %.2f kernel %.2f" % (d["ms_per_step"], d["roofline"]["kernel_ms_per_step"]))
PY
