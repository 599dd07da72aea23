#!/bin/bash
# twiddles as powers in registers everywhere (this tree) against the LDS tables (scripts/diag/_baseline_msd_fft.hpp.txt): one box
out=gpurun_out/r5s; mkdir -p $out
lines() { tag=$1
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --workload msd --steps 12 --warmup 6 --no-onsager --no-cpu-baseline > $out/b1_${tag}_$rep.json 2>> $out/err.log
    timeout -k 10 200 python bench.py --workload msd --blocks 8 --steps 12 --warmup 6 --no-onsager --no-cpu-baseline > $out/b8_${tag}_$rep.json 2>> $out/err.log
    timeout -k 10 200 python bench.py --workload msd --blocks 250 --steps 12 --warmup 6 --no-onsager --no-cpu-baseline > $out/b250_${tag}_$rep.json 2>> $out/err.log
  done
  timeout -k 10 200 python bench.py --workload msd --atoms 5000 --frames 200000 --steps 6 --warmup 4 --no-onsager --no-cpu-baseline > $out/n409600_${tag}.json 2>> $out/err.log
  timeout -k 10 200 python bench.py --workload msd --atoms 30000 --frames 32768 --steps 6 --warmup 4 --no-onsager --no-cpu-baseline > $out/n65536_${tag}.json 2>> $out/err.log
  timeout -k 10 200 python bench.py --workload msd --atoms 2000 --frames 500000 --steps 6 --warmup 4 --no-onsager --no-cpu-baseline > $out/n1048576_${tag}.json 2>> $out/err.log
  timeout -k 10 200 python bench.py --workload msd --atoms 7629 --frames 131072 --steps 6 --warmup 4 --no-onsager --no-cpu-baseline > $out/n262144_${tag}.json 2>> $out/err.log
}
timeout -k 10 600 python -m pytest tests/test_gpu_engines.py tests/test_gpu_analysis.py -m gpu -x -q -k "msd or transform or onsager or Onsager" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log | cut -c1-200
timeout -k 10 120 python scripts/msd_fuzz.py 80 17 > $out/fuzz.log 2>&1; tail -1 $out/fuzz.log
lines gen
cp mdhelper_amd/csrc/mdx_msd_fft.hpp $out/new.hpp
cp scripts/diag/_baseline_msd_fft.hpp.txt mdhelper_amd/csrc/mdx_msd_fft.hpp
make -C mdhelper_amd/csrc > $out/make_tab.log 2>&1; echo "make rc=$?"
lines tab
cp $out/new.hpp mdhelper_amd/csrc/mdx_msd_fft.hpp; rm $out/new.hpp
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5s/*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], "ms/step %.2f kernel %.2f" % (d["ms_per_step"], r["kernel_ms_per_step"]))
PY
