#!/bin/bash
out=gpurun_out/r5v; mkdir -p $out
lines() { tag=$1
  for cfg in "2000 500000" "3814 262144" "7629 131072"; do
    set -- $cfg
    for rep in 1 2; do
      timeout -k 10 200 python bench.py --workload msd --atoms $1 --frames $2 --steps 6 --warmup 4 --no-onsager --no-cpu-baseline > $out/n$2_${tag}_$rep.json 2>> $out/err.log
    done
  done
}
timeout -k 10 600 python -m pytest tests/test_gpu_engines.py -m gpu -x -q -k "msd or transform" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $out/pytest.log | cut -c1-200
timeout -k 10 150 python scripts/msd_fuzz.py 120 23 > $out/fuzz.log 2>&1; tail -1 $out/fuzz.log
lines new
cp mdhelper_amd/csrc/mdx_msd_fft.hpp $out/new.hpp
cp scripts/diag/_baseline_msd_fft.hpp.txt mdhelper_amd/csrc/mdx_msd_fft.hpp
make -C mdhelper_amd/csrc > $out/make_old.log 2>&1; echo "make rc=$?"
lines old
cp $out/new.hpp mdhelper_amd/csrc/mdx_msd_fft.hpp; rm $out/new.hpp
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5v/n*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], d["config"]["workload"].split("n_fft=")[1][:44], "kernel %.2f" % r["kernel_ms_per_step"])
PY
