#!/bin/bash
# DIF stage twiddles as powers in registers (DIF_GENERATED_TWIDDLES 1) against the LDS table (0): same box, rebuilt in place
out=gpurun_out/r5r; mkdir -p $out
lines() { tag=$1
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --workload msd --steps 12 --warmup 6 --no-onsager --no-cpu-baseline > $out/b1_${tag}_$rep.json 2>> $out/err.log
    timeout -k 10 200 python bench.py --workload msd --blocks 8 --steps 12 --warmup 6 --no-onsager --no-cpu-baseline > $out/b8_${tag}_$rep.json 2>> $out/err.log
    timeout -k 10 200 python bench.py --workload msd --blocks 250 --steps 12 --warmup 6 --no-onsager --no-cpu-baseline > $out/b250_${tag}_$rep.json 2>> $out/err.log
  done
}
lines gen
sed -i 's/#define DIF_GENERATED_TWIDDLES 1/#define DIF_GENERATED_TWIDDLES 0/' mdhelper_amd/csrc/mdx_msd_fft.hpp
make -C mdhelper_amd/csrc > $out/make_tab.log 2>&1; echo "make rc=$?"
lines tab
sed -i 's/#define DIF_GENERATED_TWIDDLES 0/#define DIF_GENERATED_TWIDDLES 1/' mdhelper_amd/csrc/mdx_msd_fft.hpp
make -C mdhelper_amd/csrc > $out/make_gen.log 2>&1; echo "make rc=$?"
lines gen2
timeout -k 10 400 python -m pytest tests/test_gpu_engines.py -m gpu -x -q -k "msd or transform" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log | cut -c1-200
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5r/b*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], "ms/step %.2f kernel %.2f frac %.4f" % (d["ms_per_step"], r["kernel_ms_per_step"], r["frac"]))
PY
