#!/bin/bash
# Does the k1 row stride of the half-transformed block (a multiple of 64 KB: the 400 lines an iteration of pass A
# stores fall on few memory channels) hold pass A back?  MDX_MSD_Y_PAD adds complex values to the stride.
mkdir -p gpurun_out/ypad
for pad in 0 8 16 72 264 520 4104 0 264; do
  MDX_MSD_Y_PAD=$pad python bench.py --workload msd --steps 16 --warmup 3 --no-onsager --no-cpu-baseline > gpurun_out/ypad/p$pad.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/ypad/p$pad.json')); print('pad $pad (', $pad*16, 'B ): kernel ms/step', round(d['roofline']['kernel_ms_per_step'],2), 'step', round(d['ms_per_step'],2), d['result_digest'][:2])"
done
for pad in 0 264; do
  MDX_MSD_Y_PAD=$pad python bench.py --workload msd --blocks 8 --steps 16 --warmup 3 --no-onsager --no-cpu-baseline > gpurun_out/ypad/b8_p$pad.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/ypad/b8_p$pad.json')); print('8 blocks pad $pad: kernel ms/step', round(d['roofline']['kernel_ms_per_step'],2), 'step', round(d['ms_per_step'],2))"
done
