# C4 size at other block counts: kernels per step (MDX_LIBRARY may point at another build for an A/B on one box).
import sys, os, json, numpy as np
sys.path.insert(0, os.getcwd())
import bench
for b in [int(x) for x in sys.argv[1:]] or (1, 2, 3, 4, 6, 8, 12, 16, 20, 32, 40):
    args = bench.parse(["--workload", "msd", "--blocks", str(b), "--steps", "12", "--warmup", "4", "--no-onsager", "--no-cpu-baseline"])
    world = bench.World(args)
    d = bench.bench_msd(args, world)
    print(os.environ.get("MDX_LIBRARY", "product"), "blocks", b, d["config"]["workload"].split("n_fft=")[1],
          "median kernel ms", round(float(np.median(d["roofline"]["kernel_ms_each_step"])), 2), flush=True)
