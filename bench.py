#!/usr/bin/env python3
"""
bench.py — headline benchmark of the MI355X-native mdhelper hot path.

    python bench.py --gpus N --steps K --warmup W [--workload rdf|rdf_wide|sq|msd|isf]

Default workload (BASELINE.json configs[1], "C2"): RDF on 32 768 atoms, cubic
box L = 68.94 A (rho = 0.1 A^-3), n_bins = 201, range = (0, 15) A, self RDF with
exclusion = (1, 1); synthetic wrapped Gaussian random walk generated in HBM.
One *step* = one pass of the hot path (mdx_rdf_accumulate_device) over one batch
of `--frames` frames already resident in HBM.  For N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank owns its own batch of
frames (weak scaling, frames shard with no data-path collective) and the
per-rank histograms meet in ONE RCCL all-reduce at the end of the timed region.

Prints one JSON line on rank 0:
  value        = pair distances binned per second, whole job (sum of all counts / time)
  roofline     = dominant kernel (rdf_cell_pair_kernel) algorithmic HBM bytes / its HIP-event time
  cpu_baseline = the C restatement of the reference path (oracle/c/rdf_oracle.c,
                 OpenMP on the host cores) on a bounded sample of the same frames;
                 its counts are also checked bit-for-bit against the GPU's.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFLOPS = 78.6   # vector fp64 spec
FP32_VALU_PEAK_TFLOPS = 157.3
CUS = 256                      # 4 SIMDs each
CLOCK_HZ = 2.4e9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="rdf", choices=["rdf", "rdf_wide", "sq", "msd", "isf"])
    ap.add_argument("--frames", type=int, default=None, help="frames per step per GPU")
    ap.add_argument("--atoms", type=int, default=None)
    ap.add_argument("--algo", default="auto", choices=["auto", "exact", "filter", "cell"])
    ap.add_argument("--blocks", type=int, default=1, help="msd: n_blocks (C4 is quoted for 1 and 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-path", action="store_true",
                    help="rdf: feed host (pageable) buffers through mdx_rdf_accumulate, i.e. the "
                         "PCIe-inclusive rate; never the headline value")
    ap.add_argument("--traj-file", action="store_true",
                    help="rdf: write the frames to an AMBER NetCDF file first and feed the engine "
                         "through the native reader (file -> pinned -> HBM inside the timed region)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline time")
    return ap.parse_args()


class World:
    """Control plane: env from torch.distributed.run; data plane: RCCL inside libmdx."""

    def __init__(self, n_gpus):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != n_gpus:
            if self.world == 1 and n_gpus > 1:
                raise SystemExit(
                    f"--gpus {n_gpus} needs one process per GPU: launch with\n  python -m "
                    f"torch.distributed.run --nnodes=1 --nproc-per-node {n_gpus} --master-addr "
                    f"127.0.0.1 --master-port 29500 bench.py --gpus {n_gpus} ...")
            raise SystemExit(f"WORLD_SIZE={self.world} does not match --gpus {n_gpus}")
        self.comm = None
        # MDX_FORCE_COMM=1 builds the communicator for a single rank too (exercises the
        # rendezvous + RCCL path on a one-GPU box)
        if self.world > 1 or os.environ.get("MDX_FORCE_COMM") == "1":
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            from mdhelper_amd.comm import rccl_comm_from_env
            self.comm = rccl_comm_from_env(self.local_rank)

    def barrier(self):
        if self.comm is not None:
            self.comm.barrier()

    def max(self, x):
        if self.comm is None:
            return x
        return float(self.comm.allreduce(np.array([x], dtype=np.float64), op="max")[0])

    def sum(self, x):
        if self.comm is None:
            return x
        return float(self.comm.allreduce(np.array([x], dtype=np.float64), op="sum")[0])


def timed_region(world, dev, steps, body, finish):
    from mdhelper_amd import _core
    world.barrier()
    _core.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        body()
    finish()
    _core.synchronize(dev)
    world.barrier()
    return world.max(time.perf_counter() - t0)


def bench_rdf(args, world, wide=False):
    from mdhelper_amd import _core
    dev = world.local_rank
    N = args.atoms or 32768
    F = args.frames or 10000
    L = 68.94 * (N / 32768.0) ** (1.0 / 3.0)
    n_bins = 201
    rng = (0.0, float(np.float32(L)) / 2) if wide else (0.0, 15.0)
    edges = np.linspace(rng[0], rng[1], n_bins + 1)
    box = np.array([L, L, L, 90, 90, 90], dtype=np.float32)

    traj = _core.synth_random_walk(F, N, box[:3], 0.3, seed=2 + world.rank, dev=dev)
    d_boxes = _core.DeviceArray.from_host(np.tile(box, (F, 1)), dev)
    eng = _core.RdfEngine(edges, (1, 1), algo=args.algo, dev=dev, timing=True)

    traj_file = None
    if args.traj_file:
        import tempfile
        from scipy.io import netcdf_file
        from mdhelper_amd.io import TrajectoryFile
        h_traj = traj.to_host()
        tmp = tempfile.NamedTemporaryFile(suffix=".nc", delete=False)
        tmp.close()
        with netcdf_file(tmp.name, "w", version=2) as nc:
            nc.Conventions = "AMBER"
            nc.createDimension("frame", None)
            nc.createDimension("spatial", 3)
            nc.createDimension("atom", N)
            nc.createDimension("cell_spatial", 3)
            nc.createDimension("cell_angular", 3)
            v_t = nc.createVariable("time", "f", ("frame",))
            v_x = nc.createVariable("coordinates", "f", ("frame", "atom", "spatial"))
            v_l = nc.createVariable("cell_lengths", "d", ("frame", "cell_spatial"))
            v_a = nc.createVariable("cell_angles", "d", ("frame", "cell_angular"))
            for f in range(F):
                v_t[f] = f
                v_x[f] = h_traj[f]
                v_l[f] = box[:3]
                v_a[f] = box[3:]
        del h_traj
        traj_file = TrajectoryFile(tmp.name)
        h_boxes = traj_file.read_boxes(np.arange(F))
        all_frames = np.arange(F)

        def step():
            eng.accumulate_traj(traj_file, all_frames, h_boxes)
    elif args.host_path:
        h_traj = traj.to_host()
        h_boxes = np.tile(box, (F, 1))

        def step():
            eng.accumulate(h_traj, None, h_boxes)
    else:
        def step():
            eng.accumulate_device(traj.ptr, N, None, N, d_boxes.ptr, F)

    for _ in range(args.warmup):
        step()
    if world.comm is not None and args.warmup:
        eng.allreduce(world.comm)      # the collective's first call (connection set-up) is warm-up too
    eng.synchronize()
    eng.reset()

    def finish():
        if world.comm is not None:
            eng.allreduce(world.comm)
        eng.synchronize()

    dt = timed_region(world, dev, args.steps, step, finish)
    counts = eng.counts()          # global sum after the all-reduce
    st = eng.stats()
    binned = int(counts.sum())
    frames_total = args.steps * F * world.world
    launches = max(st["launches"], 1)
    kernel_s = st["kernel_ms"] * 1e-3
    # one launch = one slab of frames (sort + pair kernel); algorithmic bytes 12 N + 24 per frame
    alg_bytes_per_launch = args.steps * F * (12 * N + 24) / launches
    achieved = alg_bytes_per_launch * launches / kernel_s / 1e9 if kernel_s > 0 else 0.0
    pairs_eval_rate = st["pairs_evaluated"] / kernel_s if kernel_s > 0 else 0.0
    # HBM bytes from the PMC counters come from a separate rocprofv3 run (profiles/traffic.json)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            per_frame = json.load(fh)["rdf_cell_pair_kernel"]["hbm_bytes_per_frame"]
        if N == 32768 and not wide and args.algo in ("auto", "cell"):
            traffic = per_frame * F * args.steps / launches
    except (OSError, KeyError, ValueError):
        pass
    out = {
        "metric": "pair-distances binned/sec",
        "value": binned / dt,
        "unit": "pairs/s",
        "n_gpus": world.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 filter + f64 contract arithmetic, u64 counts",
        "data": "synthetic" + (" (host buffers, PCIe copy inside the timed region)"
                               if args.host_path else ""),
        "config": {"workload": ("C2(ii)" if wide else "C2(i)") + f" RDF {N} atoms x {F} frames/GPU/step, "
                   f"L={L:.2f} A, n_bins={n_bins}, range=({rng[0]:g},{rng[1]:.4g}), exclusion=(1,1), "
                   f"algo={args.algo}" + (", host path" if args.host_path else "")
                   + (", NetCDF file through the native reader" if args.traj_file else ""),
                   "atoms": N, "frames_per_step_per_gpu": F, "n_bins": n_bins},
        "frames_per_sec": frames_total / dt,
        "pair_distances_covered_per_sec": frames_total * float(N) * N / dt,
        "pairs_binned_per_frame": binned / max(frames_total, 1),
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "kernel": "rdf_cell_pair_kernel (algo cell/auto) or rdf_tile_kernel (exact/filter)",
            "kernel_ms_per_launch": st["kernel_ms"] / launches,
            "algorithmic_bytes_per_launch": alg_bytes_per_launch,
            "note": "O(N^2) arithmetic on O(N) bytes: the kernel is VALU/LDS-atomic bound, see 'valu'",
            "valu": {
                "ordered_pairs_covered_per_sec_kernel": pairs_eval_rate,
                "distance_evaluations_per_sec_kernel": st["pairs_computed"] / kernel_s if kernel_s > 0 else 0.0,
                "evaluated_fraction_of_pair_space": st["pairs_computed"] / max(st["pairs_evaluated"], 1),
                "exact_path_fraction_of_evaluations": st["pairs_exact"] / max(st["pairs_computed"], 1),
                "image_search_path_fraction": st["cell_units_general"] / max(st["cell_units"], 1),
                # The hot step (64 evaluations) issues 10 plain VALU + v_sqrt_f32 + 2 v_cmp; at the
                # measured issue costs (profiles/r01_d_valu_issue_microbench.txt: 3.0 / 8.3 / 4.2
                # cycles per wave-instruction per SIMD) that is 46.7 cycles per step per SIMD.
                "valu_issue_bound_frac_est": (st["pairs_computed"] / kernel_s / 64.0 / (4 * CUS)
                                              * 46.7 / CLOCK_HZ) if kernel_s > 0 else 0.0,
            },
        },
    }
    if world.rank == 0 and world.world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_rdf(args, traj, box, edges, rng, n_bins, N, eng)
    if traj_file is not None:
        traj_file.close()
        os.unlink(tmp.name)
    eng.close()
    traj.free()
    d_boxes.free()
    return out


def cpu_baseline_rdf(args, traj, box, edges, rng, n_bins, N, eng):
    """C oracle (OpenMP, all host cores) on a bounded sample of the same frames + parity check."""
    from mdhelper_amd import _core
    from oracle import cbind
    # one GPU's share of the host (16 cores on the bench boxes) unless told otherwise
    threads = int(os.environ.get("MDX_CPU_THREADS", min(16, max(1, len(os.sched_getaffinity(0))))))
    frames = traj.to_host(0, 1)
    t0 = time.perf_counter()
    c0 = cbind.c_radial_histogram(frames[0], frames[0], n_bins, rng, box, exclusion=(1, 1),
                                  n_threads=threads)
    t_one = time.perf_counter() - t0
    n_sample = int(max(1, min(64, args.cpu_seconds // max(t_one, 1e-3))))
    sample = traj.to_host(0, n_sample)
    counts = np.zeros(n_bins, dtype=np.int64)
    t0 = time.perf_counter()
    for f in range(n_sample):
        cbind.c_radial_histogram(sample[f], sample[f], n_bins, rng, box, exclusion=(1, 1),
                                 n_threads=threads, counts=counts)
    t_cpu = time.perf_counter() - t0
    # parity of the very same frames on the GPU
    chk = _core.RdfEngine(edges, (1, 1), algo=args.algo, dev=eng.dev)
    chk.accumulate(sample, None, box)
    same = bool(np.array_equal(chk.counts(), counts))
    chk.close()
    return {"value": float(counts.sum()) / t_cpu, "unit": "pairs/s", "cores": threads,
            "kind": "port",
            "sample": f"{n_sample} of the bench frames, brute-force C restatement "
                      f"(oracle/c/rdf_oracle.c, OpenMP x{threads}); {t_cpu:.1f} s",
            "frames_per_sec": n_sample / t_cpu, "gpu_counts_bit_exact_on_sample": same}


def bench_sq(args, world):
    from mdhelper_amd import _core
    dev = world.local_rank
    N = args.atoms or 32768
    F = args.frames or 1000
    L = 68.94
    # 512 grid wavevectors, numpy.meshgrid's default 'xy' order (structure.py:1379-1381)
    grid = 2 * np.pi * np.arange(8) / L
    q = np.stack(np.meshgrid(grid, grid, grid), -1).reshape(-1, 3)
    sizes = [N // 2, N - N // 2]
    pairs = ((0, 0), (0, 1), (1, 1))               # mode="partial" (structure.py:1459-1464)
    traj = _core.synth_random_walk(F, N, [L, L, L], 0.3, seed=2 + world.rank, dev=dev)
    eng = _core.SqEngine(q, sizes, pairs, dev=dev, timing=True)

    def step():
        eng.accumulate_device(traj.ptr, N, F)

    for _ in range(args.warmup):
        step()
    if world.comm is not None and args.warmup:
        eng.allreduce(world.comm)
    eng.result()
    eng.reset()

    def finish():
        if world.comm is not None:
            eng.allreduce(world.comm)

    dt = timed_region(world, dev, args.steps, step, finish)
    st = eng.stats()
    ssf = eng.result()
    evals = args.steps * F * world.world * float(N) * len(q)
    kernel_s = st["kernel_ms"] * 1e-3
    alg = F * (12 * N) + 24 * len(q)
    achieved = alg * max(st["launches"], 1) / kernel_s / 1e9 if kernel_s > 0 else 0.0
    out = {
        "metric": "exp(iq.r) evaluations/sec", "value": evals / dt, "unit": "evals/s",
        "n_gpus": world.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"C3 partial S(q) {N} atoms, {len(q)} wavevectors, 2 groups, {F} frames/GPU/step"},
        "frames_per_sec": args.steps * F * world.world / dt,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "sq_rho_quads_kernel (grid wavevectors: separable phase tables in LDS, "
                               "4 columns x 8 m_z accumulators per thread)",
                     "note": "fp64 VALU bound: 4 FMAs per term + 0.5 complex products, 0.5 16-B LDS reads per "
                             "term, ~100 fp64 instructions per particle, axis and tile for the tables; "
                             "non-lattice wavevector sets take sq_rho_kernel (~40 fp64 instr each)",
                     "valu": {"evaluations_per_sec_kernel": evals / max(kernel_s, 1e-9),
                              # 4.5 v_fma_f64-class wave-instructions per 64 terms, 4.76 cycles each at the
                              # nominal 2.4 GHz on 1024 SIMDs (profiles/r01_g_valu_issue_microbench.txt)
                              "fp64_issue_bound_frac_est": evals / max(kernel_s, 1e-9) * 4.5 / 64 * 4.76
                                                           / (1024 * 2.4e9)}},
        "checksum": float(ssf.sum()),
    }
    if world.rank == 0 and world.world == 1 and not args.no_cpu_baseline:
        from oracle import fourier as of
        sample = traj.to_host(0, 2).astype(np.float64)
        t0 = time.perf_counter()
        refs = [of.fourier_sum_ref(q, sample[f]) for f in range(2)]
        t_cpu = time.perf_counter() - t0
        got = _core.fourier_sum_device(q, sample[0], dev=dev)
        err = float(np.abs(got - refs[0]).max() / np.abs(refs[0]).max())
        out["cpu_baseline"] = {"value": 2 * float(N) * len(q) / t_cpu, "unit": "evals/s", "cores": 1,
                               "kind": "port", "sample": f"2 frames, numpy exp(1j q.r) ({t_cpu:.1f} s)",
                               "gpu_max_rel_deviation_on_sample": err}
    eng.close()
    traj.free()
    return out


def bench_isf(args, world):
    """coherent + incoherent intermediate scattering functions, C3's
    particles and wavevectors, 64 lags; frames resident in HBM (``--host-path``: fed from pageable
    host memory, PCIe inside the timed region)."""
    from mdhelper_amd import _core
    dev = world.local_rank
    N = args.atoms or 32768
    F = args.frames or 256
    L = 68.94
    n_lags = 64
    grid = 2 * np.pi * np.arange(8) / L
    q = np.stack(np.meshgrid(grid, grid, grid), -1).reshape(-1, 3)
    sizes = [N // 2, N - N // 2]
    pairs = ((0, 0), (0, 1), (1, 1))
    traj = _core.synth_random_walk(F, N, [L, L, L], 0.3, seed=2 + world.rank, dev=dev)
    pos = traj.to_host(0, F)
    eng = _core.IsfEngine(q, sizes, pairs, n_lags, True, dev=dev, timing=True)

    def step():
        eng.reset()
        if args.host_path:
            eng.accumulate(pos)               # pageable host memory: PCIe inside the timed region
        else:
            eng.accumulate_device(traj.ptr, N, F)

    for _ in range(args.warmup):
        step()
    dt = timed_region(world, dev, args.steps, step, lambda: None)
    st = eng.stats()
    cisf, iisf = eng.result()
    # terms per frame: rho(q) of every particle + one displacement phase per (lag, particle, q)
    lagged = sum(min(f + 1, n_lags) for f in range(F))
    evals = args.steps * world.world * float(N) * len(q) * (F + lagged)
    kernel_s = st["kernel_ms"] * 1e-3          # of the last step: reset() clears the timer
    per_step = float(N) * len(q) * (F + lagged)
    alg = (12 * N) * F
    achieved = alg / kernel_s / 1e9 if kernel_s > 0 else 0.0
    out = {
        "metric": "exp(iq.r) evaluations/sec", "value": evals / dt, "unit": "evals/s",
        "n_gpus": world.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"ISF {N} atoms, {len(q)} wavevectors, 2 groups (partial), {n_lags} lags, "
                               f"coherent + incoherent, {F} frames/GPU/step" + (" from host memory" if args.host_path else "")},
        "frames_per_sec": args.steps * F * world.world / dt,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "isf_incoherent_quads_kernel + sq_rho_quads_kernel + isf_coherent_kernel",
                     "note": "fp64 VALU bound like S(q): two FMAs per displacement term, four per rho term",
                     "valu": {"evaluations_per_sec_kernel": per_step / max(kernel_s, 1e-9),
                              # 2.5 v_fma_f64-class wave-instructions per 64 displacement terms
                              "fp64_issue_bound_frac_est": per_step / max(kernel_s, 1e-9) * 2.5 / 64 * 4.76
                                                           / (1024 * 2.4e9)}},
        "checksum": float(cisf.sum() + iisf.sum()),
    }
    if world.rank == 0 and world.world == 1 and not args.no_cpu_baseline:
        from oracle import fourier as of
        n_s, f_s, lags_s = 2048, 8, 4
        sample = pos[:f_s, :n_s].astype(np.float64)
        t0 = time.perf_counter()
        ref = of.isf_run_ref(sample, [n_s // 2, n_s - n_s // 2], q, lags_s, mode="partial", incoherent=True,
                             sort=False, unique=False)
        t_cpu = time.perf_counter() - t0
        small = _core.IsfEngine(q, [n_s // 2, n_s - n_s // 2], pairs, lags_s, True, dev=dev)
        small.accumulate(pos[:f_s, :n_s])
        gc, gi = small.result()
        small.close()
        norm = n_s * np.arange(f_s, f_s - lags_s, -1)[:, None, None]
        err = max(float(np.abs(gc / norm - ref["cisf"]).max() / np.abs(ref["cisf"]).max()),
                  float(np.abs(gi / norm - ref["iisf"]).max() / np.abs(ref["iisf"]).max()))
        terms = float(n_s) * len(q) * (f_s + sum(min(f + 1, lags_s) for f in range(f_s)))
        out["cpu_baseline"] = {"value": terms / t_cpu, "unit": "evals/s", "cores": 1, "kind": "port",
                               "sample": f"{f_s} frames x {n_s} atoms x {lags_s} lags, numpy restatement "
                                         f"({t_cpu:.1f} s)",
                               "gpu_max_rel_deviation_on_sample": err}
    eng.close()
    traj.free()
    return out


def bench_msd(args, world):
    from mdhelper_amd import _core
    dev = world.local_rank
    N = args.atoms or 10000
    T = args.frames or 100000
    # particles shard across ranks (transport.py:1036-1039: per-particle MSDs are independent)
    traj = _core.synth_random_walk(T, N, [1.0, 1.0, 1.0], 0.1, seed=4 + world.rank, dev=dev,
                                   dtype=np.float64)
    B = max(1, args.blocks)
    eng = _core.MsdEngine(T // B, B, 2, dev=dev, timing=True)

    def step():
        eng.reset()
        eng.push_device(0, traj.ptr, N, 0, N // 2)
        eng.push_device(1, traj.ptr, N, N // 2, N - N // 2)

    box = {}

    def finish():
        if world.comm is not None:
            eng.allreduce(world.comm)
        box["msd"], box["traj"] = eng.result()

    for _ in range(args.warmup):     # includes the inverse-transform plan (rocFFT builds it once)
        step()
        finish()

    dt = timed_region(world, dev, args.steps, step, finish)
    st = eng.stats()
    atom_frames = args.steps * float(N) * T * world.world
    alg_bytes = 24.0 * N * T
    kernel_s = st["kernel_ms"] * 1e-3
    achieved = alg_bytes / max(kernel_s, 1e-9) / 1e9
    msd = box["msd"][0, 0] / (N // 2)
    traffic = None
    try:   # PMC pass of the same workload (scripts/profile_pmc.sh), per step
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            if N == 10000 and T == 100000 and B == 1 and not os.environ.get("MDX_MSD_ROCFFT"):
                key = {204800: "msd_c4_step", 262144: "msd_c4_step_pow2"}.get(eng.n_fft)
                if key:
                    traffic = json.load(fh)[key]["hbm_bytes_per_step"]
    except (OSError, KeyError, ValueError):
        pass
    out = {
        "metric": "MSD atom-frames/sec", "value": atom_frames / dt, "unit": "atom-frames/s",
        "n_gpus": world.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"C4 self-MSD {N} atoms x {T} frames, 2 groups, n_blocks={B}, n_fft={eng.n_fft}"
                               + (" (own two-pass transform)" if eng.n_fft in (1 << 13, 1 << 14, 1 << 15, 1 << 16, 204800,
                                                                                  1 << 18, 1 << 19, 1 << 20)
                                  and not os.environ.get("MDX_MSD_ROCFFT") else " (rocFFT)")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "msd pipeline of one step: sums + forward transforms + power "
                               "(msd_fft_cols/rows_power kernels for n_fft = 2^13..2^16, 2^18..2^20, else gather + "
                               "rocFFT R2C + power)",
                     "pipeline_bytes_model": st["bytes_moved"]},
        "physics_check_msd_over_3sigma2m": float(msd[10] / (3 * 0.01 * 10)),
    }
    if world.rank == 0 and world.world == 1 and not args.no_cpu_baseline:
        # the reference's msd_fft (scipy FFTs, one core) on a bounded sample of particles, and
        # the engine on the same particles checked against it
        from oracle import correlation as oc
        n_s = 64
        small = _core.synth_random_walk(T, n_s, [1.0, 1.0, 1.0], 0.1, seed=99, dev=dev, dtype=np.float64)
        h_small = small.to_host()
        t0 = time.perf_counter()
        ref = oc.msd_fft_ref(h_small[None], axis=1, average=False)[0].sum(axis=-1)
        t_cpu = time.perf_counter() - t0
        chk = _core.MsdEngine(T, 1, 1, dev=dev)
        chk.push_device(0, small.ptr, n_s, 0, n_s)
        got = chk.result()[0][0, 0]
        chk.close()
        small.free()
        err = float(np.abs(got - ref).max() / np.abs(ref).max())
        out["cpu_baseline"] = {"value": n_s * float(T) / t_cpu, "unit": "atom-frames/s", "cores": 1,
                               "kind": "port",
                               "sample": f"{n_s} particles x {T} frames, scipy-FFT msd_fft restatement "
                                         f"(oracle/correlation.py), {t_cpu:.1f} s",
                               "gpu_max_rel_deviation_on_sample": err}
    eng.close()
    traj.free()
    return out


def main():
    args = parse()
    # Libraries underneath (gloo, RCCL) print banners on stdout; keep stdout clean for the
    # one JSON line by pointing fd 1 at stderr until the result is ready.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    world = World(args.gpus)
    from mdhelper_amd import _lib
    _lib.require_device(world.local_rank)
    if args.workload in ("rdf", "rdf_wide"):
        out = bench_rdf(args, world, wide=args.workload == "rdf_wide")
    elif args.workload == "sq":
        out = bench_sq(args, world)
    elif args.workload == "isf":
        out = bench_isf(args, world)
    else:
        out = bench_msd(args, world)
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if world.rank == 0:
        print(json.dumps(out), flush=True)
    os.dup2(2, 1)          # teardown chatter goes to stderr as well
    if world.comm is not None:
        world.comm.close()
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
        except ImportError:
            pass


if __name__ == "__main__":
    main()
