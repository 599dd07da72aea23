#!/usr/bin/env python3
"""
bench.py — headline benchmark of the MI355X-native mdhelper hot path.

    python bench.py --gpus N --steps K --warmup W [--workload rdf|rdf_wide|sq|msd|isf]

Default workload (BASELINE.json configs[1], "C2"): RDF on 32 768 atoms, cubic
box L = 68.94 A (rho = 0.1 A^-3), n_bins = 201, range = (0, 15) A, self RDF with
exclusion = (1, 1); synthetic wrapped Gaussian random walk generated in HBM.
One *step* = one pass of the hot path (mdx_rdf_accumulate_device) over one batch
of `--frames` frames already resident in HBM.

N > 1: one process per GPU.  `python bench.py --gpus N` starts its own N ranks
(mdhelper_amd/launch.py: a parent that never touches the GPU spawns one fresh
child per device and relays rank 0's line); under `python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N` the ranks are the launcher's.  Either way
the ranks meet on a node-local rendezvous socket (the 128-byte RCCL id travels
there; no torch in the path), every rank owns its own batch of frames (weak
scaling, frames shard with no data-path collective) and the per-rank histograms
meet in ONE RCCL all-reduce at the end of the timed region.

Prints one JSON line on rank 0:
  value        = pair distances binned per second, whole job (sum of all counts / time)
  roofline     = dominant kernel (rdf_cell_pair_kernel): the VALU issue bound that binds it
                 (hot-loop trips counted live x instruction counts from the SQ-counter profile of
                 the same sources) with the HBM figure beside it under "hbm"
  cpu_baseline = NumPy restatement of the reference path on one core; cpu_baseline_parallel =
                 the same over every core of the affinity mask; cpu_baseline_c = the C/OpenMP
                 restatement on whole frames, whose counts are also checked bit-for-bit
                 against the GPU's.
  extra        = short C3 S(q), C4 MSD and C2(ii) wide-range legs (N = 1 only), each a full
                 line of its own workload with roofline + cpu_baseline.
"""

import argparse
import hashlib
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
CUS = 256                      # 4 SIMD-32 each
SIMDS = 4 * CUS
CLOCK_HZ = 2.4e9               # nominal peak engine clock (MI355X_MICROARCH.md)
VALU_CYCLES = 2.0              # wave64 VALU instruction on a SIMD-32: 2 cycles (guide, constants table)
VALU_TRANS_CYCLES = 4.0        # quarter-rate transcendentals (v_sqrt_f32, ...): twice that


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="rdf", choices=["rdf", "rdf_wide", "sq", "msd", "isf", "ingest"])
    ap.add_argument("--frames", type=int, default=None, help="frames per step per GPU")
    ap.add_argument("--atoms", type=int, default=None)
    ap.add_argument("--algo", default="auto", choices=["auto", "exact", "filter", "cell"])
    ap.add_argument("--blocks", type=int, default=1, help="msd: n_blocks (C4 is quoted for 1 and 8)")
    ap.add_argument("--n-points", type=int, default=8,
                    help="sq: grid points per axis (8 -> the 512 wavevectors of C3; 32 -> the reference's "
                         "default StructureFactor grid of 32 768, structure.py:1324, 1376-1381)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="rdf at N=1: skip the short C3 / C4 / C2(ii) legs embedded under 'extra'")
    ap.add_argument("--no-ingest", action="store_true",
                    help="sq / isf at N=1: skip the host-memory / file / operator-surface legs under 'ingest'")
    ap.add_argument("--dry-run", action="store_true",
                    help="walk the N-rank control plane without touching a GPU: launcher, rendezvous, the 128-byte id "
                         "broadcast, the shard plan of the workload, one host all-reduce; prints the plan")
    ap.add_argument("--no-onsager", action="store_true",
                    help="msd at N=1: skip the operator-surface legs (Onsager(...).run() on HBM / host / file data)")
    ap.add_argument("--host-path", action="store_true",
                    help="rdf: feed host (pageable) buffers through mdx_rdf_accumulate, i.e. the "
                         "PCIe-inclusive rate; never the headline value")
    ap.add_argument("--traj-file", action="store_true",
                    help="rdf: write the frames to an AMBER NetCDF file first and feed the engine "
                         "through the native reader (file -> pinned -> HBM inside the timed region)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target time of each CPU-baseline leg")
    ap.add_argument("--shard-fixed", action="store_true",
                    help="strong scaling check: ONE fixed set of --frames frames (rdf, sq) or --atoms "
                         "particles (msd), the same for any --gpus, sharded across the ranks; the reduced "
                         "result must not depend on the rank count ('result_digest')")
    ap.add_argument("--share-devices", action="store_true",
                    help="tests on a box with fewer GPUs than ranks: rank r uses device r %% devices and the "
                         "accumulators meet in a host all-reduce over the rendezvous socket (RCCL cannot put "
                         "two ranks on one GPU); never a performance number")
    return ap.parse_args(argv)


class World:
    """
    One process per GPU.  Control plane: RANK / LOCAL_RANK / WORLD_SIZE from the launcher
    (``mdhelper_amd.launch.launch`` when this script starts its own ranks, or
    ``torch.distributed.run``) + the node-local rendezvous socket; data plane: RCCL inside libmdx.
    """

    def __init__(self, args):
        from mdhelper_amd import _lib
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"WORLD_SIZE={self.world} does not match --gpus {args.gpus}")
        n_dev = _lib.device_count()
        if n_dev == 0:
            _lib.require_device(0)
        self.dev = self.local_rank % n_dev if args.share_devices else self.local_rank
        _lib.require_device(self.dev)
        self.comm = None
        self.kind = None
        self.rccl_ranks = None
        # MDX_FORCE_COMM=1 builds the RCCL communicator for a single rank too
        if self.world > 1 or os.environ.get("MDX_FORCE_COMM") == "1":
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            from mdhelper_amd.launch import Rendezvous, SocketComm
            rdzv = Rendezvous(self.rank, self.world) if self.world > 1 else None
            if args.share_devices and self.world > 1 and os.environ.get("MDX_BENCH_TRY_RCCL") != "1":
                self.comm, self.kind = SocketComm(rdzv), "host-socket (ranks share devices: test mode)"
            else:
                from mdhelper_amd.comm import rccl_comm_from_env, rccl_comm_or_socket
                if rdzv is None:
                    self.comm, self.kind = rccl_comm_from_env(self.dev, rdzv), "rccl"
                else:
                    # RCCL where every rank can build it; else the accumulators are summed over the rendezvous
                    # socket and the line says so in "comm" (the kernels and the timed region do not change)
                    self.comm, self.kind = rccl_comm_or_socket(
                        self.dev, rdzv, float(os.environ.get("MDX_RCCL_INIT_TIMEOUT", "180")))
                if self.kind == "rccl":
                    n, r, d = self.comm.rccl_info()
                    if (n, r) != (self.world, self.rank):
                        raise SystemExit(f"RCCL reports rank {r} of {n}, expected {self.rank} of {self.world}")
                    self.rccl_ranks = n
                elif self.rank == 0:
                    sys.stderr.write(f"bench.py: {self.kind}\n")

    @property
    def device_collectives(self):
        return self.kind == "rccl"

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

    def gather(self, x):
        """Every rank's scalar, in rank order."""
        if self.comm is None:
            return [float(x)]
        v = np.zeros(self.world, dtype=np.float64)
        v[self.rank] = x
        return [float(t) for t in self.comm.allreduce(v, op="sum")]

    def reduce_host(self, arr):
        """Sum of a host array over the ranks (the SocketComm route of the accumulators)."""
        return arr if self.comm is None else self.comm.allreduce(arr, op="sum")

    def describe(self):
        # "rccl": the accumulators of this run met in an RCCL all-reduce (false at N = 1 — nothing to reduce —
        # and on the host-socket route, which earns no multi-GPU credit)
        return {"comm": self.kind, "rccl_ranks": self.rccl_ranks, "rccl": self.kind == "rccl"}

    def close(self):
        if self.comm is not None:
            rdzv = getattr(self.comm, "rdzv", None)
            self.comm.close()
            if rdzv is not None and self.kind == "rccl":
                rdzv.close()


def timed_region(world, dev, steps, body, finish):
    """Barrier + device synchronise on both sides; returns (max over ranks, this rank's own time
    up to its last kernel, before the closing barrier)."""
    from mdhelper_amd import _core
    world.barrier()
    _core.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        body()
    finish()
    _core.synchronize(dev)
    own = time.perf_counter() - t0
    world.barrier()
    return world.max(time.perf_counter() - t0), own


def source_digest(*names):
    """sha256 of kernel source files: ties counter / traffic figures read from profiles/ to the
    build they were measured on."""
    h = hashlib.sha256()
    for name in names:
        with open(os.path.join(ROOT, "mdhelper_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def profiled(key, *sources):
    """Entry ``key`` of profiles/counters.json (figures from separate rocprofv3 PMC passes, written by
    scripts/profile_*.sh) — or None when it was measured on other kernel sources than these."""
    try:
        with open(os.path.join(ROOT, "profiles", "counters.json")) as fh:
            entry = json.load(fh)[key]
    except (OSError, KeyError, ValueError):
        return None
    if entry.get("source_digest") != source_digest(*sources):
        return None
    return entry


def max_rel_deviation(got, ref, floor=1e-9):
    """Largest element-wise relative deviation over the entries whose reference magnitude exceeds
    ``floor`` x the largest one (entries that are pure round-off — lag 0 of an MSD — are left out):
    max |got - ref| / |ref|, not a norm over the array."""
    got, ref = np.asarray(got), np.asarray(ref)
    mag = np.abs(ref)
    keep = mag > floor * mag.max()
    return float((np.abs(got - ref)[keep] / mag[keep]).max())


def write_amber_netcdf_fast(path, positions, box):
    """AMBER NetCDF (64-bit-offset container, the reference's own, openmm/file.py:49-52) of
    float32[F, N, 3] frames with one cell for all of them: the header comes from scipy.io.netcdf_file
    (an independent NetCDF-3 writer) with one record, the F records are then written as one
    big-endian structured array — scipy's own per-record writes take minutes at bench sizes.
    tests/test_traj_io.py reads such a file back through scipy."""
    from scipy.io import netcdf_file
    F, N, _ = positions.shape
    with netcdf_file(path, "w", version=2) as nc:
        nc.Conventions = "AMBER"
        nc.ConventionVersion = "1.0"
        nc.createDimension("frame", None)
        nc.createDimension("spatial", 3)
        nc.createDimension("atom", N)
        nc.createDimension("cell_spatial", 3)
        nc.createDimension("cell_angular", 3)
        v_t = nc.createVariable("time", "f", ("frame",))
        v_x = nc.createVariable("coordinates", "f", ("frame", "atom", "spatial"))
        v_l = nc.createVariable("cell_lengths", "d", ("frame", "cell_spatial"))
        v_a = nc.createVariable("cell_angles", "d", ("frame", "cell_angular"))
        v_t[0] = 0.0
        v_x[0] = positions[0]
        v_l[0] = box[:3]
        v_a[0] = box[3:]
    # record variables are laid out in definition order, each padded to 4 bytes; the file ends
    # with its single record
    rec = np.dtype([("time", ">f4"), ("coordinates", ">f4", (N, 3)), ("cell_lengths", ">f8", (3,)),
                    ("cell_angles", ">f8", (3,))])
    first = os.path.getsize(path) - rec.itemsize
    with open(path, "r+b") as fh:
        fh.seek(4)
        fh.write(int(F).to_bytes(4, "big"))          # numrecs
        fh.seek(first)
        step = max(1, (256 << 20) // rec.itemsize)
        for f0 in range(0, F, step):
            n = min(step, F - f0)
            block = np.empty(n, dtype=rec)
            block["time"] = np.arange(f0, f0 + n)
            block["coordinates"] = positions[f0:f0 + n]
            block["cell_lengths"] = box[:3]
            block["cell_angles"] = box[3:]
            block.tofile(fh)
        fh.truncate()


RDF_SOURCES = ("mdx_rdf.hip", "mdx_rdf_cell.hpp", "mdx_rdf_device.hpp")
MSD_SOURCES = ("mdx_msd.hip", "mdx_msd_fft.hpp")
SQ_SOURCES = ("mdx_sq.hip", "mdx_sq_device.hpp")
ISF_SOURCES = ("mdx_isf.hip", "mdx_sq_device.hpp")


def bench_rdf(args, world, wide=False):
    from mdhelper_amd import _core
    from mdhelper_amd.comm import shard_range
    dev = world.dev
    N = args.atoms or 32768
    F = args.frames or 10000
    L = 68.94 * (N / 32768.0) ** (1.0 / 3.0)
    n_bins = 201
    rng = (0.0, float(np.float32(L)) / 2) if wide else (0.0, 15.0)
    edges = np.linspace(rng[0], rng[1], n_bins + 1)
    box = np.array([L, L, L, 90, 90, 90], dtype=np.float32)

    # weak scaling: every rank owns its own F frames (seeded by rank); --shard-fixed: ONE set of F
    # frames (same seed everywhere), rank r takes the contiguous share shard_range(F, r, world)
    seed = 2 if args.shard_fixed else 2 + world.rank
    traj = _core.synth_random_walk(F, N, box[:3], 0.3, seed=seed, dev=dev)
    lo, hi = shard_range(F, world.rank, world.world) if args.shard_fixed else (0, F)
    F_mine = hi - lo
    d_boxes = _core.DeviceArray.from_host(np.tile(box, (F, 1)), dev)
    eng = _core.RdfEngine(edges, (1, 1), algo=args.algo, dev=dev, timing=True)

    traj_file = None
    if args.traj_file:
        import tempfile
        from mdhelper_amd.io import TrajectoryFile
        h_traj = traj.to_host()
        tmp = tempfile.NamedTemporaryFile(suffix=".nc", delete=False)
        tmp.close()
        write_amber_netcdf_fast(tmp.name, h_traj, box)
        del h_traj
        traj_file = TrajectoryFile(tmp.name)
        my_frames = np.arange(lo, hi)
        h_boxes = traj_file.read_boxes(my_frames)

        def step():
            eng.accumulate_traj(traj_file, my_frames, h_boxes)
    elif args.host_path:
        h_traj = traj.to_host(lo, F_mine)
        h_boxes = np.tile(box, (F_mine, 1))

        def step():
            eng.accumulate(h_traj, None, h_boxes)
    else:
        def step():
            if F_mine:
                eng.accumulate_device(traj.offset(lo), N, None, N, d_boxes.offset(lo), F_mine)

    res = {}

    def finish():
        # ONE all-reduce of the accumulators: RCCL on the uint64 counts in HBM, or (ranks sharing a
        # device in tests) the host copies over the rendezvous socket
        if world.device_collectives:
            eng.allreduce(world.comm)
        eng.synchronize()
        res["counts"] = eng.counts() if (world.comm is None or world.device_collectives) \
            else world.reduce_host(eng.counts())

    for _ in range(args.warmup):
        step()
    if args.warmup:
        finish()                       # the collective's first call (connection set-up) is warm-up too
    eng.synchronize()
    eng.reset()

    dt, own = timed_region(world, dev, args.steps, step, finish)
    counts = res["counts"]             # global sum after the all-reduce
    st = eng.stats()
    binned = int(counts.sum())
    frames_total = args.steps * (F if args.shard_fixed else F * world.world)
    launches = max(st["launches"], 1)
    kernel_s = st["kernel_ms"] * 1e-3
    # one launch = one slab of frames (sort + pair kernel); algorithmic bytes 12 N + 24 per frame
    alg_bytes_per_launch = args.steps * F_mine * (12 * N + 24) / launches
    achieved = alg_bytes_per_launch * launches / kernel_s / 1e9 if kernel_s > 0 else 0.0
    steps64 = st["pairs_computed"] / 64.0          # hot-loop trips: 64 distance evaluations each
    # this rank's own share of the reduced counts (weak scaling: every rank bins the same amount)
    binned_mine = binned / (1 if args.shard_fixed else world.world) * (F_mine / max(F, 1) if args.shard_fixed else 1.0)
    celled = args.algo in ("cell", "auto")      # auto: the cell-sorted kernel at every size (round 4)
    # VALU / SALU / LDS instructions per hot-loop trip and the engine clock under this kernel come from
    # SQ counters of a separate rocprofv3 pass over the SAME sources (scripts/profile_counters.sh ->
    # profiles/counters.json); dropped (None) when the kernel sources have changed since
    # (entries: C2(i), C2(ii), the C5 size, a C1-like small system; another size takes the entry measured at
    # the nearest size on the same kernel and says so — the mix per trip moves by a few per cent with N)
    ctr, ctr_atoms = None, None
    if args.algo == "auto":
        names = ["rdf_wide", "rdf_c1"] if wide else ["rdf_c2", "rdf_c5"]
        found = [(abs(np.log(e.get("atoms", 32768) / N)), e) for e in
                 (profiled(n, *RDF_SOURCES) for n in names) if e]
        if found:
            ctr = min(found, key=lambda t: t[0])[1]
            ctr_atoms = ctr.get("atoms", 32768)
    clock = st.get("clock_hz") or None
    valu = None
    if ctr and kernel_s > 0:
        clk = clock or ctr.get("clock_hz") or CLOCK_HZ
        cyc_per_step = ctr["valu_plain_per_step"] * VALU_CYCLES + ctr["valu_trans_per_step"] * VALU_TRANS_CYCLES
        issue = steps64 / kernel_s * cyc_per_step            # SIMD issue cycles consumed per second
        # useful work beside the utilisation: the hot step's own 13 instructions (12 plain + v_sqrt_f32 =
        # 28 issue cycles per 64 distance evaluations), counted once per evaluation that was made and
        # once per evaluation that ended in a bin
        useful = steps64 / kernel_s * (12 * VALU_CYCLES + VALU_TRANS_CYCLES)
        # a self histogram evaluates each unordered pair ONCE and adds 2: the evaluations that
        # end in a bin are half the (ordered) counts
        binned_share = min(1.0, 0.5 * binned_mine / max(st["pairs_computed"], 1))
        valu = {"achieved": issue / 1e9, "peak": SIMDS * clk / 1e9, "frac": issue / (SIMDS * clk),
                "frac_evaluations": useful / (SIMDS * clk),
                "frac_binned": useful * binned_share / (SIMDS * clk),
                "frac_at_nominal_clock": issue / (SIMDS * CLOCK_HZ),
                "clock_hz": clk, "clock_source": "s_memtime / s_memrealtime inside this run's kernel"
                if clock else "profiles/counters.json",
                "valu_instructions_per_step": ctr["valu_plain_per_step"] + ctr["valu_trans_per_step"],
                "transcendentals_per_step": ctr["valu_trans_per_step"],
                "salu_instructions_per_step": ctr.get("salu_per_step"),
                "lds_instructions_per_step": ctr.get("lds_per_step"),
                "valu_busy_fraction_of_simd_cycles": ctr.get("valu_busy_frac"),
                "counters_source": ctr.get("source"),
                "counters_measured_at_atoms": ctr_atoms}
    traffic = ctr.get("hbm_bytes_per_frame") if (ctr and ctr_atoms == N) else None
    hbm = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS,
           "traffic": traffic * F_mine * args.steps / launches if traffic else None,
           "traffic_source": ctr.get("traffic_source") if (ctr and traffic) else None,
           "algorithmic_bytes_per_launch": alg_bytes_per_launch,
           "note": "O(N^2) arithmetic on O(N) bytes: ~1e-3 of the HBM peak by construction"}
    roofline = {
        # the bound that binds: VALU issue of the pair loop (O(N^2) work on O(N) bytes)
        "bound": "valu", "unit": "G SIMD issue cycles/s",
        "achieved": valu["achieved"] if valu else None, "peak": valu["peak"] if valu else SIMDS * CLOCK_HZ / 1e9,
        "frac": valu["frac"] if valu else None,
        "frac_evaluations": valu["frac_evaluations"] if valu else None,
        "frac_binned": valu["frac_binned"] if valu else None,
        "traffic": hbm["traffic"],
        "kernel": "rdf_cell_pair_kernel" if celled else "rdf_tile_kernel",
        "kernel_ms_per_launch": st["kernel_ms"] / launches,
        "launches": launches,
        "definition": "hot-loop trips/s (64 distance evaluations each, counted by the kernel) x VALU issue "
                      "cycles per trip (SQ_INSTS_VALU per trip from profiles/, 2 cycles per wave64 VALU "
                      "instruction on a SIMD-32, 4 for v_sqrt_f32: MI355X_MICROARCH.md) / (1024 SIMDs x "
                      "engine clock measured inside the kernel); frac_evaluations / frac_binned: the same with "
                      "the hot step's own 13 instructions (28 cycles) per 64 evaluations made / binned — "
                      "useful work instead of issue-slot utilisation; traffic: sort + pair kernels",
        "valu": valu,
        "hbm": hbm,
        "work": {
            "ordered_pairs_covered_per_sec_kernel": st["pairs_evaluated"] / kernel_s if kernel_s > 0 else 0.0,
            "distance_evaluations_per_sec_kernel": st["pairs_computed"] / kernel_s if kernel_s > 0 else 0.0,
            "evaluated_fraction_of_pair_space": st["pairs_computed"] / max(st["pairs_evaluated"], 1),
            "binned_fraction_of_pair_space": binned / max(frames_total * float(N) * N, 1.0),
            # (the first counts evaluations of UNORDERED pairs over the N^2 ordered pair space, the second ordered
            # counts: an evaluation that lands in a bin adds 2)
            "evaluations_per_binned_unordered_pair": st["pairs_computed"] / max(0.5 * binned_mine, 1.0),
            "exact_path_fraction_of_evaluations": st["pairs_exact"] / max(st["pairs_computed"], 1),
            "image_search_path_fraction": st["cell_units_general"] / max(st["cell_units"], 1),
        },
    }
    per_rank = world.gather(args.steps * F_mine / own)
    out = {
        "metric": "pair-distances binned/sec",
        "value": binned / dt,
        "unit": "pairs/s",
        "n_gpus": world.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if args.shard_fixed else "weak", "vs_baseline": None,
        "dtype": "f32 filter + f64 contract arithmetic, u64 counts",
        "data": "synthetic" + (" (host buffers, PCIe copy inside the timed region)"
                               if args.host_path else ""),
        "config": {"workload": ("C2(ii)" if wide else "C2(i)") + f" RDF {N} atoms x {F} frames"
                   + ("/job/step (fixed set, sharded)" if args.shard_fixed else "/GPU/step")
                   + f", L={L:.2f} A, n_bins={n_bins}, range=({rng[0]:g},{rng[1]:.4g}), exclusion=(1,1), "
                   f"algo={args.algo}" + (", host path" if args.host_path else "")
                   + (", NetCDF file through the native reader" if args.traj_file else ""),
                   "atoms": N, "frames_per_step_per_gpu": F_mine, "n_bins": n_bins},
        "frames_per_sec": frames_total / dt,
        "per_rank_frames_per_sec": per_rank,
        **world.describe(),
        "pair_distances_covered_per_sec": frames_total * float(N) * N / dt,
        "pairs_binned_per_frame": binned / max(frames_total, 1),
        "result_digest": hashlib.sha256(np.ascontiguousarray(counts, dtype=np.int64).tobytes()).hexdigest()[:16],
        "roofline": roofline,
    }
    if world.rank == 0 and world.world == 1 and not args.no_cpu_baseline:
        out.update(cpu_baseline_rdf(args, traj, box, edges, rng, n_bins, N, eng))
    if traj_file is not None:
        traj_file.close()
        os.unlink(tmp.name)
    eng.close()
    traj.free()
    d_boxes.free()
    return out


def cpu_baseline_rdf(args, traj, box, edges, rng, n_bins, N, eng):
    """
    CPU restatements of the reference path on bounded samples of the bench frames:
    ``cpu_baseline``          NumPy pair list + numpy.histogram on ONE core (the reference's serial run());
    ``cpu_baseline_parallel`` the same over len(os.sched_getaffinity(0)) worker processes (parallel=True);
    ``cpu_baseline_c``        brute-force C/OpenMP restatement (oracle/c/rdf_oracle.c) on whole frames, whose
                              counts are also compared bit-for-bit with the GPU's on the same frames.
    (An installed reference would take MDAnalysis' nsgrid cell list at this size; it is not available
    here, so none of these is a statement about that code path.)
    """
    from mdhelper_amd import _core
    from oracle import cbind, cpu_bench
    # one GPU's share of the host (16 cores per GPU on the 8-GPU bench hosts) unless told otherwise
    visible = max(1, len(os.sched_getaffinity(0)))
    threads = int(os.environ.get("MDX_CPU_THREADS", min(16, visible)))
    cores = threads
    frame = traj.to_host(0, 1)[0]
    out = {}
    # -- NumPy, one core: calibrate on 32 rows, then a sample sized to --cpu-seconds
    _c, p0, t0 = cpu_bench.time_rdf_numpy(frame, box, n_bins, rng, (1, 1), 32, 1)
    rows = int(max(32, min(N, 32 * args.cpu_seconds / max(t0, 1e-3))))
    c1, p1, t1 = cpu_bench.time_rdf_numpy(frame, box, n_bins, rng, (1, 1), rows, 1)
    chk = cbind.c_radial_histogram(frame[:rows], frame, n_bins, rng, box, exclusion=(1, 1), n_threads=threads)
    out["cpu_baseline"] = {
        "value": float(c1.sum()) / t1, "unit": "pairs/s", "cores": 1, "kind": "port",
        "sample": f"rows 0..{rows} of frame 0 against all {N} particles ({p1:.3g} ordered pairs), NumPy "
                  f"restatement of structure.py:92-104 (pair distances + numpy.histogram, oracle/rdf.py), "
                  f"{t1:.1f} s; numpy {np.__version__}",
        "frames_per_sec": p1 / float(N) / N / t1,
        "equals_c_restatement_on_sample": bool(np.array_equal(c1, chk))}
    # -- the neighbour-search route (periodic k-d tree), one core: what capped_distance does at this size
    try:
        _ck, _pk, tk0 = cpu_bench.time_rdf_kdtree(frame, box, n_bins, rng, (1, 1), 256)
        rows_k = int(max(256, min(N, 256 * args.cpu_seconds / max(tk0, 1e-3))))
        ck, pk, tk = cpu_bench.time_rdf_kdtree(frame, box, n_bins, rng, (1, 1), rows_k)
        chk_k = cbind.c_radial_histogram(frame[:rows_k], frame, n_bins, rng, box, exclusion=(1, 1), n_threads=threads)
        out["cpu_baseline_celllist"] = {
            "value": float(ck.sum()) / tk, "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": f"rows 0..{rows_k} of frame 0 against all {N} particles: scipy cKDTree(boxsize) pairs within "
                      f"the range end + numpy.histogram, tree build included, {tk:.1f} s — the algorithm family "
                      f"MDAnalysis' capped_distance (pkdtree / nsgrid) uses at this size; MDAnalysis itself is "
                      f"not installed",
            "frames_per_sec": pk / float(N) / N / tk,
            "counts_differing_from_c_restatement": int(np.abs(ck - chk_k).sum())}
    except Exception as exc:                      # scipy missing or too old: the other baselines stand
        out["cpu_baseline_celllist"] = {"value": None, "error": repr(exc)[:200]}
    # -- NumPy, every core of this process's affinity mask
    rows_w = max(8, min(N // cores, int(rows * 0.7)))
    cp, pp, tp = cpu_bench.time_rdf_numpy(frame, box, n_bins, rng, (1, 1), rows_w, cores)
    out["cpu_baseline_parallel"] = {
        "value": float(cp.sum()) / tp, "unit": "pairs/s", "cores": cores, "kind": "port",
        "sample": f"{cores} worker processes (one GPU's share of the {visible} cores in the affinity mask) x "
                  f"{rows_w} rows of frame 0 ({pp:.3g} ordered pairs), same NumPy restatement, {tp:.1f} s "
                  f"(longest worker)",
        "frames_per_sec": pp / float(N) / N / tp}
    # -- C/OpenMP brute force on whole frames + parity of those frames on the GPU
    t0 = time.perf_counter()
    cbind.c_radial_histogram(frame, frame, n_bins, rng, box, exclusion=(1, 1), n_threads=threads)
    t_one = time.perf_counter() - t0
    n_sample = int(max(1, min(64, args.cpu_seconds // max(t_one, 1e-3))))
    sample = traj.to_host(0, n_sample)
    counts = np.zeros(n_bins, dtype=np.int64)
    t0 = time.perf_counter()
    for f in range(n_sample):
        cbind.c_radial_histogram(sample[f], sample[f], n_bins, rng, box, exclusion=(1, 1),
                                 n_threads=threads, counts=counts)
    t_cpu = time.perf_counter() - t0
    chk = _core.RdfEngine(edges, (1, 1), algo=args.algo, dev=eng.dev)
    chk.accumulate(sample, None, box)
    same = bool(np.array_equal(chk.counts(), counts))
    chk.close()
    out["cpu_baseline_c"] = {
        "value": float(counts.sum()) / t_cpu, "unit": "pairs/s", "cores": threads, "kind": "port",
        "sample": f"{n_sample} whole bench frames, brute-force C restatement (oracle/c/rdf_oracle.c, "
                  f"OpenMP x{threads}); {t_cpu:.1f} s",
        "frames_per_sec": n_sample / t_cpu, "gpu_counts_bit_exact_on_sample": same}
    return out


def bench_sq(args, world):
    from mdhelper_amd import _core
    from mdhelper_amd.comm import shard_range
    dev = world.dev
    N = args.atoms or 32768
    F = args.frames or 1000
    L = 68.94
    # n_points^3 grid wavevectors (512 at C3), numpy.meshgrid's default 'xy' order (structure.py:1379-1381)
    grid = 2 * np.pi * np.arange(args.n_points) / L
    q = np.stack(np.meshgrid(grid, grid, grid), -1).reshape(-1, 3)
    sizes = [N // 2, N - N // 2]
    pairs = ((0, 0), (0, 1), (1, 1))               # mode="partial" (structure.py:1459-1464)
    seed = 2 if args.shard_fixed else 2 + world.rank
    traj = _core.synth_random_walk(F, N, [L, L, L], 0.3, seed=seed, dev=dev)
    lo, hi = shard_range(F, world.rank, world.world) if args.shard_fixed else (0, F)
    F_mine = hi - lo
    eng = _core.SqEngine(q, sizes, pairs, dev=dev, timing=True)

    def step():
        if F_mine:
            eng.accumulate_device(traj.offset(lo), N, F_mine)

    res = {}

    def finish():
        if world.device_collectives:
            eng.allreduce(world.comm)
        res["ssf"] = eng.result() if (world.comm is None or world.device_collectives) \
            else world.reduce_host(eng.result())

    for _ in range(args.warmup):
        step()
    if args.warmup:
        finish()
    eng.reset()

    dt, own = timed_region(world, dev, args.steps, step, finish)
    st = eng.stats()
    ssf = res["ssf"]
    frames_total = args.steps * (F if args.shard_fixed else F * world.world)
    evals = frames_total * float(N) * len(q)
    kernel_s = st["kernel_ms"] * 1e-3
    evals_rank = args.steps * F_mine * float(N) * len(q)
    alg = F_mine * (12 * N) + 24 * len(q)
    achieved = alg * args.steps / kernel_s / 1e9 if kernel_s > 0 else 0.0
    # fp64 VALU bound: the kernel's fp64 wave-instructions per 64 terms (4.5 by construction in the
    # register-blocked column kernel; the measured figure and the clock come from profiles/counters.json when
    # it was taken on these sources); a wave64 fp64 instruction takes 4 cycles on a SIMD-32 (half the fp32
    # rate: 78.6 vs 157.3 TFLOP/s)
    sq_ctr = profiled({8: "sq_c3", 32: "sq_default"}.get(args.n_points, ""), *SQ_SOURCES) if N == 32768 else None
    fp64_per_64 = sq_ctr["fp64_per_64_terms"] if sq_ctr else 4.5
    issue = evals_rank / max(kernel_s, 1e-9) / 64.0 * fp64_per_64 * 4.0
    out = {
        "metric": "exp(iq.r) evaluations/sec", "value": evals / dt, "unit": "evals/s",
        "n_gpus": world.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if args.shard_fixed else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("C3" if len(q) == 512 else f"n_points={args.n_points} (reference default grid)"
                                if args.n_points == 32 else f"n_points={args.n_points}")
                               + f" partial S(q) {N} atoms, {len(q)} wavevectors, 2 groups, {F} frames"
                               + ("/job/step (fixed set, sharded)" if args.shard_fixed else "/GPU/step")},
        "frames_per_sec": frames_total / dt,
        "per_rank_frames_per_sec": world.gather(args.steps * F_mine / own),
        **world.describe(),
        "roofline": {"bound": "valu", "unit": "G SIMD issue cycles/s", "achieved": issue / 1e9,
                     "peak": SIMDS * CLOCK_HZ / 1e9, "frac": issue / (SIMDS * CLOCK_HZ),
                     "traffic": sq_ctr["hbm_bytes_per_frame"] * F_mine if sq_ctr and "hbm_bytes_per_frame" in sq_ctr else None,
                     "traffic_source": sq_ctr.get("traffic_source") if sq_ctr else None,
                     "kernel": "sq_rho_quads_kernel (grid wavevectors: separable phase tables in LDS, "
                               "4 columns x 8 m_z accumulators per thread)",
                     "definition": "terms/s / 64 x 4.5 fp64 FMA-class wave-instructions per 64 terms x 4 cycles "
                                   "(wave64 fp64 on a SIMD-32) / (1024 SIMDs x 2.4 GHz nominal); the table fill, "
                                   "address arithmetic and the DVFS clock under fp64 load (~2.07 GHz measured) "
                                   "are what is left",
                     "evaluations_per_sec_kernel": evals_rank / max(kernel_s, 1e-9),
                     "kernel_ms_per_step": st["kernel_ms"] / max(args.steps, 1),
                     "fp64_instructions_per_64_terms": fp64_per_64,
                     "valu_instructions_per_64_terms": sq_ctr.get("valu_per_64_terms") if sq_ctr else None,
                     "clock_hz_under_this_kernel": sq_ctr.get("clock_hz") if sq_ctr else None,
                     "frac_at_measured_clock": issue / (SIMDS * sq_ctr["clock_hz"]) if sq_ctr else None,
                     "counters_source": sq_ctr.get("source") if sq_ctr else None,
                     "hbm": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg}},
        "checksum": float(ssf.sum()),
        "result_digest": [float(x) for x in ssf[:, 1:4].ravel()],
    }
    if world.rank == 0 and world.world == 1 and not args.no_cpu_baseline:
        from oracle import fourier as of
        # bounded sample: whole frames at C3, the first atoms of one frame for larger wavevector sets
        n_f = 2 if len(q) <= 512 else 1
        n_s = N if len(q) <= 512 else max(256, min(N, int(6e7 // len(q))))
        sample = traj.to_host(0, n_f)[:, :n_s].astype(np.float64)
        t0 = time.perf_counter()
        refs = [of.fourier_sum_ref(q, sample[f]) for f in range(n_f)]
        t_cpu = time.perf_counter() - t0
        got = _core.fourier_sum_device(q, sample[0], dev=dev)
        out["cpu_baseline"] = {"value": n_f * float(n_s) * len(q) / t_cpu, "unit": "evals/s", "cores": 1,
                               "kind": "port", "sample": f"{n_f} bench frame(s) x {n_s} atoms, numpy exp(1j q.r) in "
                                                         f"q-chunks (oracle/fourier.py; accelerated.py:81-122), "
                                                         f"{t_cpu:.1f} s",
                               "gpu_max_rel_deviation_on_sample": max_rel_deviation(got, refs[0]),
                               "deviation_metric": "max over wavevectors of |got - ref| / |ref| (element-wise; "
                                                   "entries below 1e-9 of the largest left out)"}
    eng.close()
    traj.free()
    if world.world == 1 and len(q) == 512 and N == 32768 and not args.shard_fixed and not args.no_ingest:
        try:
            t0 = time.perf_counter()
            out["ingest"] = bench_fourier_ingest(args, world, "sq", out["frames_per_sec"])
            out["ingest"]["leg_wall_s"] = time.perf_counter() - t0
        except Exception as exc:            # never takes the resident line down
            out["ingest"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def isf_roofline(N, n_q, F, lagged, kernel_s, per_step, achieved, alg):
    """fp64 VALU bound like S(q).  Model: 2.5 FMA-class wave-instructions per 64 displacement terms (two FMAs per
    term, half a column product), 4.5 per 64 rho terms; a wave64 fp64 instruction takes 4 cycles on a SIMD-32.
    With a profiles/counters.json entry measured on these sources (scripts/make_counters.py isf) the count is the
    one the SQ counters gave (FMA + MUL + ADD f64 of every ISF kernel of a step) and the clock the one the part
    held under this load; traffic from the FETCH_SIZE / WRITE_SIZE passes."""
    model = float(N) * n_q * (lagged * 2.5 + F * 4.5) / 64.0
    ctr = profiled("isf", *ISF_SOURCES) if (N == 32768 and n_q == 512) else None
    ratio = ctr["fp64_ratio_to_model"] if ctr else 1.0
    issue = model * ratio * 4.0 / max(kernel_s, 1e-9)
    out = {"bound": "valu", "unit": "G SIMD issue cycles/s", "achieved": issue / 1e9, "peak": SIMDS * CLOCK_HZ / 1e9,
           "frac": issue / (SIMDS * CLOCK_HZ),
           "traffic": ctr["hbm_bytes_per_frame"] * F if ctr else None,
           "traffic_source": ctr.get("traffic_source") if ctr else None,
           "kernel": "isf_incoherent_quads_kernel + sq_rho_quads_kernel + isf_coherent_kernel",
           "definition": "fp64 wave-instructions of a step (counted: FMA + MUL + ADD f64 of the ISF kernels, "
                         "profiles/counters.json; without an entry the model (displacement terms x 2.5 + rho terms x "
                         "4.5) / 64) x 4 cycles per second of kernel time / (1024 SIMDs x 2.4 GHz nominal)",
           "model_fp64_instructions_per_step": model,
           "fp64_instructions_counted_over_model": ctr["fp64_ratio_to_model"] if ctr else None,
           "clock_hz_under_this_kernel": ctr.get("clock_hz") if ctr else None,
           "frac_at_measured_clock": issue / (SIMDS * ctr["clock_hz"]) if ctr else None,
           "counters_source": ctr.get("source") if ctr else None,
           "evaluations_per_sec_kernel": per_step / max(kernel_s, 1e-9),
           "hbm": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg}}
    return out


def bench_isf(args, world):
    """coherent + incoherent intermediate scattering functions, C3's
    particles and wavevectors, 64 lags; frames resident in HBM (``--host-path``: fed from pageable
    host memory, PCIe inside the timed region)."""
    from mdhelper_amd import _core
    dev = world.dev
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
    dt, _own = timed_region(world, dev, args.steps, step, lambda: None)
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
        "roofline": isf_roofline(N, len(q), F, lagged, kernel_s, per_step, achieved, alg),
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
        # element-wise on the incoherent part (positive, O(1) after normalisation); the coherent part
        # passes through zero, so its deviation is taken against the geometric mean of the two
        # self terms' scale, i.e. the lag-0 coherent value of the same wavevector
        err_i = max_rel_deviation(gi / norm, ref["iisf"], floor=1e-6)
        scale_c = np.maximum(np.abs(ref["cisf"][0:1]).max(axis=1, keepdims=True), 1e-30)
        err_c = float((np.abs(gc / norm - ref["cisf"]) / scale_c).max())
        err = max(err_i, err_c)
        terms = float(n_s) * len(q) * (f_s + sum(min(f + 1, lags_s) for f in range(f_s)))
        out["cpu_baseline"] = {"value": terms / t_cpu, "unit": "evals/s", "cores": 1, "kind": "port",
                               "sample": f"{f_s} frames x {n_s} atoms x {lags_s} lags, numpy restatement "
                                         f"({t_cpu:.1f} s)",
                               "gpu_max_rel_deviation_on_sample": err}
    eng.close()
    traj.free()
    if world.world == 1 and N == 32768 and not args.host_path and not args.no_ingest:
        try:
            t0 = time.perf_counter()
            out["ingest"] = bench_fourier_ingest(args, world, "isf", out["frames_per_sec"])
            out["ingest"]["leg_wall_s"] = time.perf_counter() - t0
        except Exception as exc:
            out["ingest"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def bench_msd(args, world):
    from mdhelper_amd import _core
    from mdhelper_amd.comm import shard_range
    dev = world.dev
    N = args.atoms or 10000
    T = args.frames or 100000
    # particles shard across ranks (transport.py:1036-1039: per-particle MSDs are independent).
    # weak: every rank owns N particles; --shard-fixed: ONE set of N particles, rank r pushes its share
    seed = 4 if args.shard_fixed else 4 + world.rank
    traj = _core.synth_random_walk(T, N, [1.0, 1.0, 1.0], 0.1, seed=seed, dev=dev, dtype=np.float64)
    B = max(1, args.blocks)
    eng = _core.MsdEngine(T // B, B, 2, dev=dev, timing=True)
    groups = [(0, N // 2), (N // 2, N - N // 2)]
    mine = []
    for first, count in groups:
        lo, hi = shard_range(count, world.rank, world.world) if args.shard_fixed else (0, count)
        mine.append((first + lo, hi - lo))
    n_mine = sum(c for _f, c in mine)

    box = {}

    def step():
        # ONE analysis: the groups' particles through the correlation kernels, the accumulators' all-reduce
        # (N > 1), and the result — inverse transforms of the summed power spectra, the S_m recurrence and
        # the D2H of msd / summed trajectories (what Onsager._conclude does per analysis, transport.py:1016-1059)
        eng.reset()
        for g, (first, count) in enumerate(mine):
            if count:
                eng.push_device(g, traj.ptr, N, first, count)
        if world.device_collectives:
            eng.allreduce(world.comm)
        msd, tr = eng.result()
        if world.comm is not None and not world.device_collectives:
            msd, tr = world.reduce_host(msd), world.reduce_host(tr)
        box["msd"], box["traj"] = msd, tr
        # (reset() clears the engine's event timer: the kernels' time is summed step by step, so that the roofline
        # figure covers the same steps as ms_per_step — the first steps after an idle period run at a lower clock)
        k_ms = eng.stats()["kernel_ms"]
        box["kernel_ms"] = box.get("kernel_ms", 0.0) + k_ms
        box["kernel_steps"] = box.get("kernel_steps", 0) + 1
        box.setdefault("kernel_ms_each", []).append(k_ms)

    def finish():
        pass

    for _ in range(args.warmup):     # includes the inverse-transform plan (rocFFT builds it once)
        step()

    box["kernel_ms"], box["kernel_steps"], box["kernel_ms_each"] = 0.0, 0, []
    dt, own = timed_region(world, dev, args.steps, step, finish)
    st = eng.stats()
    st["kernel_ms"] = box["kernel_ms"] / max(box["kernel_steps"], 1)      # average over the timed steps
    # what result() costs on its own (inverse transforms + recurrence + D2H), the pushes already finished
    eng.reset()
    for g, (first, count) in enumerate(mine):
        if count:
            eng.push_device(g, traj.ptr, N, first, count)
    _core.synchronize(dev)
    t_r = time.perf_counter()
    eng.result()
    result_ms = (time.perf_counter() - t_r) * 1e3
    n_total = N if args.shard_fixed else N * world.world
    atom_frames = args.steps * float(n_total) * T
    alg_bytes = 24.0 * n_mine * T
    kernel_s = st["kernel_ms"] * 1e-3
    achieved = alg_bytes / max(kernel_s, 1e-9) / 1e9
    msd = box["msd"][0, 0] / (N // 2 if args.shard_fixed else (N // 2) * world.world)
    own_fft, fft_r1, fft_r2 = eng.transform
    ctr = None
    if N == 10000 and T == 100000 and own_fft and world.world == 1:
        # (entries exist for one block and for the block counts scripts/make_counters.py was asked for)
        ctr = profiled("msd_c4" if B == 1 else f"msd_c4_b{B}", *MSD_SOURCES)
    out = {
        "metric": "MSD atom-frames/sec", "value": atom_frames / dt, "unit": "atom-frames/s",
        "n_gpus": world.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if args.shard_fixed else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"C4 self-MSD {N} atoms" + (" (fixed set, sharded)" if args.shard_fixed else "/GPU")
                               + f" x {T} frames, 2 groups, n_blocks={B}, n_fft={eng.n_fft}"
                               + (f" (own two-pass transform {fft_r1} x {fft_r2})" if own_fft else " (rocFFT)")},
        "per_rank_atom_frames_per_sec": world.gather(args.steps * float(n_mine) * T / own),
        **world.describe(),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": ctr.get("hbm_bytes_per_step") if ctr else None,
                     "traffic_source": ctr.get("traffic_source") if ctr else None,
                     "kernel": "msd pipeline of one step: forward transforms with the per-frame sums fused in + "
                               "power (msd_fft_cols/rows_power kernels for n_fft = 400 x 16..512, 2^13..2^16, "
                               "2^18..2^20, else gather + rocFFT R2C + power)",
                     # what the memory system actually carries: the counters' bytes of a step over the kernels' time
                     # (the half-transformed block is written and read once: 5.2 x the algorithmic bytes)
                     "hbm_throughput_GBps": (ctr["hbm_bytes_per_step"] / max(kernel_s, 1e-9) / 1e9) if ctr else None,
                     "hbm_throughput_frac_of_peak": (ctr["hbm_bytes_per_step"] / max(kernel_s, 1e-9) / 1e9 / HBM_PEAK_GBS)
                     if ctr else None,
                     "kernel_ms_per_step": st["kernel_ms"],
                     "kernel_ms_each_step": [round(x, 3) for x in box["kernel_ms_each"]],
                     # (on the shared hosts a step now and then takes twice as long — the list shows them; the
                     # figure above is the mean the contract asks for, this one the median step)
                     "frac_median_step": alg_bytes / max(float(np.median(box["kernel_ms_each"])) * 1e-3, 1e-9) / 1e9
                     / HBM_PEAK_GBS if box["kernel_ms_each"] else None,
                     "algorithmic_bytes_per_step": alg_bytes,
                     "pipeline_bytes_model": st["bytes_moved"]},
        "step": "reset + push of both groups + result() (inverse transforms, S_m recurrence, D2H): one analysis",
        "result_ms": result_ms,
        "physics_check_msd_over_3sigma2m": float(msd[10] / (3 * 0.01 * 10)),
        "result_digest": [float(x) for x in box["msd"][:, 0, 1:4].ravel()],
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
        err = max_rel_deviation(got[1:], ref[1:], floor=0.0)     # every lag but 0 (pure round-off there)
        out["cpu_baseline"] = {"value": n_s * float(T) / t_cpu, "unit": "atom-frames/s", "cores": 1,
                               "kind": "port",
                               "sample": f"{n_s} particles x {T} frames, scipy-FFT msd_fft restatement "
                                         f"(oracle/correlation.py; correlation.py:461-668), {t_cpu:.1f} s",
                               "gpu_max_rel_deviation_on_sample": err,
                               "deviation_metric": "max over lags 1..T-1 of |got - ref| / |ref| (element-wise)"}
    eng.close()
    traj.free()
    if world.world == 1 and not args.shard_fixed and not getattr(args, "no_onsager", False):
        try:
            t0 = time.perf_counter()
            out["onsager"] = bench_onsager(args, world, out["ms_per_step"])
            out["onsager"]["leg_wall_s"] = time.perf_counter() - t0
        except Exception as exc:            # never takes the engine line down
            out["onsager"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def bench_onsager(args, world, engine_ms):
    """
    BASELINE C4 as a user of the reference runs it (N = 1 only; never the headline value):
    ``Onsager((g0, g1), temperature=1, reduced=True).run()`` — three cross MSDs of the summed trajectories,
    two self MSDs, ``/ 2D`` (reference transport.py:912-1059) — on 10 000 particles x 100 000 frames, fed

    class_hbm_f64         float64 frames resident in HBM (what the engine line reads): class overhead alone
    class_hbm_f32         float32 frames resident in HBM: + gather / widening on the device
    class_host_f32        float32 frames in pageable host memory (what an MDAnalysis memory reader holds):
                          column chunks by 2-D DMA out of the caller's pages, the chunk before transformed meanwhile
    class_host_f32_pinned the same array page-locked through mdx_host_register: one DMA
    class_file            an AMBER NetCDF file in the page cache (FileUniverse): the same column chunks, read by the
                          DMA engine out of the mapped file (`first_analysis_ms`: the first read of the fresh file)

    each with ms per analysis, the ratio of the engine figure (`engine_ms`: reset + pushes + result on HBM-resident
    float64) to it, and the phases of one profiled analysis (marks wait for the device: their sum is a little
    above the unprofiled time); then ``calculate_transport_coefficients`` on the result.
    """
    import tempfile
    import mdhelper_amd
    from mdhelper_amd import _core, _lib
    from mdhelper_amd.analysis import Onsager
    from mdhelper_amd.io import FileUniverse
    dev = world.dev
    N, T, sigma, L = args.atoms or 10000, args.frames or 100000, 0.1, 50.0
    B = max(1, args.blocks)
    dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    charges = np.r_[np.ones(N // 2), -np.ones(N - N // 2)]
    gb32 = 12.0 * N * T / 1e9
    legs = {"atoms": N, "frames": T, "n_blocks": B, "engine_ms_per_analysis": engine_ms,
            "float32_GB": gb32}
    keep = {}

    def analysis(u, profile=False):
        o = Onsager((u.atoms[:N // 2], u.atoms[N // 2:]), temperature=1, reduced=True, n_blocks=B,
                    verbose=False, device=dev)
        o._profile = profile
        return o.run()

    def leg(name, u, reps=3, **more):
        t0 = time.perf_counter()
        analysis(u)                               # warm-up: plans, allocations, pinned ring, page cache
        _core.synchronize(dev)
        first_ms = (time.perf_counter() - t0) * 1e3   # (class_file: the FIRST read of the fresh file — its mapping's
        times = []                                    # page tables are populated and its pages locked for the first time)
        for _ in range(reps):
            t0 = time.perf_counter()
            o = analysis(u)
            times.append((time.perf_counter() - t0) * 1e3)
        ms = float(np.median(times))              # (the host-memory legs share the host's memory system with
        prof = analysis(u, profile=True)          # whatever else runs on the box: single repetitions scatter)
        legs[name] = {"ms_per_analysis": ms, "ms_each": times, "first_analysis_ms": first_ms,
                      "engine_over_class": engine_ms / ms,
                      "phases_ms": {k: v * 1e3 for k, v in prof._timings.items()}, **more}
        keep[name] = o
        return o

    d64 = _core.synth_random_walk(T, N, [1.0, 1.0, 1.0], sigma, seed=4, dev=dev, dtype=np.float64)
    leg("class_hbm_f64", mdhelper_amd.ArrayUniverse.from_device(d64, dims, charges=charges))
    d64.free()
    d32 = _core.synth_random_walk(T, N, [1.0, 1.0, 1.0], sigma, seed=4, wrap=False, dev=dev)
    leg("class_hbm_f32", mdhelper_amd.ArrayUniverse.from_device(d32, dims, charges=charges))
    h = d32.to_host()
    # what the link gives: the same 12 GB by one DMA out of page-locked memory, and through the pinned ring
    _lib.check(_lib.lib().mdx_host_register(dev, h.ctypes.data, h.nbytes))
    try:
        t0 = time.perf_counter()
        _lib.check(_lib.lib().mdx_upload(dev, d32.ptr, h.ctypes.data, h.nbytes))
        legs["h2d_page_locked_GB_per_sec"] = gb32 / (time.perf_counter() - t0)
    finally:
        _lib.check(_lib.lib().mdx_host_unregister(dev, h.ctypes.data))
    t0 = time.perf_counter()
    _lib.check(_lib.lib().mdx_upload(dev, d32.ptr, h.ctypes.data, h.nbytes))
    legs["h2d_pageable_ring_GB_per_sec"] = gb32 / (time.perf_counter() - t0)
    legs["io_threads"] = int(os.environ.get("MDX_IO_THREADS", "8"))
    d32.free()
    um = mdhelper_amd.ArrayUniverse(h, dims, charges=charges)
    leg("class_host_f32", um)
    _lib.check(_lib.lib().mdx_host_register(dev, h.ctypes.data, h.nbytes))
    try:
        leg("class_host_f32_pinned", um)
    finally:
        _lib.check(_lib.lib().mdx_host_unregister(dev, h.ctypes.data))
    tmp = tempfile.NamedTemporaryFile(suffix=".nc", delete=False)
    tmp.close()
    try:
        t0 = time.perf_counter()
        write_amber_netcdf_fast(tmp.name, h, dims)
        legs["file"] = (f"AMBER NetCDF (CDF-2), {os.path.getsize(tmp.name) / 1e9:.2f} GB, written in "
                        f"{time.perf_counter() - t0:.1f} s, read from the page cache")
        fu = FileUniverse(tmp.name, dt=1.0, charges=charges)
        leg("class_file", fu, reps=2)
        fu.trajectory.file.close()
    finally:
        os.unlink(tmp.name)
    del h, um
    # the host legs against what the link gave for the same 12 GB in this process
    # (class_host_f32 and class_file: against the DMA rate out of page-locked memory too — since round 5 their rows
    # are read by the DMA engine where they lie, the ring's copy threads are not on their way)
    for name, key in (("class_host_f32", "h2d_page_locked_GB_per_sec"),
                      ("class_host_f32_pinned", "h2d_page_locked_GB_per_sec"),
                      ("class_file", "h2d_page_locked_GB_per_sec")):
        if name in legs:
            legs[name]["link_bound_ms"] = gb32 / legs[key] * 1e3
            legs[name]["link_bound_over_class"] = legs[name]["link_bound_ms"] / legs[name]["ms_per_analysis"]
    # every leg analysed the same numbers (float64 = the widened float32)
    ref = keep["class_hbm_f64"].results
    for name, o in keep.items():
        legs[name]["max_rel_deviation_from_hbm_f64"] = max(
            max_rel_deviation(o.results.msd_self[:, :, 1:], ref.msd_self[:, :, 1:], floor=0.0),
            float(np.abs(o.results.msd_cross - ref.msd_cross).max() / np.abs(ref.msd_cross).max()))
    # the fit on top (host NumPy / SciPy, transport.py:59-286)
    o = keep["class_hbm_f64"]
    tb = T // B
    t0 = time.perf_counter()
    o.calculate_transport_coefficients(start=1, stop=tb // 10, scale="linear")
    fit_linear = time.perf_counter() - t0
    D_i, L_ij = o.results.D_i.copy(), o.results.L_ij.copy()
    t0 = time.perf_counter()
    o.calculate_transport_coefficients()
    legs["transport_coefficients"] = {
        "fit_linear_window_ms": fit_linear * 1e3, "fit_reference_defaults_ms": (time.perf_counter() - t0) * 1e3,
        "D_i_over_sigma2_over_2dt": [float(x) for x in (D_i / (sigma ** 2 / 2)).ravel()],
        "L_ii_over_L_ii_self_expected": [float(L_ij[b, i, i] / ((N // 2) * sigma ** 2 / 2 / L ** 3))
                                         for b in range(B) for i in range(2)],
        "window": f"lags 1 .. {tb // 10}, linear scale; free walk: D = sigma^2 / 2 dt, L_ii^self = N_i D / (kBT V)"}
    return legs


def bench_fourier_ingest(args, world, kind, resident_fps):
    """
    The drop-in path of S(q) (kind "sq": C3, partial, 512 wavevectors) and of the intermediate scattering function
    (kind "isf": 64 lags, coherent + incoherent) the ways a user of the reference feeds them (N = 1 only; never
    the headline value), each leg with frames/s, its ratio to the HBM-resident figure of this run, and — S(q) at
    C3 consumes 393 KB per frame in 3.4 us, more than the host link delivers — its ratio to what the link gave
    for the same bytes in this process (`h2d_*`: one DMA out of page-locked memory; pageable memory through the
    pinned ring), i.e. how close the pipeline runs to the bound that then applies:

    host          mdx_*_accumulate on pageable host memory (what ``ts.positions`` of a memory reader is)
    host_pinned   the same array page-locked through mdx_host_register
    file          mdx_*_accumulate_traj on an AMBER NetCDF file in the page cache
    class_memory  StructureFactor / IntermediateScatteringFunction (..., mode="partial").run() on an in-memory
                  universe (reference structure.py:1481-1527, 1980-1996), engine creation and result included
    class_file    the same on a FileUniverse
    """
    import tempfile
    import mdhelper_amd
    from mdhelper_amd import _core, _lib
    from mdhelper_amd.analysis import IntermediateScatteringFunction, StructureFactor
    from mdhelper_amd.io import FileUniverse, TrajectoryFile
    dev = world.dev
    N, L = 32768, 68.94
    F = 4000 if kind == "sq" else 512
    n_lags = 64
    box = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    grid = 2 * np.pi * np.arange(8) / L
    q = np.stack(np.meshgrid(grid, grid, grid), -1).reshape(-1, 3)
    sizes, pairs = [N // 2, N - N // 2], ((0, 0), (0, 1), (1, 1))
    d_traj = _core.synth_random_walk(F, N, box[:3], 0.3, seed=2, dev=dev)
    h_traj = d_traj.to_host()
    gb = F * 12.0 * N / 1e9
    legs = {"frames": F, "atoms": N, "wavevectors": len(q), "coordinate_bytes_per_frame": 12 * N,
            "resident_frames_per_sec": resident_fps, "io_threads": int(os.environ.get("MDX_IO_THREADS", "8"))}
    # what the host link gives for these bytes (best of 3)
    def h2d():
        t0 = time.perf_counter()
        _lib.check(_lib.lib().mdx_upload(dev, d_traj.ptr, h_traj.ctypes.data, h_traj.nbytes))
        return gb / (time.perf_counter() - t0)
    legs["h2d_pageable_ring_GB_per_sec"] = max(h2d() for _ in range(3))
    _lib.check(_lib.lib().mdx_host_register(dev, h_traj.ctypes.data, h_traj.nbytes))
    try:
        legs["h2d_page_locked_GB_per_sec"] = max(h2d() for _ in range(3))
    finally:
        _lib.check(_lib.lib().mdx_host_unregister(dev, h_traj.ctypes.data))
    d_traj.free()
    link = {"host": legs["h2d_pageable_ring_GB_per_sec"], "host_pinned": legs["h2d_page_locked_GB_per_sec"]}
    tmp = tempfile.NamedTemporaryFile(suffix=".nc", delete=False)
    tmp.close()
    write_amber_netcdf_fast(tmp.name, h_traj, box)

    def make():
        return (_core.SqEngine(q, sizes, pairs, dev=dev) if kind == "sq"
                else _core.IsfEngine(q, sizes, pairs, n_lags, True, dev=dev))

    def timed(fn, reps=3):
        fn()
        _core.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        _core.synchronize(dev)
        return (time.perf_counter() - t0) / reps

    def leg(name, seconds, result=None, link_gbs=None):
        fps = F / seconds
        legs[name] = {"frames_per_sec": fps, "GB_per_sec": gb / seconds, "ratio_to_resident": fps / resident_fps}
        if link_gbs:
            bound = min(resident_fps, link_gbs * 1e9 / (12.0 * N))       # kernel-bound or link-bound, whichever is lower
            legs[name]["bound_frames_per_sec"] = bound
            legs[name]["ratio_to_bound"] = fps / bound
        if result is not None:
            legs[name]["checksum"] = float(np.sum(result[0] if isinstance(result, tuple) else result))

    try:
        eng = make()

        def host():
            eng.reset()
            eng.accumulate(h_traj)
            box_["r"] = eng.result()
        box_ = {}
        leg("host", timed(host), box_["r"], link["host"])
        _lib.check(_lib.lib().mdx_host_register(dev, h_traj.ctypes.data, h_traj.nbytes))
        try:
            leg("host_pinned", timed(host), box_["r"], link["host_pinned"])
        finally:
            _lib.check(_lib.lib().mdx_host_unregister(dev, h_traj.ctypes.data))
        tf = TrajectoryFile(tmp.name)
        frames = np.arange(F)

        def from_file():
            eng.reset()
            eng.accumulate_traj(tf, frames)
            box_["r"] = eng.result()
        leg("file", timed(from_file), box_["r"])
        eng.close()
        tf.close()
        cls = StructureFactor if kind == "sq" else IntermediateScatteringFunction
        kw = dict(mode="partial", n_points=8, verbose=False, device=dev)      # sort / unique: the reference's defaults
        if kind == "isf":
            kw.update(n_lags=n_lags, incoherent=True)
        res = {}
        um = mdhelper_amd.ArrayUniverse(h_traj, box)

        def cls_mem():
            res["m"] = cls((um.atoms[:N // 2], um.atoms[N // 2:]), **kw).run()
        leg("class_memory", timed(cls_mem, reps=2), link_gbs=link["host"])
        fu = FileUniverse(tmp.name)

        def cls_file():
            res["f"] = cls((fu.atoms[:N // 2], fu.atoms[N // 2:]), **kw).run()
        leg("class_file", timed(cls_file, reps=2))
        fu.trajectory.file.close()
        a = res["m"].results.ssf if kind == "sq" else res["m"].results.cisf
        b = res["f"].results.ssf if kind == "sq" else res["f"].results.cisf
        legs["class_memory_vs_class_file_max_rel_deviation"] = float(np.abs(a - b).max() / np.abs(a).max())
    finally:
        os.unlink(tmp.name)
    return legs


def bench_rdf_ingest(args, world, resident_fps):
    """
    The drop-in path at the kernel's rate (N = 1 only; never the headline value): the same C2(i)
    analysis fed the ways a user of the reference feeds it, each leg with frames/s and its ratio
    to the HBM-resident figure of this run.

    rdf_host          mdx_rdf_accumulate on pageable host memory (what ``ts.positions`` is): library
                      threads copy into the pinned ring, DMA to HBM, kernels overlapped
    rdf_host_pinned   the same buffer page-locked through mdx_host_register (DMA reads it in place)
    rdf_file          mdx_rdf_accumulate_traj on an AMBER NetCDF file in the page cache
    file_to_hbm       the ingest alone (mdx_traj_load_device: pread -> pinned -> HBM -> byte swap), no analysis
    rdf_class_memory  RadialDistributionFunction(u.atoms, exclusion=(1, 1)).run() on an in-memory universe
                      (reference base.py:539-584, structure.py:750-791), engine creation and result included
    rdf_class_file    the same on a FileUniverse over the NetCDF file
    """
    import tempfile
    import mdhelper_amd
    from mdhelper_amd import _core, _lib
    from mdhelper_amd.analysis import RadialDistributionFunction
    from mdhelper_amd.io import FileUniverse, TrajectoryFile
    dev = world.dev
    N, F, L = 32768, 6000, 68.94
    box = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    edges = np.linspace(0.0, 15.0, 202)
    d_traj = _core.synth_random_walk(F, N, box[:3], 0.3, seed=2, dev=dev)
    h_traj = d_traj.to_host()
    d_traj.free()
    h_boxes = np.tile(box, (F, 1))
    tmp = tempfile.NamedTemporaryFile(suffix=".nc", delete=False)
    tmp.close()
    t0 = time.perf_counter()
    write_amber_netcdf_fast(tmp.name, h_traj, box)
    t_write = time.perf_counter() - t0
    gb = F * 12.0 * N / 1e9
    legs = {"frames": F, "atoms": N, "coordinate_bytes_per_frame": 12 * N,
            "resident_frames_per_sec": resident_fps,
            "io_threads": int(os.environ.get("MDX_IO_THREADS", "8")),
            "file": f"AMBER NetCDF (CDF-2), {os.path.getsize(tmp.name) / 1e9:.2f} GB, written in {t_write:.1f} s, "
                    f"read from the page cache"}

    def timed(fn, reps=2):
        fn()                                   # warm-up: allocations, pinned ring, page cache
        _core.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        _core.synchronize(dev)
        return (time.perf_counter() - t0) / reps

    def leg(name, seconds, counts=None, **more):
        legs[name] = {"frames_per_sec": F / seconds, "GB_per_sec": gb / seconds,
                      "ratio_to_resident": F / seconds / resident_fps, **more}
        if counts is not None:
            legs[name]["result_digest"] = hashlib.sha256(
                np.ascontiguousarray(counts, dtype=np.int64).tobytes()).hexdigest()[:16]

    try:
        eng = _core.RdfEngine(edges, (1, 1), dev=dev)

        def host():
            eng.reset()
            eng.accumulate(h_traj, None, h_boxes)
            eng.synchronize()
        leg("rdf_host", timed(host), eng.counts())
        _lib.check(_lib.lib().mdx_host_register(dev, h_traj.ctypes.data, h_traj.nbytes))
        try:
            leg("rdf_host_pinned", timed(host), eng.counts())
        finally:
            _lib.check(_lib.lib().mdx_host_unregister(dev, h_traj.ctypes.data))
        tf = TrajectoryFile(tmp.name)
        frames = np.arange(F)

        def from_file():
            eng.reset()
            eng.accumulate_traj(tf, frames, h_boxes)
            eng.synchronize()
        leg("rdf_file", timed(from_file), eng.counts())
        eng.close()
        d_out = _core.DeviceArray((F, N, 3), np.float32, dev)
        leg("file_to_hbm", timed(lambda: tf.load_device(frames, d_out.ptr, dev=dev)),
            note="ingest alone, analysis kernels off")
        d_out.free()
        tf.close()

        res = {}
        u = mdhelper_amd.ArrayUniverse(h_traj, box)

        def cls_mem():
            res["m"] = RadialDistributionFunction(u.atoms, exclusion=(1, 1), verbose=False, device=dev).run()
        leg("rdf_class_memory", timed(cls_mem), res["m"].results.counts)
        fu = FileUniverse(tmp.name)

        def cls_file():
            res["f"] = RadialDistributionFunction(fu.atoms, exclusion=(1, 1), verbose=False, device=dev).run()
        leg("rdf_class_file", timed(cls_file), res["f"].results.counts)
        fu.trajectory.file.close()
        digests = {legs[k]["result_digest"] for k in legs if isinstance(legs[k], dict) and "result_digest" in legs[k]}
        legs["all_legs_same_counts"] = len(digests) == 1
        # what one host has to deliver for the 8 ranks of C2 / C5 at this kernel rate
        legs["host_demand_8_ranks_GB_per_sec"] = {
            "C2": 8 * resident_fps * 12.0 * N / 1e9,
            "note": "coordinates only; pageable memory costs one extra host copy (read + write) per byte, "
                    "pinned or registered memory and files in the page cache one read"}
    finally:
        os.unlink(tmp.name)
    return legs


def run_extras(args, world):
    """Short C3 / C4 / C2(ii) legs after the headline one, embedded in the same JSON line so that the
    S(q) and MSD paths (accelerated.py:81-165, correlation.py:461-668) are in the driver-timed record
    too.  Each is a complete line of its own workload (value, roofline, cpu_baseline)."""
    import copy
    extra = {}
    # (ten and twelve steps: a step is 3.5 / 27 ms, and the first few after another workload run at a lower clock)
    plan = (("sq", dict(workload="sq", frames=1000, steps=10, warmup=2)),
            ("msd", dict(workload="msd", frames=None, steps=12, warmup=5)),
            ("rdf_wide", dict(workload="rdf_wide", frames=1000, steps=3, warmup=1)))
    for name, over in plan:
        a = copy.copy(args)
        a.atoms, a.blocks, a.algo, a.host_path, a.traj_file, a.shard_fixed = None, 1, "auto", False, False, False
        a.cpu_seconds = min(args.cpu_seconds, 5.0)
        for k, v in over.items():
            setattr(a, k, v)
        t0 = time.perf_counter()
        try:
            if name == "sq":
                extra[name] = bench_sq(a, world)
            elif name == "msd":
                extra[name] = bench_msd(a, world)
            else:
                extra[name] = bench_rdf(a, world, wide=True)
            extra[name]["leg_wall_s"] = time.perf_counter() - t0
        except Exception as exc:            # an extra leg never takes the headline line down
            extra[name] = {"error": f"{type(exc).__name__}: {exc}"}
    return extra


def run_ingest(args, world, resident_fps):
    t0 = time.perf_counter()
    try:
        out = bench_rdf_ingest(args, world, resident_fps)
        out["leg_wall_s"] = time.perf_counter() - t0
        return out
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}"}


def dry_run(args):
    """One rank of ``bench.py --gpus N --dry-run``: everything an N-rank run does around the kernels, with the
    engines left out and no HIP call — rendezvous (rank 0 hosts it), the broadcast of a 128-byte id (what
    ncclGetUniqueId's travels as), the shard plan of the workload (frames for rdf / sq / isf, particles for msd:
    comm.shard_range, the plan the engines receive), one host all-reduce of per-rank work counts and a barrier.
    What it leaves untested on a real node is RCCL itself."""
    from mdhelper_amd.comm import shard_range
    from mdhelper_amd.launch import Rendezvous
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    from mdhelper_amd import _lib
    rt = _lib.runtime_summary()          # (dlopen + version queries: no device is touched)
    rdzv = Rendezvous(rank, world) if world > 1 else None
    token = bytes((7 * i + 1) & 255 for i in range(128))
    got = rdzv.bcast(token if rank == 0 else None) if rdzv else token
    if args.workload == "msd":
        n, unit = args.atoms or 10000, "particles"
        groups = [(0, n // 2), (n // 2, n - n // 2)]
        mine = [[first + lo, first + hi] for first, count in groups
                for lo, hi in [shard_range(count, rank, world) if args.shard_fixed else (0, count)]]
        work = sum(hi - lo for lo, hi in mine)
        total = n if args.shard_fixed else n * world
    else:
        n, unit = args.frames or (1000 if args.workload in ("sq", "isf", "rdf_wide") else 10000), "frames"
        lo, hi = shard_range(n, rank, world) if args.shard_fixed else (0, n)
        mine, work = [[lo, hi]], hi - lo
        total = n if args.shard_fixed else n * world
    plans = [None] * world
    if rdzv:
        blob = rdzv.gather((json.dumps({"rank": rank, "device": local, unit: mine, "librccl": rt["librccl"],
                                        "rccl_version": rt["rccl_version"], "one_runtime": rt["one_runtime"]})
                            + "\n").encode()).decode()
        plans = [json.loads(x) for x in blob.splitlines()]
        summed = int(rdzv.allreduce(np.array([work], dtype=np.int64))[0])
        rdzv.barrier()
        rdzv.close()
    else:
        plans, summed = [{"rank": 0, "device": 0, unit: mine, "librccl": rt["librccl"],
                          "rccl_version": rt["rccl_version"], "one_runtime": rt["one_runtime"]}], work
    if rank == 0:
        print(json.dumps({"dry_run": True, "workload": args.workload, "n_gpus": world,
                          "scaling": "strong" if args.shard_fixed else "weak",
                          "id_broadcast_ok": got == token, "ranks_in_plan": sorted(p["rank"] for p in plans),
                          "devices": [p["device"] for p in plans], "plan": plans, "unit": unit,
                          "work_all_ranks": summed, "work_expected": total, "plan_covers_the_work": summed == total,
                          "runtime": rt, "one_runtime_on_every_rank": all(p.get("one_runtime") for p in plans),
                          "rccl": False, "comm": "not started (dry run: control plane only)"}), flush=True)
    return 0


def launch_ranks(args):
    """``python bench.py --gpus N`` outside any launcher: start N fresh ranks of this script (one per
    GPU) from this process, which never touches the GPU, and relay rank 0's line."""
    from mdhelper_amd import launch
    try:
        rc, text = launch.launch(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:],
                                 share_devices=args.share_devices, check_devices=not args.dry_run)
    except RuntimeError as exc:
        sys.stderr.write(f"bench.py: {exc}\n")
        return 2
    line = launch.last_json_line(text)
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        rc = 1
    if line is not None:
        line["launcher"] = "bench.py (mdhelper_amd.launch: one fresh process per GPU)"
        print(json.dumps(line), flush=True)
    return rc


def runtime_record(world):
    """The user-mode ROCm stack libmdx.so ran on in this process (`mdhelper_amd._lib.runtime`: the files its HIP /
    rocFFT / RCCL calls are bound to, versions, `one_runtime`) — rank 0's, plus whether every rank reported the
    same (one number per rank through the communicator that is there anyway)."""
    import zlib
    from mdhelper_amd import _lib
    rec = _lib.runtime_summary()
    mine = float(zlib.crc32(json.dumps(rec, sort_keys=True).encode()))
    rec["same_on_all_ranks"] = all(v == mine for v in world.gather(mine))
    return rec


def configs_summary(extra):
    """BASELINE configs[2], [3] and C2(ii) in one compact object (<= 600 characters)."""
    def brief(line, **more):
        if not isinstance(line, dict) or "value" not in line:
            return {"error": str((line or {}).get("error", "missing"))[:80]}
        out = {"value": float(f"{line['value']:.4g}"), "unit": line.get("unit"),
               "ms_per_step": round(line.get("ms_per_step", 0.0), 3),
               "frac": round((line.get("roofline") or {}).get("frac") or 0.0, 4),
               "bound": (line.get("roofline") or {}).get("bound"),
               "cpu_value": float(f"{(line.get('cpu_baseline') or {}).get('value') or 0.0:.4g}")}
        out.update(more)
        return out
    msd = extra.get("msd") or {}
    ons = (msd.get("onsager") or {}) if isinstance(msd, dict) else {}
    more = {}
    for key in ("class_hbm_f64", "class_hbm_f32", "class_host_f32", "class_host_f32_pinned", "class_file"):
        leg = ons.get(key)
        if isinstance(leg, dict) and "ms_per_analysis" in leg:
            more[key + "_ms"] = round(leg["ms_per_analysis"], 1)
            if "link_bound_ms" in leg:
                more[key + "_link_ms"] = round(leg["link_bound_ms"], 1)
            if key == "class_file" and "first_analysis_ms" in leg:
                more["class_file_first_read_ms"] = round(leg["first_analysis_ms"], 1)
    return {"C3": brief(extra.get("sq")), "C4": brief(msd, **more), "C2ii": brief(extra.get("rdf_wide"))}


def product_library():
    """bench.py measures the product library and nothing else: MDX_LIBRARY (mdhelper_amd/_lib.py) may
    not point at a stand-in."""
    from mdhelper_amd import _lib
    want = os.path.join(ROOT, "mdhelper_amd", "libmdx.so")
    if os.path.realpath(str(_lib.LIB_PATH)) != os.path.realpath(want):
        raise SystemExit(f"bench.py: MDX_LIBRARY points at {_lib.LIB_PATH}; the bench runs {want} only")
    return os.path.relpath(want, ROOT)


def main():
    args = parse()
    library = product_library()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    if args.dry_run:
        sys.exit(dry_run(args))
    # Libraries underneath (RCCL) print banners on stdout; keep stdout clean for the
    # one JSON line by pointing fd 1 at stderr until the result is ready.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    world = World(args)
    if args.workload == "ingest":
        # the host-buffer / file / operator-surface legs on their own, behind a short resident run
        args.frames, args.no_cpu_baseline = args.frames or 3000, True
        out = bench_rdf(args, world)
        out["extra"] = {"ingest": run_ingest(args, world, out["frames_per_sec"])}
    elif args.workload in ("rdf", "rdf_wide"):
        out = bench_rdf(args, world, wide=args.workload == "rdf_wide")
    elif args.workload == "sq":
        out = bench_sq(args, world)
    elif args.workload == "isf":
        out = bench_isf(args, world)
    else:
        out = bench_msd(args, world)
    plain = not (args.host_path or args.traj_file or args.shard_fixed or args.atoms or args.algo != "auto")
    if args.workload == "rdf" and world.world == 1 and plain and not args.no_extras:
        out["extra"] = run_extras(args, world)
        out["extra"]["ingest"] = run_ingest(args, world, out["frames_per_sec"])
    out["library"] = library
    out["runtime"] = runtime_record(world)
    if "extra" in out:
        # LAST key of the line (the driver keeps a 2 000-character tail): C3 / C4 / C2(ii) in a few numbers each
        out["configs"] = configs_summary(out["extra"])
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if world.rank == 0:
        print(json.dumps(out), flush=True)
    os.dup2(2, 1)          # teardown chatter goes to stderr as well
    world.close()


if __name__ == "__main__":
    main()
