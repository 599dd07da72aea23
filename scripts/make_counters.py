#!/usr/bin/env python3
"""
Run ON THE GPU BOX (through gpurun): SQ / memory counters of the bench kernels from separate rocprofv3 PMC
passes (counters only with --kernel-trace, as MI355X_MICROARCH.md prescribes), reduced to the per-unit
figures bench.py reads back for its roofline objects:

    gpurun_out/counters/counters.json   ->  copy to profiles/counters.json
    gpurun_out/counters/*_kernel_stats.csv, *_pmc.csv  -> copy to profiles/

Each entry carries the sha256 of the kernel sources it was measured on (bench.source_digest); bench.py
ignores an entry whose digest differs from the sources it runs.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (source_digest only; nothing touches the GPU in this process)

ROUND = os.environ.get("MDX_ROUND", "r04")
OUT = os.path.join(ROOT, "gpurun_out", "counters")
os.makedirs(OUT, exist_ok=True)
ENV = dict(os.environ, TMPDIR="/tmp")


def run_pmc(tag, counters, bench_args):
    d = os.path.join(OUT, f"{tag}_{'_'.join(counters)[:40]}")
    cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", tag,
           "--", sys.executable, os.path.join(ROOT, "bench.py"), *bench_args]
    p = subprocess.run(cmd, cwd="/tmp", env=ENV, capture_output=True, text=True, timeout=900)
    if p.returncode != 0:
        sys.stderr.write(p.stderr[-2000:])
        raise SystemExit(f"{tag}: rocprofv3 failed")
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    res = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                res[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[row["Kernel_Name"]].add(row["Dispatch_Id"])
    for path in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                dur[row["Kernel_Name"]] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    with open(os.path.join(OUT, f"{tag}_{'_'.join(counters)[:40]}_pmc.csv"), "w") as fh:
        fh.write("kernel,dispatches,duration_ns," + ",".join(counters) + "\n")
        for k in sorted(res, key=lambda k: -dur[k]):
            fh.write(f"\"{k[:100]}\",{len(calls[k])},{dur[k]:.0f}," + ",".join(f"{res[k][c]:.0f}" for c in counters) + "\n")
    return json.loads(line), res, calls, dur


def run_stats(tag, bench_args):
    d = os.path.join(OUT, f"{tag}_stats")
    cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", tag,
           "--", sys.executable, os.path.join(ROOT, "bench.py"), *bench_args]
    p = subprocess.run(cmd, cwd="/tmp", env=ENV, capture_output=True, text=True, timeout=900)
    if p.returncode != 0:
        sys.stderr.write(p.stderr[-2000:])
        raise SystemExit(f"{tag}: rocprofv3 --stats failed")
    for path in glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True):
        with open(path) as src, open(os.path.join(OUT, f"{tag}_kernel_stats.csv"), "w") as dst:
            dst.write(src.read())
    # the profiler's averages take in the cold first launch and the one step during which it flushes its buffers
    # (a step of twice the time in every profiled run): medians of the same trace beside them
    import statistics
    per = collections.defaultdict(list)
    for path in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                per[row["Kernel_Name"]].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    with open(os.path.join(OUT, f"{tag}_kernel_medians.csv"), "w") as fh:
        fh.write("kernel,calls,median_ns,min_ns,mean_ns,max_ns\n")
        for k in sorted(per, key=lambda k: -sum(per[k])):
            v = per[k]
            fh.write(f"\"{k[:110]}\",{len(v)},{statistics.median(v):.0f},{min(v):.0f},{sum(v) / len(v):.0f},{max(v):.0f}\n")
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    with open(os.path.join(OUT, f"{tag}_bench_under_rocprof.json"), "w") as fh:
        fh.write(line + "\n")
    return json.loads(line)


def short(name):
    """The kernel's own identifier out of a demangled name (templates and argument lists dropped)."""
    import re
    m = re.search(r"(\w*kernel\w*)", name)
    return m.group(1) if m else name[:60]


def pick(res, needle):
    for k in res:
        if needle in k:
            return k
    raise SystemExit(f"no kernel matching {needle}")


def rdf_entry(tag, workload, frames, atoms=None):
    """SQ instruction mix per hot-loop trip (64 distance evaluations), clock, LDS figures and HBM traffic of
    the RDF kernels of one bench configuration.  The dominant kernel is the cell-sorted pair kernel, or the
    brute-force tile kernel for systems below 1 024 particles (C1-like)."""
    args = ["--workload", workload, "--frames", str(frames), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
            "--no-extras"] + (["--atoms", str(atoms)] if atoms else [])
    out = {}
    line, res, calls, dur = run_pmc(tag, ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH"], args)
    needle = "rdf_cell_pair_kernel" if any("rdf_cell_pair_kernel" in k for k in res) else "rdf_tile_kernel"
    k = pick(res, needle)
    evals = line["roofline"]["work"]["distance_evaluations_per_sec_kernel"] * line["roofline"]["kernel_ms_per_launch"] \
        * 1e-3 * max(1, line["roofline"].get("launches", 1))
    steps = evals / 64.0
    valu = res[k]["SQ_INSTS_VALU"] / steps
    out.update(kernel=needle, valu_per_step=valu,
               salu_per_step=res[k]["SQ_INSTS_SALU"] / steps, lds_per_step=res[k]["SQ_INSTS_LDS"] / steps,
               branch_per_step=res[k]["SQ_INSTS_BRANCH"] / steps, steps_per_frame=steps / frames)
    line, res, calls, dur = run_pmc(tag, ["GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_INSTS_VALU_TRANS_F32", "SQ_WAVE_CYCLES"], args)
    k = pick(res, needle)
    trans = res[k]["SQ_INSTS_VALU_TRANS_F32"] / steps
    out.update(clock_hz=res[k]["GRBM_GUI_ACTIVE"] / 8.0 / dur[k] * 1e9,          # summed over the 8 XCDs
               waves_per_simd=res[k]["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * res[k]["GRBM_GUI_ACTIVE"] / 8.0) / 4.0 * 4.0,
               valu_trans_per_step=trans, valu_plain_per_step=valu - trans)
    line, res, calls, dur = run_pmc(tag, ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS",
                                          "SQ_WAIT_INST_LDS"], args)
    k = pick(res, needle)
    out.update(lds_bank_conflict_cycles_per_step=res[k]["SQ_LDS_BANK_CONFLICT"] / steps,
               lds_idx_active_cycles_per_step=res[k]["SQ_LDS_IDX_ACTIVE"] / steps,
               lds_wait_quadcycles_per_step=res[k]["SQ_WAIT_INST_LDS"] / steps)
    # HBM traffic of EVERY RDF kernel of the step (sort / pack + pair), per frame
    per_kernel = collections.defaultdict(float)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        line, res, calls, dur = run_pmc(tag, [ctr], args)
        for kk in res:
            if "rdf_" in kk:
                # gfx950: FETCH_SIZE reports half of a wide coalesced read stream -> doubled; WRITE_SIZE exact
                per_kernel[kk.split("(")[0][-40:]] += res[kk][ctr] * 1024.0 * (2.0 if ctr == "FETCH_SIZE" else 1.0) / frames
    out.update(hbm_bytes_per_frame=sum(per_kernel.values()),
               hbm_bytes_per_frame_by_kernel=dict(per_kernel),
               traffic_source=f"profiles/{ROUND}_{tag}_FETCH_SIZE_pmc.csv + {ROUND}_{tag}_WRITE_SIZE_pmc.csv "
                              f"(separate passes, FETCH doubled per MI355X_MICROARCH.md; sort + pair kernels)",
               source=f"profiles/{ROUND}_{tag}_SQ_INSTS_VALU_SQ_INSTS_SALU_SQ_INSTS_LDS_pmc.csv, "
                      f"{frames} frames, scripts/make_counters.py",
               atoms=atoms or 32768,
               source_digest=bench.source_digest(*bench.RDF_SOURCES))
    return out


def rdf_requests(tag, workload, frames, atoms=None):
    """Bytes the RDF kernels move between L2 and the fabric, from the REQUEST counters (sizes known per request:
    no correction factor): a cross-check of the FETCH_SIZE / WRITE_SIZE figures, whose prescribed doubling of
    FETCH_SIZE is calibrated on wide coalesced streams — not what a cell-sorted gather looks like."""
    args = ["--workload", workload, "--frames", str(frames), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
            "--no-extras"] + (["--atoms", str(atoms)] if atoms else [])
    out = collections.defaultdict(dict)
    line, res, calls, dur = run_pmc(tag + "_req", ["TCC_EA0_RDREQ", "TCC_EA0_RDREQ_32B", "TCC_EA0_RDREQ_64B",
                                                   "TCC_EA0_RDREQ_128B"], args)
    for k in res:
        if "rdf_" in k:
            r = res[k]
            other = r["TCC_EA0_RDREQ"] - r["TCC_EA0_RDREQ_32B"] - r["TCC_EA0_RDREQ_64B"] - r["TCC_EA0_RDREQ_128B"]
            out[k.split("(")[0][-40:]]["read_bytes_per_frame"] = (
                32.0 * r["TCC_EA0_RDREQ_32B"] + 64.0 * (r["TCC_EA0_RDREQ_64B"] + max(other, 0.0))
                + 128.0 * r["TCC_EA0_RDREQ_128B"]) / frames
            out[k.split("(")[0][-40:]]["read_requests_per_frame"] = r["TCC_EA0_RDREQ"] / frames
    line, res, calls, dur = run_pmc(tag + "_req", ["TCC_EA0_WRREQ", "TCC_EA0_WRREQ_64B"], args)
    for k in res:
        if "rdf_" in k:
            r = res[k]
            out[k.split("(")[0][-40:]]["write_bytes_per_frame"] = (
                64.0 * r["TCC_EA0_WRREQ_64B"] + 32.0 * (r["TCC_EA0_WRREQ"] - r["TCC_EA0_WRREQ_64B"])) / frames
            out[k.split("(")[0][-40:]]["write_requests_per_frame"] = r["TCC_EA0_WRREQ"] / frames
    total = sum(v.get("read_bytes_per_frame", 0.0) + v.get("write_bytes_per_frame", 0.0) for v in out.values())
    return dict(by_kernel=dict(out), bytes_per_frame=total,
                source=f"profiles/{ROUND}_{tag}_req_TCC_EA0_*_pmc.csv: 32 / 64 / 128-byte read requests, 64-byte and "
                       f"partial (counted as 32-byte) write requests at the L2 - fabric interface")


def traffic_passes(tag, args, match, per):
    """HBM bytes of the kernels whose name contains one of `match`, from separate FETCH_SIZE / WRITE_SIZE passes
    (FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950), divided by `per`."""
    total, by_kernel = 0.0, {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        line, res, calls, dur = run_pmc(tag, [ctr], args)
        for k in res:
            if any(m in k for m in match):
                v = res[k][ctr] * 1024.0 * (2.0 if ctr == "FETCH_SIZE" else 1.0) / per
                total += v
                by_kernel[short(k) + ":" + ctr] = by_kernel.get(short(k) + ":" + ctr, 0.0) + v
    return total, by_kernel, (f"profiles/{ROUND}_{tag}_FETCH_SIZE_pmc.csv + {ROUND}_{tag}_WRITE_SIZE_pmc.csv (separate "
                              f"passes, FETCH doubled per MI355X_MICROARCH.md)")


def sq_entry(tag, n_points=8, frames=1000):
    """fp64 instruction mix of the S(q) kernel per 64 phase terms, the clock it ran at, its HBM traffic."""
    steps = 2
    args = ["--workload", "sq", "--steps", str(steps), "--warmup", "0", "--no-cpu-baseline", "--no-ingest",
            "--n-points", str(n_points), "--frames", str(frames)]
    line, res, calls, dur = run_pmc(tag, ["SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64",
                                          "SQ_INSTS_VALU_ADD_F64"], args)
    k = pick(res, "sq_rho_quads_kernel")
    terms64 = line["roofline"]["evaluations_per_sec_kernel"] * line["roofline"]["kernel_ms_per_step"] * 1e-3 \
        * line["steps"] / 64.0
    out = dict(valu_per_64_terms=res[k]["SQ_INSTS_VALU"] / terms64,
               fp64_per_64_terms=(res[k]["SQ_INSTS_VALU_FMA_F64"] + res[k]["SQ_INSTS_VALU_MUL_F64"]
                                  + res[k]["SQ_INSTS_VALU_ADD_F64"]) / terms64,
               kernel_share_of_step=dur[k] / max(sum(dur.values()), 1.0), n_points=n_points, frames=frames)
    line, res, calls, dur = run_pmc(tag, ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS"], args)
    k = pick(res, "sq_rho_quads_kernel")
    out.update(clock_hz=res[k]["GRBM_GUI_ACTIVE"] / 8.0 / dur[k] * 1e9,
               lds_per_64_terms=res[k]["SQ_INSTS_LDS"] / terms64,
               raw_pass2={c: res[k][c] for c in ("GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU")},
               source=f"profiles/{ROUND}_{tag}_SQ_INSTS_VALU_SQ_INSTS_VALU_FMA_F64_SQ_INST_pmc.csv, scripts/make_counters.py",
               source_digest=bench.source_digest(*bench.SQ_SOURCES))
    # LDS side: cycles the LDS array is busy and what bank conflicts add, per 64 terms (LDS-array cycles, per CU)
    line, res, calls, dur = run_pmc(tag, ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS",
                                          "SQ_WAVE_CYCLES"], args)
    k = pick(res, "sq_rho_quads_kernel")
    out.update(lds_idx_active_per_64_terms=res[k]["SQ_LDS_IDX_ACTIVE"] / terms64,
               lds_bank_conflict_per_64_terms=res[k]["SQ_LDS_BANK_CONFLICT"] / terms64,
               lds_wait_share_of_wave_cycles=res[k]["SQ_WAIT_INST_LDS"] / max(res[k]["SQ_WAVE_CYCLES"], 1.0))
    total, by_kernel, src = traffic_passes(tag, args, ("sq_",), steps * frames)
    out.update(hbm_bytes_per_frame=total, hbm_bytes_per_frame_by_kernel=by_kernel, traffic_source=src)
    return out


ISF_KERNELS = ("isf_incoherent", "sq_rho", "isf_coherent", "isf_reduce", "isf_rho_merge")


def _sum_by(keys, value):
    out = {}
    for k in keys:
        out[short(k)] = out.get(short(k), 0.0) + value(k)
    return out


def isf_entry(tag):
    """The ISF step (32 768 particles, 512 wavevectors, 64 lags, 256 frames): fp64 instructions actually issued by its
    kernels against the model bench.py prices (2.5 per 64 displacement terms + 4.5 per 64 rho terms), the clock,
    the HBM traffic."""
    frames = 256
    args = ["--workload", "isf", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-ingest",
            "--frames", str(frames)]
    line, res, calls, dur = run_pmc(tag, ["SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64",
                                          "SQ_INSTS_VALU_ADD_F64"], args)
    ks = [k for k in res if any(m in k for m in ISF_KERNELS)]
    fp64 = sum(res[k]["SQ_INSTS_VALU_FMA_F64"] + res[k]["SQ_INSTS_VALU_MUL_F64"] + res[k]["SQ_INSTS_VALU_ADD_F64"]
               for k in ks)
    valu = sum(res[k]["SQ_INSTS_VALU"] for k in ks)
    model = line["roofline"]["model_fp64_instructions_per_step"]
    out = dict(fp64_instructions_per_step=fp64, valu_instructions_per_step=valu, model_fp64_instructions_per_step=model,
               fp64_ratio_to_model=fp64 / model, frames=frames,
               per_kernel_fp64=_sum_by(ks, lambda k: res[k]["SQ_INSTS_VALU_FMA_F64"] + res[k]["SQ_INSTS_VALU_MUL_F64"]
                                       + res[k]["SQ_INSTS_VALU_ADD_F64"]))
    line, res, calls, dur = run_pmc(tag, ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS"], args)
    ks = [k for k in res if any(m in k for m in ISF_KERNELS)]
    big = max(ks, key=lambda k: dur[k])
    out.update(clock_hz=res[big]["GRBM_GUI_ACTIVE"] / 8.0 / dur[big] * 1e9, clock_kernel=short(big),
               kernel_ns=_sum_by(ks, lambda k: dur[k]),
               source=f"profiles/{ROUND}_{tag}_SQ_INSTS_VALU_SQ_INSTS_VALU_FMA_F64_SQ_INST_pmc.csv, scripts/make_counters.py",
               source_digest=bench.source_digest(*bench.ISF_SOURCES))
    total, by_kernel, src = traffic_passes(tag, args, ISF_KERNELS, frames)
    out.update(hbm_bytes_per_frame=total, hbm_bytes_per_frame_by_kernel=by_kernel, traffic_source=src)
    return out


# What the MSD passes ask of the memory system (a counter that names the cause): request sizes
# and stalls at the L2 <-> fabric interface, L2 hit rate, queue depths.  Separate passes; per kernel sums.
MSD_TCC_PASSES = (
    ["TCC_EA0_RDREQ", "TCC_EA0_RDREQ_32B", "TCC_EA0_RDREQ_64B", "TCC_EA0_RDREQ_128B"],
    ["TCC_EA0_WRREQ", "TCC_EA0_WRREQ_64B", "TCC_EA0_WRREQ_STALL", "TCC_TOO_MANY_EA_WRREQS_STALL"],
    ["TCC_EA0_RDREQ_DRAM_CREDIT_STALL", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL", "TCC_EA0_RDREQ_LEVEL", "TCC_EA0_WRREQ_LEVEL"],
    ["TCC_HIT", "TCC_MISS", "TCC_TAG_STALL", "TCC_BUSY"],
    ["TCC_CYCLE", "TCC_REQ", "TCP_PENDING_STALL_CYCLES", "TCP_TCR_TCP_STALL_CYCLES"],
)


def msd_tcc(tag, blocks=1):
    args = ["--workload", "msd", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-onsager",
            "--blocks", str(blocks)]
    table = collections.defaultdict(dict)
    for ctrs in MSD_TCC_PASSES:
        try:
            line, res, calls, dur = run_pmc(tag + "_tcc", ctrs, args)
        except SystemExit as exc:          # a counter this build of rocprofv3 refuses: keep the other passes
            sys.stderr.write(f"{ctrs}: {exc}\n")
            continue
        for k in res:
            if "msd" in k:
                name = short(k)
                table[name]["dispatches"] = len(calls[k])
                table[name]["duration_ms"] = dur[k] / 1e6
                for c in ctrs:
                    table[name][c] = res[k][c]
    with open(os.path.join(OUT, f"{tag}_tcc_summary.json"), "w") as fh:
        json.dump(table, fh, indent=1, sort_keys=True)
    return table


def msd_entry(tag, blocks=1):
    args = ["--workload", "msd", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-onsager",
            "--blocks", str(blocks)]
    # warm-up + timed step + the extra result() pass of bench_msd: three identical passes of the push kernels
    total, per_kernel, src = traffic_passes(tag, args, ("msd",), 3.0)
    return dict(hbm_bytes_per_step=total, per_kernel_bytes_per_step=per_kernel, traffic_source=src, n_blocks=blocks,
                source_digest=bench.source_digest(*bench.MSD_SOURCES))


def main():
    which = sys.argv[1:] or ["rdf_c2", "rdf_wide", "rdf_c5", "rdf_c1", "rdf_req", "sq_c3", "sq_default", "isf", "msd_c4",
                             "msd_c4_b8", "msd_tcc", "stats"]
    path = os.path.join(OUT, "counters.json")
    data = {}
    if os.path.exists(os.path.join(ROOT, "profiles", "counters.json")):
        with open(os.path.join(ROOT, "profiles", "counters.json")) as fh:
            data = json.load(fh)
    if os.path.exists(path):                       # entries of an earlier call of this script on this box
        with open(path) as fh:
            data.update(json.load(fh))
    if "rdf_c2" in which:
        data["rdf_c2"] = rdf_entry("rdf_c2", "rdf", 2000)
    if "rdf_wide" in which:
        data["rdf_wide"] = rdf_entry("rdf_wide", "rdf_wide", 400)
    if "rdf_c5" in which:                          # C5 size on one GPU
        data["rdf_c5"] = rdf_entry("rdf_c5", "rdf", 250, atoms=131072)
    if "rdf_c1" in which:                          # C1-like: 1 000 atoms, range (0, L/2)
        data["rdf_c1"] = rdf_entry("rdf_c1", "rdf_wide", 20000, atoms=1000)
    if "rdf_req" in which and "rdf_c2" in data:   # request-level cross-check of the C2(i) traffic figure
        data["rdf_c2"]["hbm_requests"] = rdf_requests("rdf_c2", "rdf", 2000)
    if "sq_c3" in which:
        data["sq_c3"] = sq_entry("sq_c3")
    if "sq_default" in which:                     # the reference's default grid: n_points = 32, 32 768 wavevectors
        data["sq_default"] = sq_entry("sq_default", n_points=32, frames=100)
    if "isf" in which:
        data["isf"] = isf_entry("isf")
    if "msd_c4" in which:
        data["msd_c4"] = msd_entry("msd_c4")
    for name in which:                             # msd_c4_b<N>: C4 in N blocks (8 is the quoted one)
        if name.startswith("msd_c4_b") and name[8:].isdigit():
            data[name] = msd_entry(name, blocks=int(name[8:]))
    with open(path, "w") as fh:
        json.dump(data, fh, indent=1, sort_keys=True)
    if "msd_tcc" in which:
        msd_tcc("msd_c4")
        msd_tcc("msd_c4_b8", blocks=8)
    if "stats_msd" in which:
        # the MSD lines alone (after a change to the transforms), plus one block count for each new row kernel
        run_stats("msd_c4", ["--workload", "msd", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-onsager"])
        run_stats("msd_c4_b8", ["--workload", "msd", "--blocks", "8", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-onsager"])
        run_stats("msd_c4_b2", ["--workload", "msd", "--blocks", "2", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-onsager"])
        run_stats("msd_c4_b4", ["--workload", "msd", "--blocks", "4", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-onsager"])
        run_stats("msd_c4_b16", ["--workload", "msd", "--blocks", "16", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-onsager"])
        run_stats("msd_c4_b32", ["--workload", "msd", "--blocks", "32", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-onsager"])
    if "stats" in which:
        # the default command itself (what the driver runs): its kernel averages must agree with the line's own
        run_stats("bench_default", ["--no-extras", "--cpu-seconds", "2"])
        run_stats("rdf_c2", ["--frames", "2000", "--steps", "4", "--no-cpu-baseline", "--no-extras"])
        run_stats("rdf_wide", ["--workload", "rdf_wide", "--frames", "500", "--steps", "2", "--no-cpu-baseline"])
        run_stats("rdf_c5", ["--atoms", "131072", "--frames", "500", "--steps", "2", "--no-cpu-baseline", "--no-extras"])
        run_stats("rdf_c1", ["--workload", "rdf_wide", "--atoms", "1000", "--frames", "20000", "--steps", "2",
                             "--no-cpu-baseline", "--no-extras"])
        run_stats("msd_c4", ["--workload", "msd", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-onsager"])
        run_stats("msd_c4_b8", ["--workload", "msd", "--blocks", "8", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-onsager"])
        run_stats("sq_c3", ["--workload", "sq", "--steps", "5", "--no-cpu-baseline", "--no-ingest"])
        run_stats("sq_default_grid", ["--workload", "sq", "--n-points", "32", "--frames", "200", "--steps", "3",
                                      "--no-cpu-baseline", "--no-ingest"])
        run_stats("isf", ["--workload", "isf", "--steps", "2", "--no-cpu-baseline", "--no-ingest"])
    print(json.dumps(data, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
