"""
ctypes binding of ``libmdx.so`` (C-ABI declared in ``include/mdx.h``).

There is no CPU fallback: if the shared library is missing, or no HIP device is
visible when a compute entry point is called, an exception is raised.
"""

from __future__ import annotations

import ctypes
import os
import pathlib
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_PKG = pathlib.Path(__file__).resolve().parent
LIB_PATH = pathlib.Path(os.environ.get("MDX_LIBRARY", _PKG / "libmdx.so"))

MDX_OK = 0
_EXC = {
    -1: ValueError,            # MDX_ERR_INVALID_VALUE
    -2: NotImplementedError,   # MDX_ERR_UNSUPPORTED
    -3: RuntimeError,          # MDX_ERR_NO_DEVICE
    -4: RuntimeError,          # MDX_ERR_HIP
    -5: RuntimeError,          # MDX_ERR_ROCFFT
    -6: RuntimeError,          # MDX_ERR_RCCL
    -7: MemoryError,           # MDX_ERR_OUT_OF_MEMORY
    -8: RuntimeError,          # MDX_ERR_STATE
    -9: OSError,               # MDX_ERR_IO
}

RDF_ALGO = {"auto": 0, "exact": 1, "filter": 2, "cell": 3}

_vp = c_void_p
_SIGNATURES = {
    # runtime
    "mdx_last_error": (c_char_p, []),
    "mdx_version": (c_int, []),
    "mdx_device_count": (c_int, [POINTER(c_int)]),
    "mdx_device_info": (c_int, [c_int, c_char_p, c_size_t, POINTER(c_int), POINTER(c_size_t), POINTER(c_size_t)]),
    "mdx_malloc": (c_int, [c_int, c_size_t, POINTER(_vp)]),
    "mdx_free": (c_int, [c_int, _vp]),
    "mdx_memcpy_h2d": (c_int, [c_int, _vp, _vp, c_size_t]),
    "mdx_memcpy_d2h": (c_int, [c_int, _vp, _vp, c_size_t]),
    "mdx_memset": (c_int, [c_int, _vp, c_int, c_size_t]),
    "mdx_upload": (c_int, [c_int, _vp, _vp, c_size_t]),
    "mdx_upload_rows": (c_int, [c_int, _vp, _vp, c_size_t, c_size_t, c_size_t]),
    "mdx_trim_cache": (c_int, [c_int, POINTER(c_size_t)]),
    "mdx_cached_bytes": (c_int, [c_int, POINTER(c_size_t)]),
    "mdx_runtime_info": (c_int, [c_char_p, c_size_t]),
    "mdx_device_synchronize": (c_int, [c_int]),
    "mdx_host_register": (c_int, [c_int, _vp, c_size_t]),
    "mdx_host_unregister": (c_int, [c_int, _vp]),
    "mdx_synth_random_walk": (c_int, [c_int, _vp, c_int64, c_int64, POINTER(c_float), c_float, c_uint64, c_int]),
    "mdx_synth_random_walk_f64": (c_int, [c_int, _vp, c_int64, c_int64, POINTER(c_float), c_float, c_uint64]),
    # collectives
    "mdx_comm_unique_id": (c_int, [_vp]),
    "mdx_comm_init_rank": (c_int, [POINTER(_vp), c_int, _vp, c_int, c_int]),
    "mdx_comm_destroy": (c_int, [_vp]),
    "mdx_comm_count": (c_int, [_vp, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "mdx_comm_barrier": (c_int, [_vp]),
    "mdx_comm_allreduce_f64": (c_int, [_vp, _vp, c_int64, c_int]),
    "mdx_comm_allreduce_i64": (c_int, [_vp, _vp, c_int64]),
    # RDF
    "mdx_rdf_create": (c_int, [POINTER(_vp), c_int, c_int, _vp, c_int64, c_int64, c_int]),
    "mdx_rdf_destroy": (c_int, [_vp]),
    "mdx_rdf_reset": (c_int, [_vp]),
    "mdx_rdf_accumulate": (c_int, [_vp, _vp, c_int64, _vp, c_int64, _vp, c_int64]),
    "mdx_rdf_accumulate_device": (c_int, [_vp, _vp, c_int64, _vp, c_int64, _vp, c_int64]),
    "mdx_rdf_set_drop_axis": (c_int, [_vp, c_int]),
    "mdx_rdf_set_grouping": (c_int, [_vp, c_int, c_int64, _vp, _vp]),
    "mdx_rdf_counts": (c_int, [_vp, _vp]),
    "mdx_rdf_synchronize": (c_int, [_vp]),
    "mdx_rdf_allreduce": (c_int, [_vp, _vp]),
    "mdx_rdf_stats": (c_int, [_vp, POINTER(c_int64), POINTER(c_double), POINTER(c_int64), POINTER(c_int64),
                              POINTER(c_int64)]),
    "mdx_rdf_enable_timing": (c_int, [_vp, c_int]),
    "mdx_rdf_debug_counters": (c_int, [_vp, _vp]),
    "mdx_rdf_kernel_clock": (c_int, [_vp, POINTER(c_double)]),
    "mdx_rdf_debug_sorted": (c_int, [_vp, c_int64, c_int64, _vp, _vp]),
    "mdx_radial_histogram": (c_int, [c_int, _vp, c_int64, _vp, c_int64, c_int, _vp, _vp, c_int64, c_int64, _vp]),
    # structure factor
    "mdx_sq_create": (c_int, [POINTER(_vp), c_int, _vp, c_int64, _vp, c_int, _vp, c_int]),
    "mdx_sq_destroy": (c_int, [_vp]),
    "mdx_sq_reset": (c_int, [_vp]),
    "mdx_sq_accumulate": (c_int, [_vp, _vp, c_int64, c_int64]),
    "mdx_sq_accumulate_device": (c_int, [_vp, _vp, c_int64, c_int64]),
    "mdx_sq_result": (c_int, [_vp, _vp]),
    "mdx_sq_allreduce": (c_int, [_vp, _vp]),
    "mdx_sq_stats": (c_int, [_vp, POINTER(c_int64), POINTER(c_double)]),
    "mdx_sq_enable_timing": (c_int, [_vp, c_int]),
    "mdx_fourier_sum": (c_int, [c_int, _vp, c_int64, _vp, c_int64, _vp]),
    # intermediate scattering function
    "mdx_isf_create": (c_int, [POINTER(_vp), c_int, _vp, c_int64, _vp, c_int, _vp, c_int, c_int, c_int]),
    "mdx_isf_destroy": (c_int, [_vp]),
    "mdx_isf_reset": (c_int, [_vp]),
    "mdx_isf_accumulate": (c_int, [_vp, _vp, c_int64, c_int64]),
    "mdx_isf_result": (c_int, [_vp, _vp, _vp]),
    "mdx_isf_stats": (c_int, [_vp, POINTER(c_int64), POINTER(c_double), POINTER(c_int64)]),
    "mdx_isf_enable_timing": (c_int, [_vp, c_int]),
    # time correlation
    "mdx_msd_create": (c_int, [POINTER(_vp), c_int, c_int64, c_int, c_int]),
    "mdx_msd_destroy": (c_int, [_vp]),
    "mdx_msd_reset": (c_int, [_vp]),
    "mdx_msd_n_fft": (c_int, [_vp, POINTER(c_int64)]),
    "mdx_msd_transform": (c_int, [_vp, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "mdx_msd_push": (c_int, [_vp, c_int, _vp, c_int64, c_int64, c_int64, c_int]),
    "mdx_msd_push_device": (c_int, [_vp, c_int, _vp, c_int64, c_int64, c_int64, c_int]),
    "mdx_msd_push_device_f32": (c_int, [_vp, c_int, _vp, c_int64, c_int64, c_int64, c_int]),
    "mdx_msd_result": (c_int, [_vp, _vp, _vp]),
    "mdx_msd_allreduce": (c_int, [_vp, _vp]),
    "mdx_msd_stats": (c_int, [_vp, POINTER(c_int64), POINTER(c_double), POINTER(c_int64)]),
    "mdx_msd_enable_timing": (c_int, [_vp, c_int]),
    "mdx_correlate": (c_int, [c_int, _vp, _vp, c_int64, c_int64, _vp, _vp]),
    # trajectory ingest
    "mdx_traj_open": (c_int, [POINTER(_vp), c_char_p]),
    "mdx_traj_close": (c_int, [_vp]),
    "mdx_traj_info": (c_int, [_vp, POINTER(c_int64), POINTER(c_int64), POINTER(c_int), POINTER(c_int),
                              POINTER(c_int)]),
    "mdx_traj_read_positions": (c_int, [_vp, _vp, c_int64, _vp]),
    "mdx_traj_read_boxes": (c_int, [_vp, _vp, c_int64, _vp]),
    "mdx_traj_read_times": (c_int, [_vp, _vp, c_int64, _vp]),
    "mdx_traj_load_device": (c_int, [_vp, c_int, _vp, c_int64, _vp, c_int64, _vp]),
    "mdx_rdf_accumulate_traj": (c_int, [_vp, _vp, _vp, c_int64, _vp, _vp, c_int64, _vp, c_int64]),
    "mdx_sq_set_grouping": (c_int, [_vp, c_int64, _vp, _vp]),
    "mdx_isf_accumulate_device": (c_int, [_vp, _vp, c_int64, c_int64]),
    "mdx_inner": (c_int, [c_int, _vp, c_int64, _vp, c_int64, _vp]),
    "mdx_trig_rowsums": (c_int, [c_int, _vp, c_int64, c_int64, _vp, _vp]),
    "mdx_trig_rowsums_device": (c_int, [c_int, _vp, c_int64, c_int64, _vp, _vp]),
    "mdx_isf_synchronize": (c_int, [_vp]),
    "mdx_sq_synchronize": (c_int, [_vp]),
    "mdx_isf_set_grouping": (c_int, [_vp, c_int64, _vp, _vp]),
    "mdx_sq_accumulate_traj": (c_int, [_vp, _vp, _vp, c_int64, _vp, c_int64]),
    "mdx_isf_accumulate_traj": (c_int, [_vp, _vp, _vp, c_int64, _vp, c_int64]),
    "mdx_msd_push_traj": (c_int, [_vp, c_int, _vp, _vp, c_int64, _vp, c_int64, c_int, _vp, c_int, _vp]),
    "mdx_msd_push_f32": (c_int, [_vp, c_int, _vp, c_int64, c_int64, c_int, _vp, c_int, _vp]),
    "mdx_msd_system_com_f32": (c_int, [_vp, _vp, c_int64, c_int64, _vp, c_int, _vp, c_int, _vp]),
    "mdx_msd_push_f64": (c_int, [_vp, c_int, _vp, c_int64, c_int64, c_int, _vp, c_int, _vp]),
    "mdx_msd_system_com_f64": (c_int, [_vp, _vp, c_int64, c_int64, _vp, c_int, _vp, c_int, _vp]),
    "mdx_msd_set_grouping": (c_int, [_vp, c_int64, _vp, _vp]),
    "mdx_msd_set_initial_images": (c_int, [_vp, _vp, c_int64]),
    "mdx_msd_result_acf": (c_int, [_vp, _vp]),
    "mdx_msd_system_com_traj": (c_int, [_vp, _vp, _vp, c_int64, _vp, c_int64, _vp, c_int, _vp, c_int, _vp]),
    "mdx_msd_cross": (c_int, [_vp, _vp, c_int64, _vp]),
    "mdx_msd_push_frames_device": (c_int, [_vp, c_int, _vp, c_int, c_int64, c_int64, _vp, c_int64, c_int, _vp,
                                           c_int, _vp]),
    "mdx_msd_system_com_device": (c_int, [_vp, _vp, c_int, c_int64, c_int64, _vp, c_int64, _vp, c_int, _vp,
                                          c_int, _vp]),
}

EXPORTS = tuple(_SIGNATURES)

_lib = None


def lib():
    """The loaded ``libmdx.so``; raises ImportError when it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP library first "
                "(`make -C mdhelper_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "mdhelper_amd has no CPU fallback.")
        # the HIP runtime reads its flags at its first call: the library's own load-time initialiser sets this one as
        # well (csrc/mdx_runtime.hip, mdx_process_init — why: the runtime must never page-lock caller memory on its
        # own); here for a runtime that something else in the process initialises between now and then
        os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")
        handle = ctypes.CDLL(str(LIB_PATH))
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        problems = runtime(handle)["problems"]
        if problems and os.environ.get("MDX_ALLOW_FOREIGN_RUNTIME") != "1":
            raise ImportError(
                "libmdx.so is not running on the ROCm runtime it was built for: " + "; ".join(problems) +
                ".  This happens when another copy of the HIP / rocFFT / RCCL libraries is loaded into the process "
                "(e.g. `import torch`: the wheel bundles its own ROCm).  mdhelper_amd needs no torch; import it in "
                "a process of its own, or set MDX_ALLOW_FOREIGN_RUNTIME=1 to run on the foreign runtime anyway.")
        _lib = handle
    return _lib


_ROCM_LIBS = ("libamdhip64", "libhsa-runtime64", "librocfft", "librccl")


def mapped_rocm_libraries() -> dict:
    """``{stem: [files]}`` of the HIP / HSA / rocFFT / RCCL shared objects mapped into this process
    (``/proc/self/maps``; real paths, each file once)."""
    found = {stem: [] for stem in _ROCM_LIBS}
    try:
        with open("/proc/self/maps") as f:
            lines = f.readlines()
    except OSError:
        return found
    for line in lines:
        parts = line.split(None, 5)
        if len(parts) < 6:
            continue
        path = parts[5].strip()
        base = os.path.basename(path)
        for stem in _ROCM_LIBS:
            if base.startswith(stem + ".so"):
                real = os.path.realpath(path)
                if real not in found[stem]:
                    found[stem].append(real)
    return found


def runtime(handle=None) -> dict:
    """
    The user-mode ROCm stack ``libmdx.so`` runs on in this process: the files its HIP / rocFFT / RCCL calls are
    bound to (``mdx_runtime_info``: ``dladdr`` of the resolved symbols), the versions they report, every copy of
    these libraries mapped into the process, and ``problems``: a list that is empty exactly when each library is
    mapped once and from the ROCm installation the library was built for.  One process, one runtime — what the
    reference's single-process drivers have by construction (reference src/mdhelper/analysis/base.py:137-172).
    """
    h = handle if handle is not None else lib()
    buf = ctypes.create_string_buffer(4096)
    h.mdx_runtime_info.restype = c_int
    h.mdx_runtime_info.argtypes = [c_char_p, c_size_t]
    if h.mdx_runtime_info(buf, 4096) != MDX_OK:
        raise RuntimeError("mdx_runtime_info failed")
    info = dict(line.split("=", 1) for line in buf.value.decode().splitlines() if "=" in line)
    root = os.path.realpath(info.get("rocm_root", "/opt/rocm"))
    mapped = mapped_rocm_libraries()
    problems = []
    for key in ("libamdhip64", "librocfft", "librccl"):
        bound = os.path.realpath(info.get(key, "?"))
        info[key] = bound
        if not bound.startswith(root + os.sep):
            problems.append(f"{key} is bound to {bound}, not to {root}")
    for stem, files in mapped.items():
        if len(files) > 1:
            problems.append(f"{stem} is mapped {len(files)} times ({', '.join(files)})")
        for f in files:
            if not f.startswith(root + os.sep) and not any(f in p for p in problems):
                problems.append(f"{stem} is mapped from {f}, not from {root}")
    for key in ("hip_runtime_version", "hip_driver_version", "rccl_version"):
        if key in info:
            info[key] = int(info[key])
    info["rocm_root"] = root
    info["mapped"] = mapped
    info["problems"] = problems
    return info


def runtime_summary() -> dict:
    """``runtime()`` without the per-library lists: what bench.py, smoke() and the launcher print per rank."""
    r = runtime()
    out = {k: r[k] for k in ("libamdhip64", "librocfft", "librccl", "hip_runtime_version", "rccl_version",
                             "rocfft_version") if k in r}
    hsa = r["mapped"].get("libhsa-runtime64") or []
    out["libhsa-runtime64"] = hsa[0] if len(hsa) == 1 else hsa
    out["one_runtime"] = not r["problems"]
    return out


def check(rc: int) -> None:
    if rc != MDX_OK:
        msg = lib().mdx_last_error()
        msg = msg.decode("utf-8", "replace") if msg else f"libmdx error {rc}"
        raise _EXC.get(rc, RuntimeError)(msg)


def device_count() -> int:
    n = c_int(0)
    check(lib().mdx_device_count(ctypes.byref(n)))
    return n.value


def require_device(dev: int = 0) -> None:
    if device_count() <= dev:
        raise RuntimeError(
            f"HIP device {dev} is not available; mdhelper_amd runs its analysis kernels on an "
            "AMD GPU only (no CPU fallback).")
