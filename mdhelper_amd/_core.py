"""
Thin object wrappers over the C-ABI handles of ``libmdx.so``: device arrays,
the RCCL communicator and the three analysis engines.  Host-side glue only —
all arithmetic happens in the HIP library.
"""

from __future__ import annotations

import ctypes
from ctypes import byref, c_double, c_int, c_int64, c_size_t, c_void_p

import numpy as np

from . import _lib
from ._lib import check, lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


class DeviceArray:
    """A typed HBM allocation (``mdx_malloc``); inputs of the ``*_device`` entry points."""

    def __init__(self, shape, dtype, dev: int = 0):
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        self.dev = dev
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = c_void_p()
        check(lib().mdx_malloc(dev, self.nbytes, byref(p)))
        self.ptr = p

    @classmethod
    def from_host(cls, arr, dev: int = 0):
        arr = np.ascontiguousarray(arr)
        out = cls(arr.shape, arr.dtype, dev)
        check(lib().mdx_memcpy_h2d(dev, out.ptr, _ptr(arr), arr.nbytes))
        return out

    @classmethod
    def upload(cls, arr, dev: int = 0):
        """Large host arrays: through the library's pinned ring and its copy threads, or by one DMA
        when the array is page-locked (``mdx_upload``)."""
        arr = np.ascontiguousarray(arr)
        out = cls(arr.shape, arr.dtype, dev)
        try:
            check(lib().mdx_upload(dev, out.ptr, _ptr(arr), arr.nbytes))
        except Exception:
            out.free()
            raise
        return out

    def upload_columns(self, host, first: int, count: int):
        """Fill this array ``[T, count, ...]`` with columns ``[first, first + count)`` of the host array
        ``[T, N, ...]`` (``mdx_upload_rows``: rows of >= 4 KB in pageable anonymous memory are page-locked, read by 2-D DMA
        and unlocked slice by slice; page-locked memory: one 2-D DMA; short rows, file mappings: copy threads + pinned ring)."""
        host = np.asarray(host)
        if host.dtype != self.dtype or not host.flags.c_contiguous or host.shape[0] != self.shape[0] \
                or host.shape[2:] != self.shape[2:] or self.shape[1] != count \
                or not 0 <= first <= first + count <= host.shape[1]:
            raise ValueError("upload_columns: shapes / dtype do not match")
        item = int(np.prod(host.shape[2:], dtype=np.int64)) * host.dtype.itemsize
        src = c_void_p(host.ctypes.data + first * item)
        check(lib().mdx_upload_rows(self.dev, self.ptr, src, count * item, host.shape[1] * item, host.shape[0]))

    def to_host(self, first: int = 0, count: int | None = None):
        """Copy rows [first, first+count) of the leading axis back to the host."""
        n0 = self.shape[0]
        count = n0 - first if count is None else count
        row = self.nbytes // max(n0, 1)
        out = np.empty((count,) + self.shape[1:], dtype=self.dtype)
        src = c_void_p(self.ptr.value + first * row)
        check(lib().mdx_memcpy_d2h(self.dev, _ptr(out), src, count * row))
        return out

    def offset(self, first: int):
        """Raw pointer to row ``first`` of the leading axis."""
        row = self.nbytes // max(self.shape[0], 1)
        return c_void_p(self.ptr.value + first * row)

    @staticmethod
    def view(base, shape):
        """The leading bytes of ``base`` as an array of another shape (same dtype) that does not own its memory."""
        v = object.__new__(_DeviceView)
        v.shape = tuple(int(x) for x in shape)
        v.dtype, v.dev = base.dtype, base.dev
        v.nbytes = int(np.prod(v.shape)) * base.dtype.itemsize
        if v.nbytes > base.nbytes:
            raise ValueError("view larger than its base")
        v.ptr = c_void_p(base.ptr.value)
        v.base = base
        return v

    def rows(self, first: int, count: int):
        """Rows [first, first+count) of the leading axis as a DeviceArray that does not own its memory
        (valid as long as this one is)."""
        if not 0 <= first <= first + count <= self.shape[0]:
            raise IndexError("rows out of range")
        view = object.__new__(_DeviceView)
        view.shape = (int(count),) + self.shape[1:]
        view.dtype, view.dev = self.dtype, self.dev
        view.nbytes = self.nbytes // max(self.shape[0], 1) * int(count)
        view.ptr = self.offset(first)
        view.base = self
        return view

    def free(self):
        if getattr(self, "ptr", None) is not None and self.ptr.value:
            lib().mdx_free(self.dev, self.ptr)
            self.ptr = c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _DeviceView(DeviceArray):
    """Non-owning window into a DeviceArray (``DeviceArray.rows``)."""

    def free(self):
        self.ptr = c_void_p()


def device_info(dev: int = 0):
    name = ctypes.create_string_buffer(256)
    cus = c_int()
    total = c_size_t()
    free = c_size_t()
    check(lib().mdx_device_info(dev, name, 256, byref(cus), byref(total), byref(free)))
    cached = c_size_t()
    check(lib().mdx_cached_bytes(dev, byref(cached)))
    # hbm_free_bytes: what the driver reports; hbm_cached_bytes: blocks of destroyed handles this process keeps
    # (mdx_trim_cache) — the library's own allocations can take them, nobody else can
    return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": total.value,
            "hbm_free_bytes": free.value, "hbm_cached_bytes": cached.value,
            "hbm_available_bytes": free.value + cached.value}


def synchronize(dev: int = 0):
    check(lib().mdx_device_synchronize(dev))


def synth_random_walk(n_frames, n_atoms, box_lengths, sigma, seed, *, wrap=True, dev=0,
                      dtype=np.float32):
    """Synthetic trajectory generated in HBM (``mdx_synth_random_walk``)."""
    out = DeviceArray((n_frames, n_atoms, 3), dtype, dev)
    L = (ctypes.c_float * 3)(*[float(x) for x in box_lengths])
    if np.dtype(dtype) == np.float32:
        check(lib().mdx_synth_random_walk(dev, out.ptr, n_frames, n_atoms, L, float(sigma),
                                          int(seed), int(bool(wrap))))
    else:
        check(lib().mdx_synth_random_walk_f64(dev, out.ptr, n_frames, n_atoms, L, float(sigma),
                                              int(seed)))
    return out


class RcclComm:
    """One rank of an RCCL communicator (``mdx_comm_*``)."""

    device_collectives = True

    def __init__(self, rank: int, world_size: int, unique_id: bytes, dev: int = 0):
        assert len(unique_id) == 128
        self.rank, self.world_size, self.dev = rank, world_size, dev
        buf = ctypes.create_string_buffer(unique_id, 128)
        h = c_void_p()
        check(lib().mdx_comm_init_rank(byref(h), dev, buf, rank, world_size))
        self.handle = h

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        check(lib().mdx_comm_unique_id(buf))
        return buf.raw

    def barrier(self):
        check(lib().mdx_comm_barrier(self.handle))

    def rccl_info(self):
        """(ranks, this rank, device) as RCCL reports them for the communicator."""
        n, r, d = c_int(), c_int(), c_int()
        check(lib().mdx_comm_count(self.handle, byref(n), byref(r), byref(d)))
        return n.value, r.value, d.value

    def allreduce(self, arr, op="sum"):
        arr = np.ascontiguousarray(arr)
        if arr.dtype == np.int64 and op == "sum":
            check(lib().mdx_comm_allreduce_i64(self.handle, _ptr(arr), arr.size))
        else:
            arr = arr.astype(np.float64)
            check(lib().mdx_comm_allreduce_f64(self.handle, _ptr(arr), arr.size, int(op == "max")))
        return arr

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            lib().mdx_comm_destroy(self.handle)
            self.handle = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Engine:
    _destroy = None

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            getattr(lib(), self._destroy)(self.handle)
            self.handle = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RdfEngine(_Engine):
    """``mdx_rdf_*``: pair-distance histogram accumulated over batches of frames."""

    _destroy = "mdx_rdf_destroy"

    def __init__(self, edges, exclusion=None, *, algo="auto", dev=0, timing=False):
        self.edges = np.ascontiguousarray(edges, dtype=np.float64)
        self.n_bins = self.edges.size - 1
        e1, e2 = (0, 0) if exclusion is None else (int(exclusion[0]), int(exclusion[1]))
        h = c_void_p()
        check(lib().mdx_rdf_create(byref(h), dev, self.n_bins, _ptr(self.edges), e1, e2,
                                   _lib.RDF_ALGO[algo]))
        self.handle = h
        self.dev = dev
        if timing:
            check(lib().mdx_rdf_enable_timing(h, 1))

    def accumulate(self, pos1, pos2=None, boxes=None):
        """pos: float32[F, N, 3] (host); boxes float32[F, 6] or None."""
        p1 = np.ascontiguousarray(pos1, dtype=np.float32)
        if p1.ndim == 2:
            p1 = p1[None]
        F, n1 = p1.shape[0], p1.shape[1]
        if pos2 is None or pos2 is pos1:
            p2, n2 = None, n1
        else:
            p2 = np.ascontiguousarray(pos2, dtype=np.float32)
            if p2.ndim == 2:
                p2 = p2[None]
            n2 = p2.shape[1]
            if p2.shape[0] != F:
                raise ValueError("pos1 and pos2 must hold the same number of frames.")
        b = None
        if boxes is not None:
            b = np.ascontiguousarray(np.broadcast_to(np.asarray(boxes, dtype=np.float32), (F, 6)))
        check(lib().mdx_rdf_accumulate(self.handle, _ptr(p1), n1, _ptr(p2), n2, _ptr(b), F))

    def accumulate_device(self, d_pos1, n1, d_pos2, n2, d_boxes, n_frames):
        check(lib().mdx_rdf_accumulate_device(self.handle, d_pos1, n1, d_pos2, n2, d_boxes, n_frames))

    def set_drop_axis(self, axis):
        """2-D mode: coordinate ``axis`` (0, 1, 2; ``None`` = off) is zeroed on the device and the
        cell length along it set to the largest one (``mdx_rdf_set_drop_axis``)."""
        check(lib().mdx_rdf_set_drop_axis(self.handle, -1 if axis is None else int(axis)))

    def set_grouping(self, which, offsets, masses):
        """Rows of set ``which`` (1 or 2) become particles of molecules ``[offsets[g], offsets[g+1])``
        whose centres of mass are binned; ``offsets=None`` removes the grouping."""
        if offsets is None:
            check(lib().mdx_rdf_set_grouping(self.handle, which, 0, None, None))
            return
        o = np.ascontiguousarray(offsets, dtype=np.int64)
        m = np.ascontiguousarray(masses, dtype=np.float64)
        if len(m) != o[-1]:
            raise ValueError("masses must hold one entry per particle of the grouping.")
        check(lib().mdx_rdf_set_grouping(self.handle, which, len(o) - 1, _ptr(o), _ptr(m)))

    def accumulate_traj(self, traj_file, frames, boxes, index1=None, index2=None, same=True):
        """Frames of a native trajectory file (``io.TrajectoryFile``); ``index``: particle
        selections (None = all particles), ``same``: one group against itself."""
        f = np.ascontiguousarray(frames, dtype=np.int64)
        b = None if boxes is None else np.ascontiguousarray(boxes, dtype=np.float32)
        i1 = None if index1 is None else np.ascontiguousarray(index1, dtype=np.int32)
        i2 = None if (same or index2 is None) else np.ascontiguousarray(index2, dtype=np.int32)
        n1 = 0 if i1 is None else len(i1)
        # index2 NULL with n2 == 0 means "the same group twice"; n2 < 0 can never mean that
        n2 = 0 if same else (traj_file.n_atoms if i2 is None else len(i2))
        check(lib().mdx_rdf_accumulate_traj(self.handle, traj_file.handle, _ptr(f), len(f), _ptr(b),
                                            _ptr(i1), n1, _ptr(i2), n2))

    def counts(self):
        out = np.zeros(self.n_bins, dtype=np.int64)
        check(lib().mdx_rdf_counts(self.handle, _ptr(out)))
        return out

    def synchronize(self):
        check(lib().mdx_rdf_synchronize(self.handle))

    def reset(self):
        check(lib().mdx_rdf_reset(self.handle))

    def allreduce(self, comm: RcclComm):
        check(lib().mdx_rdf_allreduce(self.handle, comm.handle))

    def stats(self):
        n, ms, pe, px, pc = c_int64(), c_double(), c_int64(), c_int64(), c_int64()
        check(lib().mdx_rdf_stats(self.handle, byref(n), byref(ms), byref(pe), byref(px), byref(pc)))
        raw = np.zeros(4, dtype=np.int64)
        check(lib().mdx_rdf_debug_counters(self.handle, _ptr(raw)))
        hz = c_double()
        check(lib().mdx_rdf_kernel_clock(self.handle, byref(hz)))
        return {"launches": n.value, "kernel_ms": ms.value, "pairs_evaluated": pe.value,
                "pairs_exact": px.value, "pairs_computed": pc.value,
                "cell_units": int(raw[1]), "cell_units_general": int(raw[2]),
                "clock_hz": hz.value}


def radial_histogram_device(pos1, pos2, n_bins, edges, dims, exclusion=None, dev=0):
    """``mdx_radial_histogram``: one frame, host buffers."""
    p1 = np.ascontiguousarray(pos1, dtype=np.float32).reshape(-1, 3)
    p2 = np.ascontiguousarray(pos2, dtype=np.float32).reshape(-1, 3)
    box = None if dims is None else np.ascontiguousarray(dims, dtype=np.float32).reshape(6)
    edges = np.ascontiguousarray(edges, dtype=np.float64)
    e1, e2 = (0, 0) if exclusion is None else (int(exclusion[0]), int(exclusion[1]))
    counts = np.zeros(n_bins, dtype=np.int64)
    same = pos2 is pos1
    check(lib().mdx_radial_histogram(dev, _ptr(p1), p1.shape[0], _ptr(p1 if same else p2),
                                     p2.shape[0], n_bins, _ptr(edges), _ptr(box), e1, e2,
                                     _ptr(counts)))
    return counts


class SqEngine(_Engine):
    """``mdx_sq_*``: fused exp(i q.r) accumulation and pair products."""

    _destroy = "mdx_sq_destroy"

    def __init__(self, wavevectors, group_sizes, pairs, *, dev=0, timing=False):
        self.q = np.ascontiguousarray(wavevectors, dtype=np.float64).reshape(-1, 3)
        self.offsets = np.concatenate(([0], np.cumsum(group_sizes))).astype(np.int64)
        self.pairs = np.ascontiguousarray(
            [(-1, -1) if p[0] is None else (int(p[0]), int(p[1])) for p in pairs], dtype=np.int32)
        h = c_void_p()
        check(lib().mdx_sq_create(byref(h), dev, _ptr(self.q), self.q.shape[0], _ptr(self.offsets),
                                  len(group_sizes), _ptr(self.pairs), self.pairs.shape[0]))
        self.handle = h
        self.dev = dev
        if timing:
            check(lib().mdx_sq_enable_timing(h, 1))

    _set_grouping = "mdx_sq_set_grouping"

    def set_grouping(self, offsets, masses):
        """Incoming rows become particles of molecules ``[offsets[g], offsets[g+1])`` whose float32
        centres of mass enter the Fourier sums; ``offsets=None`` removes the grouping."""
        fn = getattr(lib(), self._set_grouping)
        if offsets is None:
            check(fn(self.handle, 0, None, None))
            return
        o = np.ascontiguousarray(offsets, dtype=np.int64)
        m = np.ascontiguousarray(masses, dtype=np.float64)
        if len(m) != o[-1]:
            raise ValueError("masses must hold one entry per particle of the grouping.")
        check(fn(self.handle, len(o) - 1, _ptr(o), _ptr(m)))

    def accumulate(self, pos):
        p = np.ascontiguousarray(pos, dtype=np.float32)
        if p.ndim == 2:
            p = p[None]
        check(lib().mdx_sq_accumulate(self.handle, _ptr(p), p.shape[1], p.shape[0]))

    def accumulate_device(self, d_pos, n, n_frames):
        """Asynchronous on the engine's stream: ``synchronize()`` before the frames are overwritten."""
        check(lib().mdx_sq_accumulate_device(self.handle, d_pos, n, n_frames))

    def synchronize(self):
        check(lib().mdx_sq_synchronize(self.handle))

    def accumulate_traj(self, traj_file, frames, index=None):
        """Frames of a native trajectory file; ``index``: particles in concatenated-group order."""
        f = np.ascontiguousarray(frames, dtype=np.int64)
        i = None if index is None else np.ascontiguousarray(index, dtype=np.int32)
        check(lib().mdx_sq_accumulate_traj(self.handle, traj_file.handle, _ptr(f), len(f), _ptr(i),
                                           0 if i is None else len(i)))

    def result(self):
        out = np.zeros((self.pairs.shape[0], self.q.shape[0]), dtype=np.float64)
        check(lib().mdx_sq_result(self.handle, _ptr(out)))
        return out

    def reset(self):
        check(lib().mdx_sq_reset(self.handle))

    def allreduce(self, comm: RcclComm):
        check(lib().mdx_sq_allreduce(self.handle, comm.handle))

    def stats(self):
        n, ms = c_int64(), c_double()
        check(lib().mdx_sq_stats(self.handle, byref(n), byref(ms)))
        return {"launches": n.value, "kernel_ms": ms.value}


class IsfEngine(_Engine):
    """``mdx_isf_*``: coherent / incoherent intermediate scattering functions."""

    _destroy = "mdx_isf_destroy"

    def __init__(self, wavevectors, group_sizes, pairs, n_lags, incoherent=False, *, dev=0,
                 timing=False):
        self.q = np.ascontiguousarray(wavevectors, dtype=np.float64).reshape(-1, 3)
        self.offsets = np.concatenate(([0], np.cumsum(group_sizes))).astype(np.int64)
        self.pairs = np.ascontiguousarray(
            [(-1, -1) if p[0] is None else (int(p[0]), int(p[1])) for p in pairs], dtype=np.int32)
        self.n_lags = int(n_lags)
        self.incoherent = bool(incoherent)
        self.n_slots = 1 if self.pairs[0, 0] < 0 else len(group_sizes)
        h = c_void_p()
        check(lib().mdx_isf_create(byref(h), dev, _ptr(self.q), self.q.shape[0], _ptr(self.offsets),
                                   len(group_sizes), _ptr(self.pairs), self.pairs.shape[0],
                                   self.n_lags, int(self.incoherent)))
        self.handle = h
        self.dev = dev
        if timing:
            check(lib().mdx_isf_enable_timing(h, 1))

    _set_grouping = "mdx_isf_set_grouping"

    def set_grouping(self, offsets, masses):
        """Incoming rows become particles of molecules ``[offsets[g], offsets[g+1])`` whose float32
        centres of mass enter the Fourier sums; ``offsets=None`` removes the grouping."""
        fn = getattr(lib(), self._set_grouping)
        if offsets is None:
            check(fn(self.handle, 0, None, None))
            return
        o = np.ascontiguousarray(offsets, dtype=np.int64)
        m = np.ascontiguousarray(masses, dtype=np.float64)
        if len(m) != o[-1]:
            raise ValueError("masses must hold one entry per particle of the grouping.")
        check(fn(self.handle, len(o) - 1, _ptr(o), _ptr(m)))

    def accumulate(self, pos):
        """pos: float32[F, N, 3], frames in analysis order."""
        p = np.ascontiguousarray(pos, dtype=np.float32)
        if p.ndim == 2:
            p = p[None]
        check(lib().mdx_isf_accumulate(self.handle, _ptr(p), p.shape[1], p.shape[0]))

    def accumulate_device(self, d_pos, n, n_frames):
        """float32[n_frames][n][3] already in HBM (raw device pointer).  Asynchronous on the engine's
        stream: ``synchronize()`` before the frames are overwritten."""
        check(lib().mdx_isf_accumulate_device(self.handle, d_pos, n, n_frames))

    def synchronize(self):
        check(lib().mdx_isf_synchronize(self.handle))

    def accumulate_traj(self, traj_file, frames, index=None):
        """Frames (in analysis order) of a native trajectory file; ``index`` as for SqEngine."""
        f = np.ascontiguousarray(frames, dtype=np.int64)
        i = None if index is None else np.ascontiguousarray(index, dtype=np.int32)
        check(lib().mdx_isf_accumulate_traj(self.handle, traj_file.handle, _ptr(f), len(f), _ptr(i),
                                            0 if i is None else len(i)))

    def result(self):
        cisf = np.zeros((self.n_lags, self.pairs.shape[0], self.q.shape[0]))
        iisf = np.zeros((self.n_lags, self.n_slots, self.q.shape[0])) if self.incoherent else None
        check(lib().mdx_isf_result(self.handle, _ptr(cisf), _ptr(iisf)))
        return cisf, iisf

    def reset(self):
        check(lib().mdx_isf_reset(self.handle))

    def stats(self):
        n, ms, fr = c_int64(), c_double(), c_int64()
        check(lib().mdx_isf_stats(self.handle, byref(n), byref(ms), byref(fr)))
        return {"launches": n.value, "kernel_ms": ms.value, "frames": fr.value}


def fourier_sum_device(wavevectors, positions, dev=0):
    """``mdx_fourier_sum``: complex128[N_q] = sum_j exp(i q.r_j), float64 positions."""
    q = np.ascontiguousarray(wavevectors, dtype=np.float64).reshape(-1, 3)
    r = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
    out = np.zeros(q.shape[0], dtype=np.complex128)
    check(lib().mdx_fourier_sum(dev, _ptr(q), q.shape[0], _ptr(r), r.shape[0], _ptr(out)))
    return out


def inner_device(wavevectors, positions, dev=0):
    """``mdx_inner``: float64[N_q, N] of q_i . r_j."""
    q = np.ascontiguousarray(wavevectors, dtype=np.float64).reshape(-1, 3)
    r = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
    out = np.zeros((q.shape[0], r.shape[0]), dtype=np.float64)
    if out.size:
        for lo in range(0, q.shape[0], 65535):           # (one launch covers 65 535 wavevectors)
            hi = min(q.shape[0], lo + 65535)
            check(lib().mdx_inner(dev, _ptr(q[lo:hi]), hi - lo, _ptr(r), r.shape[0], _ptr(out[lo:hi])))
    return out


def trig_rowsums_device(xs, *, cos=True, sin=True, dev=0):
    """``mdx_trig_rowsums``: ``(sum_j cos(xs[i, j]), sum_j sin(xs[i, j]))`` of a float64 matrix (``None`` for a
    sum that was not asked for)."""
    x = np.ascontiguousarray(xs, dtype=np.float64)
    if x.ndim != 2:
        raise ValueError("xs must be a two-dimensional array.")
    c = np.zeros(x.shape[0]) if cos else None
    s = np.zeros(x.shape[0]) if sin else None
    if x.shape[0]:
        check(lib().mdx_trig_rowsums(dev, _ptr(x), x.shape[0], x.shape[1], _ptr(c), _ptr(s)))
    return c, s


class MsdEngine(_Engine):
    """``mdx_msd_*``: per-group sums of per-particle MSDs through rocFFT."""

    _destroy = "mdx_msd_destroy"

    def __init__(self, n_frames_block, n_blocks, n_groups, *, dev=0, timing=False):
        h = c_void_p()
        check(lib().mdx_msd_create(byref(h), dev, int(n_frames_block), int(n_blocks), int(n_groups)))
        self.handle = h
        self.dev = dev
        self.t_block, self.n_blocks, self.n_groups = int(n_frames_block), int(n_blocks), int(n_groups)
        self.has_grouping = False
        if timing:
            check(lib().mdx_msd_enable_timing(h, 1))

    @property
    def n_fft(self):
        n = c_int64()
        check(lib().mdx_msd_n_fft(self.handle, byref(n)))
        return n.value

    @property
    def transform(self):
        """``(own, r1, r2)``: the engine's own two-pass transform ``n_fft = r1 * r2`` (``own`` True), or the rocFFT
        pipeline (``(False, 0, 0)``)."""
        own, r1, r2 = c_int(), c_int(), c_int()
        check(lib().mdx_msd_transform(self.handle, byref(own), byref(r1), byref(r2)))
        return bool(own.value), r1.value, r2.value

    def push(self, group, positions, first, count, zero_dims=0):
        """positions: float64[T, N_total, 3] on the host."""
        p = np.ascontiguousarray(positions, dtype=np.float64)
        if p.shape[0] < self.t_block * self.n_blocks:
            raise ValueError("positions hold fewer frames than n_blocks * n_frames_block.")
        check(lib().mdx_msd_push(self.handle, group, _ptr(p), p.shape[1], first, count, zero_dims))

    def push_device(self, group, d_pos, n_total, first, count, zero_dims=0):
        check(lib().mdx_msd_push_device(self.handle, group, d_pos, n_total, first, count, zero_dims))

    @property
    def reads_f32(self):
        """Whether the first pass of this engine reads float32 positions where they lie (``push_device_f32``): the
        transforms with a 400- or 64-point first factor (the single-pass kernel, the 400 x R2 family, 2^13 .. 2^16:
        every block length up to 204 800 frames)."""
        own, r1, _r2 = self.transform
        return bool(own) and r1 in (400, 64)

    def push_device_f32(self, group, d_pos, n_total, first, count, zero_dims=0):
        """``mdx_msd_push_device_f32``: a plain particle range of float32 frames resident in HBM, widened by pass A as
        it stages them (``NotImplementedError`` from engines whose transforms do not: see :attr:`reads_f32`)."""
        check(lib().mdx_msd_push_device_f32(self.handle, group, d_pos, n_total, first, count, zero_dims))

    def set_grouping(self, offsets, masses):
        """Rows of the following ``push_traj`` calls are particles of molecules
        ``[offsets[g], offsets[g+1])``; the engine receives their float64 centres of mass.
        ``offsets=None`` removes the grouping (``mdx_msd_set_grouping``)."""
        self.has_grouping = offsets is not None
        if offsets is None:
            check(lib().mdx_msd_set_grouping(self.handle, 0, None, None))
            return
        o = np.ascontiguousarray(offsets, dtype=np.int64)
        m = np.ascontiguousarray(masses, dtype=np.float64)
        if len(m) != o[-1]:
            raise ValueError("masses must hold one entry per particle of the grouping.")
        check(lib().mdx_msd_set_grouping(self.handle, len(o) - 1, _ptr(o), _ptr(m)))

    def set_initial_images(self, images):
        """int[n_sel, 3] image flags the rows of the following unwrapped pushes / system-COM calls
        start from (molecules made whole in the first analysed frame); ``None`` clears."""
        if images is None:
            check(lib().mdx_msd_set_initial_images(self.handle, None, 0))
            return
        im = np.ascontiguousarray(images, dtype=np.int32).reshape(-1, 3)
        check(lib().mdx_msd_set_initial_images(self.handle, _ptr(im), im.shape[0]))

    def push_f32(self, group, positions, *, unwrap_dims=None, zero_dims=0, shift=None):
        """A group's float32 (or float64) positions ``[T, n_sel, 3]`` from host memory through the
        device-side frame preparation (unwrap, molecule centres, shift): ``mdx_msd_push_f32`` /
        ``mdx_msd_push_f64``."""
        f64 = np.asarray(positions).dtype == np.float64
        p = np.ascontiguousarray(positions, dtype=np.float64 if f64 else np.float32)
        d = None if unwrap_dims is None else np.ascontiguousarray(unwrap_dims, dtype=np.float64)[:3]
        sh = None if shift is None else np.ascontiguousarray(shift, dtype=np.float64)
        fn = lib().mdx_msd_push_f64 if f64 else lib().mdx_msd_push_f32
        check(fn(self.handle, group, _ptr(p), p.shape[0], p.shape[1], 0 if d is None else 1, _ptr(d),
                 zero_dims, _ptr(sh)))

    def system_com_f32(self, positions, masses, *, unwrap_dims=None, wrap_dims=None):
        """float64[T, 3] system centre of mass per frame of float32 / float64 positions
        ``[T, n_sel, 3]``; with a grouping set: of the molecules' (wrapped) centres."""
        f64 = np.asarray(positions).dtype == np.float64
        p = np.ascontiguousarray(positions, dtype=np.float64 if f64 else np.float32)
        m = np.ascontiguousarray(masses, dtype=np.float64)
        dims = unwrap_dims if unwrap_dims is not None else wrap_dims
        d = None if dims is None else np.ascontiguousarray(dims, dtype=np.float64)[:3]
        out = np.empty((p.shape[0], 3), dtype=np.float64)
        fn = lib().mdx_msd_system_com_f64 if f64 else lib().mdx_msd_system_com_f32
        check(fn(self.handle, _ptr(p), p.shape[0], p.shape[1], _ptr(m), 0 if unwrap_dims is None else 1,
                 _ptr(d), 0 if wrap_dims is None else 1, _ptr(out)))
        return out

    def push_frames_device(self, group, d_pos, n_total, index=None, *, unwrap_dims=None, zero_dims=0,
                           shift=None):
        """A group's positions out of frames resident in HBM (``DeviceArray`` float32 / float64
        ``[T, n_total, 3]``): rows ``index`` (None: all) gathered by the frame-preparation kernels
        (``mdx_msd_push_frames_device``)."""
        i = None if index is None else np.ascontiguousarray(index, dtype=np.int32)
        d = None if unwrap_dims is None else np.ascontiguousarray(unwrap_dims, dtype=np.float64)[:3]
        sh = None if shift is None else np.ascontiguousarray(shift, dtype=np.float64)
        check(lib().mdx_msd_push_frames_device(
            self.handle, group, d_pos.ptr, d_pos.dtype.itemsize, d_pos.shape[0], int(n_total), _ptr(i),
            0 if i is None else len(i), 0 if d is None else 1, _ptr(d), zero_dims, _ptr(sh)))

    def system_com_device(self, d_pos, n_total, index, masses, *, unwrap_dims=None, wrap_dims=None):
        """float64[T, 3] system centre of mass per frame of HBM-resident frames."""
        i = None if index is None else np.ascontiguousarray(index, dtype=np.int32)
        m = np.ascontiguousarray(masses, dtype=np.float64)
        dims = unwrap_dims if unwrap_dims is not None else wrap_dims
        d = None if dims is None else np.ascontiguousarray(dims, dtype=np.float64)[:3]
        out = np.empty((d_pos.shape[0], 3), dtype=np.float64)
        check(lib().mdx_msd_system_com_device(
            self.handle, d_pos.ptr, d_pos.dtype.itemsize, d_pos.shape[0], int(n_total), _ptr(i), len(m),
            _ptr(m), 0 if unwrap_dims is None else 1, _ptr(d), 0 if wrap_dims is None else 1, _ptr(out)))
        return out

    def system_com_traj(self, traj_file, frames, index, masses, *, unwrap_dims=None, wrap_dims=None):
        """float64[len(frames), 3] system centre of mass per frame (``Onsager(center=True)``)."""
        f = np.ascontiguousarray(frames, dtype=np.int64)
        i = None if index is None else np.ascontiguousarray(index, dtype=np.int32)
        m = np.ascontiguousarray(masses, dtype=np.float64)
        dims = unwrap_dims if unwrap_dims is not None else wrap_dims
        d = None if dims is None else np.ascontiguousarray(dims, dtype=np.float64)[:3]
        out = np.empty((len(f), 3), dtype=np.float64)
        check(lib().mdx_msd_system_com_traj(self.handle, traj_file.handle, _ptr(f), len(f), _ptr(i),
                                            len(m), _ptr(m), 0 if unwrap_dims is None else 1, _ptr(d),
                                            0 if wrap_dims is None else 1, _ptr(out)))
        return out

    def push_traj(self, group, traj_file, frames, index=None, *, unwrap_dims=None, zero_dims=0,
                  shift=None):
        """A group's positions straight from a native trajectory file; ``unwrap_dims``: box
        lengths for the device-side global unwrap (None: positions are used as stored)."""
        f = np.ascontiguousarray(frames, dtype=np.int64)
        i = None if index is None else np.ascontiguousarray(index, dtype=np.int32)
        d = None if unwrap_dims is None else np.ascontiguousarray(unwrap_dims, dtype=np.float64)[:3]
        sh = None if shift is None else np.ascontiguousarray(shift, dtype=np.float64)
        check(lib().mdx_msd_push_traj(self.handle, group, traj_file.handle, _ptr(f), len(f), _ptr(i),
                                      0 if i is None else len(i), 0 if d is None else 1, _ptr(d),
                                      zero_dims, _ptr(sh)))

    def result(self, want_msd=True):
        msd = np.zeros((self.n_groups, self.n_blocks, self.t_block)) if want_msd else None
        traj = np.zeros((self.n_groups, self.n_blocks, self.t_block, 3))
        check(lib().mdx_msd_result(self.handle, _ptr(msd), _ptr(traj)))
        return msd, traj

    def cross(self, pairs):
        """float64[n_pairs, n_blocks, t_block]: ``msd_fft(sum_i r, sum_j r)`` for every pair of groups, from the
        summed trajectories in HBM (``mdx_msd_cross``)."""
        p = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        out = np.empty((len(p), self.n_blocks, self.t_block))
        check(lib().mdx_msd_cross(self.handle, _ptr(p), len(p), _ptr(out)))
        return out

    def result_acf(self):
        """float64[n_groups, n_blocks, t_block]: sum over the pushed entities and dimensions of
        ``sum_k x(k) x(k+m)`` (the un-normalised vector ACF; ``mdx_msd_result_acf``)."""
        acf = np.zeros((self.n_groups, self.n_blocks, self.t_block))
        check(lib().mdx_msd_result_acf(self.handle, _ptr(acf)))
        return acf

    def reset(self):
        check(lib().mdx_msd_reset(self.handle))

    def synchronize(self):
        """Wait for everything queued on the engine's stream."""
        check(lib().mdx_msd_stats(self.handle, None, None, None))

    def allreduce(self, comm: RcclComm):
        check(lib().mdx_msd_allreduce(self.handle, comm.handle))

    def stats(self):
        n, ms, nb = c_int64(), c_double(), c_int64()
        check(lib().mdx_msd_stats(self.handle, byref(n), byref(ms), byref(nb)))
        return {"launches": n.value, "kernel_ms": ms.value, "bytes_moved": nb.value}


def correlate_device(a, b=None, *, negative=False, dev=0):
    """
    ``mdx_correlate``: un-normalised linear correlations of real series.

    a, b : float64[n_series, n_t].  Returns ``pos[n_series, n_t]`` with
    ``pos[s, m] = sum_k a[s, k] b[s, k+m]`` and, when ``negative``, also
    ``neg[s, m] = sum_k a[s, k+m] b[s, k]``.
    """
    a = np.ascontiguousarray(a, dtype=np.float64)
    n_series, n_t = a.shape
    bb = None
    if b is not None:
        bb = np.ascontiguousarray(b, dtype=np.float64)
        if bb.shape != a.shape:
            raise ValueError("The arrays must have the same dimensions.")
    out = np.zeros_like(a)
    neg = np.zeros_like(a) if negative else None
    check(lib().mdx_correlate(dev, _ptr(a), _ptr(bb), n_series, n_t, _ptr(out), _ptr(neg)))
    return (out, neg) if negative else out
