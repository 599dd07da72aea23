"""
Runtime entry points of libmdx.so that move and hold device memory: uploads through the pinned ring (whole
arrays and strided rows, pageable and page-locked memory), the per-device block cache behind mdx_malloc /
mdx_free / the handles' buffers, non-owning windows into device arrays.
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from mdhelper_amd import _core, _lib  # noqa: E402
from mdhelper_amd._lib import check, lib  # noqa: E402


@pytest.mark.parametrize("pinned", [False, True])
def test_upload_and_strided_rows_round_trip(pinned):
    rng = np.random.default_rng(3)
    T, N = 700, 4099                                   # 34 MB: several 16 MB ring chunks, ragged tails
    h = rng.normal(size=(T, N, 3)).astype(np.float32)
    if pinned:
        check(lib().mdx_host_register(0, h.ctypes.data, h.nbytes))
    try:
        d = _core.DeviceArray.upload(h)
        assert np.array_equal(d.to_host(), h)
        d.free()
        # (pageable: 12 KB rows 37 KB apart are page-locked slice by slice and read by 2-D DMA; rows 1.2 KB apart from
        # the next one, short rows and single coordinates are gathered into the pinned ring by the copy threads)
        for first, count in ((0, N), (17, 1000), (50, 4000), (4000, 99), (5, 1)):
            buf = _core.DeviceArray((T, count, 3), np.float32)
            buf.upload_columns(h, first, count)
            assert np.array_equal(buf.to_host(), h[:, first:first + count]), (first, count)
            buf.free()
        with pytest.raises(ValueError):
            _core.DeviceArray((T, 10, 3), np.float32).upload_columns(h, N - 5, 10)
    finally:
        if pinned:
            check(lib().mdx_host_unregister(0, h.ctypes.data))


def test_strided_rows_out_of_a_file_mapping_take_the_ring(tmp_path):
    """Rows of a numpy.memmap (file-backed pages) are never page-locked for the DMA engine: a registration on pages
    another program can truncate blocks every later GPU operation of the process for as long as it exists
    (scripts/diag/mmap_truncate_probe.py).  They go through the copy threads and the pinned ring; the file may lose its
    tail afterwards and the device stays usable."""
    import os
    T, N = 300, 4099
    rng = np.random.default_rng(5)
    path = tmp_path / "frames.bin"
    ref = rng.normal(size=(T, N, 3)).astype(np.float32)
    ref.tofile(path)
    h = np.memmap(path, dtype=np.float32, mode="r", shape=(T, N, 3))
    buf = _core.DeviceArray((T, 1000, 3), np.float32)
    buf.upload_columns(h, 17, 1000)                    # 12 KB rows: locked slices if the memory were anonymous
    assert np.array_equal(buf.to_host(), ref[:, 17:1017])
    del h
    os.truncate(path, os.path.getsize(path) // 2)
    again = _core.DeviceArray.from_host(np.arange(1000.0))
    assert np.array_equal(again.to_host(), np.arange(1000.0))
    buf.free()
    again.free()


def test_a_range_registered_too_short_goes_through_the_ring():
    """memory whose START is page-locked but whose end is not must not be handed to the DMA engine.
    (A page-aligned buffer, whole pages registered; the registration ends before the buffer does.)"""
    import mmap
    n = 24 << 20
    mm = mmap.mmap(-1, n)
    h = np.frombuffer(mm, dtype=np.float32)
    h[:] = np.arange(h.size, dtype=np.float32)
    check(lib().mdx_host_register(0, h.ctypes.data, 1 << 20))      # only the first MiB
    try:
        d = _core.DeviceArray.upload(h)
        got = d.to_host()
        d.free()
    finally:
        _core.synchronize(0)
        check(lib().mdx_host_unregister(0, h.ctypes.data))
    assert np.array_equal(got, h)
    del h
    mm.close()


def test_block_cache_reuses_and_trims():
    freed = ctypes.c_size_t()
    check(lib().mdx_trim_cache(0, ctypes.byref(freed)))
    a = _core.DeviceArray((48 << 20,), np.uint8)
    ptr = a.ptr.value
    a.free()                                           # goes to the cache
    b = _core.DeviceArray((40 << 20,), np.uint8)       # fits the cached 48 MiB block (<= 2 x the request)
    assert b.ptr.value == ptr
    b.free()
    c = _core.DeviceArray((4 << 20,), np.uint8)        # too small for it: a block of its own
    assert c.ptr.value != ptr
    c.free()
    check(lib().mdx_trim_cache(0, ctypes.byref(freed)))
    assert freed.value >= (48 << 20)
    check(lib().mdx_trim_cache(0, ctypes.byref(freed)))
    assert freed.value == 0
    # small allocations bypass the cache
    s = _core.DeviceArray((1000,), np.uint8)
    s.free()
    check(lib().mdx_trim_cache(0, ctypes.byref(freed)))
    assert freed.value == 0


def test_device_array_windows_do_not_own_their_memory():
    h = np.arange(5 * 7 * 3, dtype=np.float64).reshape(5, 7, 3)
    d = _core.DeviceArray.from_host(h)
    w = d.rows(1, 3)
    assert w.shape == (3, 7, 3) and np.array_equal(w.to_host(), h[1:4])
    v = _core.DeviceArray.view(d, (2, 7, 3))
    assert np.array_equal(v.to_host(), h[:2])
    w.free()
    v.free()
    assert np.array_equal(d.to_host(), h)              # still there
    with pytest.raises(IndexError):
        d.rows(3, 4)
    with pytest.raises(ValueError):
        _core.DeviceArray.view(d, (6, 7, 3))
    d.free()


def test_engine_cross_msd_equals_the_oracle():
    """mdx_msd_cross: msd_fft(sum_i r, sum_j r) of every pair of groups from the summed trajectories in HBM
    (reference correlation.py:461-668 as transport.py:1034, 1052 call it), blocks and a zeroed dimension."""
    from oracle import correlation as oc
    rng = np.random.default_rng(12)
    B, Tb, sizes = 3, 500, (7, 4, 9)
    pos = np.cumsum(rng.normal(0, 0.3, (B * Tb, sum(sizes), 3)), axis=0) + rng.uniform(0, 20, (1, sum(sizes), 3))
    eng = _core.MsdEngine(Tb, B, 3)
    first = 0
    for g, n in enumerate(sizes):
        eng.push(g, pos, first, n, zero_dims=2)
        first += n
    pairs = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    got = eng.cross(pairs)
    _msd, traj = eng.result()
    eng.close()
    p = pos.copy()
    p[..., 1] = 0.0
    sums, first = [], 0
    for n in sizes:
        sums.append(p[:, first:first + n].sum(axis=1).reshape(B, Tb, 3))
        first += n
    assert np.allclose(traj, np.stack(sums), rtol=1e-12, atol=1e-9)
    for k, (i, j) in enumerate(pairs):
        want = oc.msd_fft_ref(sums[i], axis=1) if i == j else oc.msd_fft_ref(sums[i], sums[j], axis=1)
        scale = np.sqrt(np.abs(oc.msd_fft_ref(sums[i], axis=1) * oc.msd_fft_ref(sums[j], axis=1))).max()
        assert np.allclose(got[k], want, rtol=1e-8, atol=1e-9 * scale), (i, j)


def test_host_registration_is_page_aligned_tracked_and_exclusive():
    """registrations are widened to whole pages and tracked; a range that shares a page with a live
    registration is refused, an unknown pointer cannot be unregistered, and a buffer that starts in the middle of
    a page (what malloc / numpy hand out) registers, uploads by DMA and unregisters cleanly."""
    raw = np.zeros((6 << 20) + 4096, dtype=np.uint8)
    off = (-raw.ctypes.data) % 4096 + 40                  # 40 bytes into a page
    h = raw[off:off + (6 << 20)].view(np.float32)
    h[:] = np.arange(h.size, dtype=np.float32)
    assert h.ctypes.data % 4096 == 40
    check(lib().mdx_host_register(0, h.ctypes.data, h.nbytes))
    try:
        # the same pages again, and a neighbour inside the first page: refused while the first one lives
        with pytest.raises(RuntimeError, match="shares pages"):
            check(lib().mdx_host_register(0, h.ctypes.data, h.nbytes))
        with pytest.raises(RuntimeError, match="shares pages"):
            check(lib().mdx_host_register(0, h.ctypes.data - 32, 16))
        with pytest.raises(ValueError, match="was not registered"):
            check(lib().mdx_host_unregister(0, h.ctypes.data + 4096))
        d = _core.DeviceArray.upload(h)
        assert np.array_equal(d.to_host(), h)
        d.free()
    finally:
        check(lib().mdx_host_unregister(0, h.ctypes.data))
    with pytest.raises(ValueError, match="was not registered"):
        check(lib().mdx_host_unregister(0, h.ctypes.data))
    # the pages are ordinary pageable memory again: a fresh registration of an overlapping range works
    check(lib().mdx_host_register(0, raw.ctypes.data, raw.nbytes))
    check(lib().mdx_host_unregister(0, raw.ctypes.data))


def _page_aligned(shape, dtype=np.float64):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    raw = np.empty(n + 2 * 4096, dtype=np.uint8)
    off = -raw.ctypes.data % 4096
    return raw[off:off + n].view(dtype).reshape(shape)


@pytest.mark.parametrize("pinned", [False, True])
def test_cross_correlation_of_large_inputs_is_ordered(pinned):
    """mdx_correlate with inputs >= 1 MiB each — `b` used to be uploaded, on another stream, into the
    buffer the padding kernel of `a` was still reading (page-locked `b`: the DMA starts at once).  Both inputs have
    their own buffers now and the whole call is ordered on one stream.  Checked against the oracle's restatement
    of correlation.py:185-197 for cross-correlations, positive and negative lags."""
    from oracle import correlation as oc
    rng = np.random.default_rng(41)
    n_series, n_t = 96, 4096                               # 3 MiB per input
    # (page-aligned with a page of slack behind: two malloc'd arrays can share a page, and a registration takes
    # whole pages exclusively — the second mdx_host_register is then refused, as the header says)
    a, b = _page_aligned((n_series, n_t)), _page_aligned((n_series, n_t))
    a[:] = np.cumsum(rng.normal(size=(n_series, n_t)), axis=1)
    b[:] = rng.normal(size=(n_series, n_t))
    if pinned:
        check(lib().mdx_host_register(0, a.ctypes.data, a.nbytes))
        check(lib().mdx_host_register(0, b.ctypes.data, b.nbytes))
    try:
        for _ in range(3):
            pos, neg = _core.correlate_device(a, b, negative=True)
    finally:
        if pinned:
            check(lib().mdx_host_unregister(0, a.ctypes.data))
            check(lib().mdx_host_unregister(0, b.ctypes.data))
    w = n_t - np.arange(n_t)
    full = oc.correlation_fft_ref(a.T, b.T, axis=0)       # [2 n_t - 1, n_series], lags -(n_t-1) .. n_t-1, per-lag mean
    want_pos = full[n_t - 1:].T * w
    want_neg = full[n_t - 1::-1].T * w
    scale = np.abs(want_pos).max()
    assert np.allclose(pos, want_pos, rtol=1e-9, atol=1e-10 * scale)
    assert np.allclose(neg, want_neg, rtol=1e-9, atol=1e-10 * scale)


def test_cross_msd_of_more_rows_than_one_grid_holds():
    """mdx_msd_cross batches its (pair, block) rows: 13 groups at 1 000 blocks = 91 pairs x 1 000 rows
    (> 65 535, the grid's y extent) used to be refused.  Spot rows against the oracle."""
    from oracle import correlation as oc
    rng = np.random.default_rng(43)
    B, Tb, G = 1000, 6, 13
    pos = np.cumsum(rng.normal(0, 0.3, (B * Tb, G, 3)), axis=0)
    eng = _core.MsdEngine(Tb, B, G)
    for g in range(G):
        eng.push(g, pos, g, 1)
    pairs = [(i, j) for i in range(G) for j in range(i, G)]
    assert len(pairs) * B > 65535
    got = eng.cross(pairs)
    eng.close()
    sums = [pos[:, g].reshape(B, Tb, 3) for g in range(G)]
    for k in (0, 1, 12, 13, 45, 90):
        i, j = pairs[k]
        want = oc.msd_fft_ref(sums[i], axis=1) if i == j else oc.msd_fft_ref(sums[i], sums[j], axis=1)
        assert np.allclose(got[k], want, rtol=1e-8, atol=1e-9 * np.abs(want).max()), (i, j)
