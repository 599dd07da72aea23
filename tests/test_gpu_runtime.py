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
        for first, count in ((0, N), (17, 1000), (4000, 99), (5, 1)):
            buf = _core.DeviceArray((T, count, 3), np.float32)
            buf.upload_columns(h, first, count)
            assert np.array_equal(buf.to_host(), h[:, first:first + count]), (first, count)
            buf.free()
        with pytest.raises(ValueError):
            _core.DeviceArray((T, 10, 3), np.float32).upload_columns(h, N - 5, 10)
    finally:
        if pinned:
            check(lib().mdx_host_unregister(0, h.ctypes.data))


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
