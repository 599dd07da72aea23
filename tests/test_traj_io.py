"""
Native trajectory readers (csrc/mdx_traj.hip) against files written by an independent
NetCDF-3 implementation (scipy.io.netcdf_file) and by the DCD writer of tests/trajfiles.py.
Host-side parsing and reads only: runs without a GPU.
"""

import pathlib

import numpy as np
import pytest

from mdhelper_amd.io import FileUniverse, TrajectoryFile
from trajfiles import write_amber_netcdf, write_dcd


def _walk(F, N, L, seed):
    rng = np.random.default_rng(seed)
    return (rng.uniform(0, L, (1, N, 3)) + np.cumsum(rng.normal(0, 0.3, (F, N, 3)), axis=0)).astype(np.float32)


@pytest.mark.parametrize("version", [1, 2])
@pytest.mark.parametrize("velocities", [False, True])
def test_netcdf_matches_what_was_written(tmp_path, version, velocities):
    pos = _walk(7, 37, 12.0, 1)
    lengths = np.array([[12.0 + 0.01 * f, 13.0, 14.5] for f in range(7)])
    path = tmp_path / "t.nc"
    write_amber_netcdf(path, pos, lengths, version=version, velocities=velocities)
    t = TrajectoryFile(path)
    assert (t.n_frames, t.n_atoms, t.has_box, t.has_time, t.format) == (7, 37, True, True, "NETCDF")
    frames = [0, 6, 3, 3]
    np.testing.assert_array_equal(t.read_positions(frames), pos[frames])
    boxes = t.read_boxes(frames)
    np.testing.assert_array_equal(boxes[:, :3], lengths[frames].astype(np.float32))
    np.testing.assert_array_equal(boxes[:, 3:], 90.0)
    np.testing.assert_array_equal(t.read_times(frames), np.float32(0.5) * np.array(frames))
    t.close()


def test_netcdf_single_record_variable_and_float_cell(tmp_path):
    # no cell, no velocities: 'time' + 'coordinates' only; then a float32 cell
    pos = _walk(3, 5, 4.0, 2)
    write_amber_netcdf(tmp_path / "a.nc", pos, None)
    t = TrajectoryFile(tmp_path / "a.nc")
    assert not t.has_box
    np.testing.assert_array_equal(t.read_positions([2, 0]), pos[[2, 0]])
    with pytest.raises(RuntimeError):
        t.read_boxes([0])
    write_amber_netcdf(tmp_path / "b.nc", pos, (4.0, 5.0, 6.0), cell_float=True)
    t = TrajectoryFile(tmp_path / "b.nc")
    np.testing.assert_array_equal(t.read_boxes([1])[0], [4, 5, 6, 90, 90, 90])


def test_netcdf_large_header(tmp_path):
    # an attribute larger than the first header prefix the parser reads
    from scipy.io import netcdf_file
    pos = _walk(2, 4, 3.0, 3)
    path = tmp_path / "big.nc"
    write_amber_netcdf(path, pos, (3.0, 3.0, 3.0))
    with netcdf_file(path, "a") as nc:
        nc.history = "x" * 300_000
    t = TrajectoryFile(path)
    np.testing.assert_array_equal(t.read_positions([0, 1]), pos)


@pytest.mark.parametrize("big_endian", [False, True])
@pytest.mark.parametrize("cell", [None, "degrees", "cosines"])
def test_dcd_matches_what_was_written(tmp_path, big_endian, cell):
    pos = _walk(5, 23, 9.0, 4)
    unitcell = None if cell is None else np.array([[9.0 + f, 10.0, 11.0, 90.0, 90.0, 90.0] for f in range(5)])
    path = tmp_path / "t.dcd"
    write_dcd(path, pos, unitcell, big_endian=big_endian, cosines=(cell == "cosines"),
              istart=100, nsavc=10, delta=0.5)
    t = TrajectoryFile(path)
    assert (t.n_frames, t.n_atoms, t.format) == (5, 23, "DCD")
    assert t.has_box == (cell is not None)
    frames = [4, 0, 2]
    np.testing.assert_array_equal(t.read_positions(frames), pos[frames])
    if cell is not None:
        np.testing.assert_allclose(t.read_boxes(frames), unitcell[frames], rtol=0, atol=1e-5)
    np.testing.assert_allclose(t.read_times(frames), (100 + 10 * np.array(frames)) * 0.5 * 4.888821e-2)


def test_dcd_triclinic_angles_roundtrip(tmp_path):
    pos = _walk(2, 3, 5.0, 5)
    cellv = np.array([[8.0, 9.0, 10.0, 75.0, 80.0, 110.0]])
    for cosines in (False, True):
        write_dcd(tmp_path / "tri.dcd", pos, cellv, cosines=cosines)
        got = TrajectoryFile(tmp_path / "tri.dcd").read_boxes([0, 1])
        np.testing.assert_allclose(got, np.repeat(cellv, 2, 0), atol=2e-5)


def test_errors(tmp_path):
    with pytest.raises(OSError):
        TrajectoryFile(tmp_path / "missing.nc")
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"\x00" * 4096)
    with pytest.raises(ValueError):
        TrajectoryFile(bad)
    hdf = tmp_path / "h.nc"
    hdf.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\x00" * 100)
    with pytest.raises(NotImplementedError):
        TrajectoryFile(hdf)
    pos = _walk(4, 6, 3.0, 6)
    write_amber_netcdf(tmp_path / "ok.nc", pos, (3.0, 3.0, 3.0))
    t = TrajectoryFile(tmp_path / "ok.nc")
    with pytest.raises(ValueError):
        t.read_positions([4])
    with pytest.raises(ValueError):
        t.read_positions([-1])
    # truncated file: the declared frames do not fit
    data = (tmp_path / "ok.nc").read_bytes()
    (tmp_path / "cut.nc").write_bytes(data[:-40])
    with pytest.raises(OSError):
        TrajectoryFile(tmp_path / "cut.nc")


def test_file_universe_surface(tmp_path):
    pos = _walk(6, 11, 7.0, 7)
    write_amber_netcdf(tmp_path / "u.nc", pos, (7.0, 7.5, 8.0), times=np.arange(6) * 2.0)
    u = FileUniverse(tmp_path / "u.nc")
    assert u.trajectory.n_frames == 6 and u.atoms.n_atoms == 11 and u.trajectory.dt == 2.0
    ts = u.trajectory[3]
    np.testing.assert_array_equal(ts.positions, pos[3])
    np.testing.assert_array_equal(u.atoms[[1, 4]].positions, pos[3][[1, 4]])
    np.testing.assert_array_equal(ts.dimensions, [7, 7.5, 8, 90, 90, 90])
    assert ts.volume == pytest.approx(7 * 7.5 * 8)
    sel = u.trajectory[1:6:2]
    assert [t.frame for t in sel] == [1, 3, 5]
    np.testing.assert_array_equal(u.trajectory.frame_block([5, 0]), pos[[5, 0]])


def test_mutated_headers_fail_cleanly(tmp_path):
    """Random corruption of the first kilobyte: every file is either read or refused with a
    Python exception — never a crash, hang or out-of-bounds read."""
    pos = _walk(3, 9, 5.0, 8)
    write_amber_netcdf(tmp_path / "good.nc", pos, (5.0, 5.0, 5.0))
    write_dcd(tmp_path / "good.dcd", pos, [[5.0, 5.0, 5.0, 90, 90, 90]])
    rng = np.random.default_rng(9)
    outcomes = {"ok": 0, "refused": 0}
    for name in ("good.nc", "good.dcd"):
        data = bytearray((tmp_path / name).read_bytes())
        for trial in range(150):
            bad = bytearray(data)
            for _ in range(int(rng.integers(1, 6))):
                at = int(rng.integers(0, min(len(bad), 1024)))
                bad[at] = int(rng.integers(0, 256))
            if trial % 5 == 0:
                bad = bad[:int(rng.integers(8, len(bad)))]
            path = tmp_path / f"m{trial}{name[-4:]}"
            path.write_bytes(bytes(bad))
            try:
                t = TrajectoryFile(path)
                if 0 < t.n_frames <= 1000 and 0 < t.n_atoms <= 100000:
                    t.read_positions([0, t.n_frames - 1])
                    if t.has_box:
                        t.read_boxes([t.n_frames - 1])
                    if t.has_time:
                        t.read_times([0])
                t.close()
                outcomes["ok"] += 1
            except (ValueError, OSError, NotImplementedError, RuntimeError, MemoryError):
                outcomes["refused"] += 1
    assert outcomes["ok"] + outcomes["refused"] == 300 and outcomes["refused"] > 20


def test_bench_fast_netcdf_writer_is_read_back_by_scipy_and_by_the_native_reader(tmp_path):
    """bench.py writes its NetCDF input as scipy's header + one big-endian structured array of
    records (scipy's per-record writes take minutes at bench sizes): scipy — an independent NetCDF-3
    implementation — and the native reader must both return what went in."""
    import sys
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
    import bench
    from scipy.io import netcdf_file
    rng = np.random.default_rng(5)
    pos = (rng.random((9, 1237, 3)) * 40).astype(np.float32)
    box = np.array([40, 41, 42, 90, 90, 90], dtype=np.float32)
    path = tmp_path / "fast.nc"
    bench.write_amber_netcdf_fast(str(path), pos, box)
    with netcdf_file(str(path), "r", mmap=False) as nc:
        assert np.array_equal(nc.variables["coordinates"][:], pos)
        assert np.array_equal(nc.variables["cell_lengths"][:], np.tile(box[:3].astype(np.float64), (9, 1)))
        assert np.array_equal(nc.variables["time"][:], np.arange(9, dtype=np.float32))
    t = TrajectoryFile(path)
    assert (t.n_frames, t.n_atoms) == (9, 1237)
    assert np.array_equal(t.read_positions(np.arange(9)), pos)
    assert np.array_equal(t.read_boxes([4])[0], box)
    t.close()
