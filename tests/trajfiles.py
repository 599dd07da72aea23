"""
Writers for the trajectory-file tests: AMBER-convention NetCDF through
``scipy.io.netcdf_file`` (an independent NetCDF-3 implementation) and a DCD
writer restating the CHARMM/NAMD record layout.
"""

import struct

import numpy as np
from scipy.io import netcdf_file


def write_amber_netcdf(path, positions, lengths=None, angles=None, times=None, *, version=2,
                       velocities=False, cell_float=False):
    """positions float[F, N, 3]; version 1 = classic, 2 = 64-bit offset (the reference's)."""
    positions = np.asarray(positions, dtype=np.float32)
    F, N, _ = positions.shape
    with netcdf_file(path, "w", version=version) as nc:
        nc.Conventions = "AMBER"
        nc.ConventionVersion = "1.0"
        nc.program = "mdhelper_amd tests"
        nc.createDimension("frame", None)
        nc.createDimension("spatial", 3)
        nc.createDimension("atom", N)
        nc.createDimension("cell_spatial", 3)
        nc.createDimension("cell_angular", 3)
        nc.createDimension("label", 5)
        sp = nc.createVariable("spatial", "c", ("spatial",))
        sp[:] = np.array(list("xyz"), dtype="S1")
        t = nc.createVariable("time", "f", ("frame",))
        t.units = "picosecond"
        xyz = nc.createVariable("coordinates", "f", ("frame", "atom", "spatial"))
        xyz.units = "angstrom"
        if lengths is not None:
            cl = nc.createVariable("cell_lengths", "f" if cell_float else "d", ("frame", "cell_spatial"))
            cl.units = "angstrom"
            ca = nc.createVariable("cell_angles", "f" if cell_float else "d", ("frame", "cell_angular"))
            ca.units = "degree"
        if velocities:
            v = nc.createVariable("velocities", "f", ("frame", "atom", "spatial"))
            v.units = "angstrom/picosecond"
        for f in range(F):
            t[f] = float(f) * 0.5 if times is None else times[f]
            xyz[f] = positions[f]
            if lengths is not None:
                cl[f] = np.broadcast_to(lengths, (F, 3))[f]
                ca[f] = (90.0, 90.0, 90.0) if angles is None else np.broadcast_to(angles, (F, 3))[f]
            if velocities:
                v[f] = -positions[f]


def write_dcd(path, positions, unitcell=None, *, big_endian=False, istart=0, nsavc=1, delta=0.02,
              cosines=False, titles=("written by the mdhelper_amd tests",)):
    """
    positions float[F, N, 3]; unitcell float[F or 1, 6] as (lx, ly, lz, alpha, beta, gamma) or None.
    CHARMM-format header (version word 24).
    """
    e = ">" if big_endian else "<"
    positions = np.asarray(positions, dtype=np.float32)
    F, N, _ = positions.shape
    icntrl = [0] * 20
    icntrl[0], icntrl[1], icntrl[2], icntrl[3] = F, istart, nsavc, F * nsavc
    icntrl[9] = struct.unpack(e + "i", struct.pack(e + "f", delta))[0]
    icntrl[10] = 1 if unitcell is not None else 0
    icntrl[19] = 24
    with open(path, "wb") as fh:
        fh.write(struct.pack(e + "i4s20ii", 84, b"CORD", *icntrl, 84))
        body = struct.pack(e + "i", len(titles)) + b"".join(t.encode().ljust(80)[:80] for t in titles)
        fh.write(struct.pack(e + "i", len(body)) + body + struct.pack(e + "i", len(body)))
        fh.write(struct.pack(e + "iii", 4, N, 4))
        if unitcell is not None:
            unitcell = np.broadcast_to(np.asarray(unitcell, dtype=np.float64), (F, 6))
        for f in range(F):
            if unitcell is not None:
                lx, ly, lz, al, be, ga = unitcell[f]
                if cosines:
                    al, be, ga = (np.cos(np.radians(x)) for x in (al, be, ga))
                fh.write(struct.pack(e + "i6di", 48, lx, ga, ly, be, al, lz, 48))
            for k in range(3):
                plane = positions[f, :, k].astype(e + "f4").tobytes()
                fh.write(struct.pack(e + "i", 4 * N) + plane + struct.pack(e + "i", 4 * N))


class PerFrameTrajectory:
    """Wraps an in-memory trajectory and hides its block interface, so that an analysis takes the
    generic frame-by-frame protocol (host NumPy frame preparation): the reference for the batched
    and device-prepared paths."""

    def __init__(self, traj):
        self._traj = traj

    def __getattr__(self, name):
        if name in ("frame_block", "box_block", "native", "_positions", "device_block", "device_array"):
            raise AttributeError(name)
        return getattr(self._traj, name)

    def __getitem__(self, item):
        return self._traj[item]

    def __len__(self):
        return len(self._traj)


def per_frame(analysis):
    """Route an analysis object built on an in-memory universe through the per-frame protocol."""
    analysis._trajectory = PerFrameTrajectory(analysis._trajectory)
    return analysis
