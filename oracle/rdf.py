"""
oracle.rdf — CPU restatement of the radial-histogram / RDF path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``): never imported by the
product package.

Follows, line by line:

* ``radial_histogram``            reference ``src/mdhelper/analysis/structure.py:32-104``
* ``RadialDistributionFunction``  ``_prepare :734-748``, ``_single_frame :750-791``,
                                  ``_conclude :837-862`` (``rdf_normalise`` below)

The pair search of the reference is ``MDAnalysis.lib.distances.capped_distance``
(third party, absent here, version unpinned ``mdanalysis >= 2.2.0``).  Its
published brute-force / orthorhombic algorithm
(``MDAnalysis/lib/src/calc_distances.h``: ``_calc_distance_array_ortho`` +
``minimum_image``; ``MDAnalysis/lib/distances.py``: ``_bruteforce_capped``) is
restated in ``pair_distances`` — **parity unpinned at the ULP level** for this
one step; everything after it (exclusion mask, ``numpy.histogram``) is the
reference's literal code path.
"""

from __future__ import annotations

import numpy as np

_EPS = np.finfo(np.float64).eps


def _c_round(s: np.ndarray) -> np.ndarray:
    """C99 ``round``: nearest integer, halfway cases away from zero."""
    t = np.trunc(s)
    return np.where(np.abs(s - t) == 0.5, t + np.sign(s), np.rint(s))


def check_box(dims):
    """float32 box[6] as MDAnalysis stores it."""
    if dims is None:
        return None
    box = np.asarray(dims, dtype=np.float32).reshape(-1)
    if box.shape[0] != 6:
        raise ValueError("dims must have six entries (lx, ly, lz, alpha, beta, gamma).")
    return box


def is_triclinic(box) -> bool:
    return box is not None and not np.all(box[3:] == 90.0)


def triclinic_vectors(box) -> np.ndarray:
    """
    float32[3, 3] cell matrix (rows a, b, c; lower triangular) of (lx, ly, lz, alpha, beta, gamma),
    restating ``MDAnalysis.lib.mdamath.triclinic_vectors`` (third party, absent; float64
    arithmetic through libm, entries stored as float32, exact zeros for right angles).
    """
    import math
    lx, ly, lz = (float(x) for x in box[:3])
    ca = 0.0 if box[3] == 90.0 else math.cos(math.radians(1.0) * float(box[3]))
    cb = 0.0 if box[4] == 90.0 else math.cos(math.radians(1.0) * float(box[4]))
    cg = 0.0 if box[5] == 90.0 else math.cos(math.radians(1.0) * float(box[5]))
    sg = 1.0 if box[5] == 90.0 else math.sin(math.radians(1.0) * float(box[5]))
    B = np.zeros((3, 3), dtype=np.float32)
    B[0, 0] = lx
    B[1, 0] = ly * cg
    B[1, 1] = ly * sg
    cx = lz * cb
    cy = lz * (ca - cb * cg) / sg
    B[2, 0] = cx
    B[2, 1] = cy
    B[2, 2] = math.sqrt(lz * lz - cx * cx - cy * cy)
    return B


def triclinic_wrap(pos, B) -> np.ndarray:
    """Coordinates moved into the central cell, c then b then a axis, in double; float32 out."""
    r = np.asarray(pos, dtype=np.float32).reshape(-1, 3).astype(np.float64)
    Bd = B.astype(np.float64)
    for k in (2, 1, 0):
        s = np.floor(r[:, k] / Bd[k, k])
        for c in range(k + 1):
            r[:, c] -= s * Bd[k, c]
    return r.astype(np.float32)


def pair_distances_triclinic(pos1, pos2, B) -> np.ndarray:
    """
    Minimum over the 27 neighbouring images of the float32 difference of WRAPPED coordinates
    (``MDAnalysis/lib/src/calc_distances.h``: ``minimum_image_triclinic``; published algorithm
    restated — parity unpinned): ix outermost, first strictly smaller squared length wins,
    ``rsq = (x*x + y*y) + z*z`` in double.
    """
    ref = np.ascontiguousarray(pos1, dtype=np.float32).reshape(-1, 3)
    conf = np.ascontiguousarray(pos2, dtype=np.float32).reshape(-1, 3)
    dx = (conf[None, :, :] - ref[:, None, :]).astype(np.float64)
    Bd = B.astype(np.float64)
    best = np.full(dx.shape[:2], 1.0e300)
    for ix in (-1, 0, 1):
        rx = dx[..., 0] + Bd[0, 0] * ix
        for iy in (-1, 0, 1):
            ry0 = rx + Bd[1, 0] * iy
            ry1 = dx[..., 1] + Bd[1, 1] * iy
            for iz in (-1, 0, 1):
                rz0 = ry0 + Bd[2, 0] * iz
                rz1 = ry1 + Bd[2, 1] * iz
                rz2 = dx[..., 2] + Bd[2, 2] * iz
                dsq = (rz0 * rz0 + rz1 * rz1) + rz2 * rz2
                best = np.where(dsq < best, dsq, best)
    return np.sqrt(best)


def pair_distances(pos1: np.ndarray, pos2: np.ndarray, box) -> np.ndarray:
    """
    float64 minimum-image distances d[i, j] between float32 coordinates.

    Arithmetic contract, per component k::

        dx  = (double)(conf[j][k] - ref[i][k])        # float32 subtract, widened
        inv = (float)(1.0 / box[k])                    # float32 inverse box
        s   = inv * dx                                 # double
        dx  = box[k] * (s - round(s))                  # double, C round()
        rsq = (dx*dx + dy*dy) + dz*dz                  # double, no FMA
        d   = sqrt(rsq)
    """
    ref = np.ascontiguousarray(pos1, dtype=np.float32).reshape(-1, 3)
    conf = np.ascontiguousarray(pos2, dtype=np.float32).reshape(-1, 3)
    d32 = conf[None, :, :] - ref[:, None, :]            # float32 arithmetic
    dx = d32.astype(np.float64)
    if box is not None:
        b = box[:3].astype(np.float64)
        inv = (1.0 / b).astype(np.float32).astype(np.float64)
        s = inv * dx
        dx = b * (s - _c_round(s))
    sq = dx * dx
    rsq = (sq[..., 0] + sq[..., 1]) + sq[..., 2]
    return np.sqrt(rsq)


def radial_histogram_ref(pos1, pos2, n_bins, range, dims, *, exclusion=None,
                         chunk_pairs: int = 4_000_000, i_offset: int = 0) -> np.ndarray:
    """
    int64[n_bins] histogram of in-range ordered pair distances.

    Mirrors ``structure.py:92-104``: capped pairs with
    ``range[0] - eps < d <= range[1]``, optional exclusion
    ``i // e0 != j // e1``, then ``numpy.histogram(dist, n_bins, range)``.
    ``i_offset``: index of ``pos1[0]`` in the full first set when ``pos1`` is a block of its rows
    (the exclusion rule counts rows of the full set).
    """
    box = check_box(dims)
    ref = np.ascontiguousarray(pos1, dtype=np.float32).reshape(-1, 3)
    conf = np.ascontiguousarray(pos2, dtype=np.float32).reshape(-1, 3)
    tri = is_triclinic(box)
    if tri:
        B = triclinic_vectors(box)
        ref, conf = triclinic_wrap(ref, B), triclinic_wrap(conf, B)
    n1, n2 = ref.shape[0], conf.shape[0]
    counts = np.zeros(n_bins, dtype=np.int64)
    if n1 == 0 or n2 == 0:
        return counts
    max_cut = np.float64(range[1])
    min_cut = np.float64(range[0]) - _EPS
    rows = max(1, int(chunk_pairs // max(n2, 1)))
    j_idx = np.arange(n2)
    for lo in np.arange(0, n1, rows):
        hi = min(n1, lo + rows)
        d = (pair_distances_triclinic(ref[lo:hi], conf, B) if tri
             else pair_distances(ref[lo:hi], conf, box))
        keep = (d <= max_cut) & (d > min_cut)
        if exclusion is not None:
            i_idx = np.arange(lo, hi) + i_offset
            keep &= (i_idx[:, None] // exclusion[0]) != (j_idx[None, :] // exclusion[1])
        counts += np.histogram(d[keep], bins=n_bins, range=range)[0]
    return counts


def rdf_edges(n_bins, range):
    """``_prepare`` (structure.py:737-739)."""
    edges = np.linspace(*range, n_bins + 1)
    bins = (edges[:-1] + edges[1:]) / 2
    return edges, bins


def rdf_normalise(counts, edges, n_frames, n1, n2, volume_sum, *, norm="rdf",
                  exclusion=None, drop_axis=None):
    """``_conclude`` (structure.py:846-862)."""
    nrm = n_frames
    if norm is not None:
        if drop_axis is None:
            nrm = nrm * (4 * np.pi * np.diff(edges ** 3) / 3)
        else:
            nrm = nrm * (np.pi * np.diff(edges ** 2))
        if norm == "rdf":
            _n2 = n2
            if exclusion:
                _n2 -= exclusion[1]
            nrm = nrm * (n1 * _n2 * n_frames / volume_sum)
    return counts / nrm


def rdf_run_ref(frames, boxes, n_bins=201, range=(0.0, 15.0), *, sel1=None,
                sel2=None, exclusion=None, norm="rdf"):
    """
    Serial driver restating ``_prepare/_single_frame/_conclude`` for
    ``groupings="atoms"``, no ``drop_axis``, no ``n_batches``.

    frames : float32[F, N, 3];  boxes : float32[F, 6] (or [6]).
    """
    frames = np.asarray(frames, dtype=np.float32)
    boxes = np.broadcast_to(np.asarray(boxes, dtype=np.float32), (frames.shape[0], 6))
    edges, bins = rdf_edges(n_bins, range)
    counts = np.zeros(n_bins, dtype=np.int64)
    vol = 0.0
    sel1 = slice(None) if sel1 is None else sel1
    sel2 = sel1 if sel2 is None else sel2
    for f in np.arange(frames.shape[0]):
        dims = boxes[f]
        vol += float(np.prod(dims[:3].astype(np.float64)))
        counts += radial_histogram_ref(frames[f][sel1], frames[f][sel2], n_bins, range,
                                       dims, exclusion=exclusion)
    n1 = frames[0][sel1].shape[0]
    n2 = frames[0][sel2].shape[0]
    rdf = rdf_normalise(counts, edges, frames.shape[0], n1, n2, vol, norm=norm,
                        exclusion=exclusion)
    return {"edges": edges, "bins": bins, "counts": counts, "rdf": rdf, "volume": vol}
