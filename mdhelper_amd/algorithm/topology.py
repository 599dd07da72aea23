"""
Frame-preparation helpers ``unwrap``, ``unwrap_edge`` and ``wrap`` (reference
``src/mdhelper/algorithm/topology.py:294-383``, ``:385-529`` and ``:531-578``), used by
``Onsager`` before the positions are stored (reference analysis/transport.py:979-981,
997-1004) and by ``EndToEndVector`` for its reference frame (analysis/polymer.py:700-727).
O(N) per frame, host NumPy.
"""

from __future__ import annotations

import warnings

import numpy as np


def unwrap(positions, positions_old, dimensions, *, thresholds=None, images=None,
           in_place: bool = True):
    """
    Globally unwrap positions by tracking boundary crossings: a displacement of at
    least ``thresholds`` (default: half the shortest box length) since the previous
    frame counts as one crossing against its sign.
    """
    if thresholds is None:
        thresholds = np.min(dimensions) / 2
    if images is None:
        images = np.zeros_like(positions, dtype=int)
    if not in_place:
        positions = positions.copy()
        images = images.copy()
    dpos = positions - positions_old
    crossed = np.abs(dpos) >= thresholds
    images[crossed] -= np.sign(dpos[crossed]).astype(int)
    if in_place:
        positions_old[:] = positions
        positions += images * dimensions
        return None
    positions_old = positions.copy()
    positions += images * dimensions
    return positions, positions_old, images


def make_whole_images(universe, dimensions) -> np.ndarray:
    """
    Image flags that make every fragment of the CURRENT frame whole: int[n_atoms, 3] with
    ``positions + images * dimensions`` = what ``MDAnalysis.lib.mdamath.make_whole`` leaves in each
    fragment — the walk the reference takes before it stores the starting positions of
    ``Onsager(unwrap=True)`` (reference transport.py:936-941).

    Restated from MDAnalysis' published algorithm ([ext]; the package is absent here, so this
    step is parity-unpinned at the ULP level): breadth-first from the fragment's first atom along
    the bonds; an atom reached over a bond is placed at its placed neighbour + the minimum image of
    the bond vector.  Orthorhombic cells.  A universe without bonds gives zeros.
    """
    n = universe.atoms.n_atoms
    images = np.zeros((n, 3), dtype=int)
    bonds = getattr(universe, "_bonds", None)
    if bonds is None or len(bonds) == 0:
        return images
    L = np.asarray(dimensions, dtype=float)[:3]
    pos = np.asarray(universe.atoms.positions, dtype=float)
    # adjacency in CSR form, neighbours in index order
    both = np.concatenate((bonds, bonds[:, ::-1]))
    both = both[np.lexsort((both[:, 1], both[:, 0]))]
    start = np.searchsorted(both[:, 0], np.arange(n + 1))
    placed = np.zeros(n, dtype=bool)
    for root in range(n):
        if placed[root] or start[root] == start[root + 1]:
            continue
        placed[root] = True
        queue = [root]
        while queue:
            nxt = []
            for a in queue:
                for b in both[start[a]:start[a + 1], 1]:
                    if placed[b]:
                        continue
                    # bond vector between the STORED coordinates, folded to its minimum image; the
                    # neighbour then sits in the image of the atom it was reached from, shifted by
                    # the fold
                    d = pos[b] - pos[a]
                    images[b] = images[a] - np.rint(d / L).astype(int)
                    placed[b] = True
                    nxt.append(b)
            queue = nxt
    return images


def _minimum_image(vectors, dimensions):
    """Shortest periodic image of displacement vectors (``MDAnalysis.lib.distances.
    minimize_vectors`` as called at reference topology.py:497-500): orthorhombic cells by
    rounding, triclinic cells by searching the 27 neighbouring images of the rounded guess."""
    vectors = np.asarray(vectors, dtype=float)
    lengths = np.asarray(dimensions[:3], dtype=float)
    angles = np.asarray(dimensions[3:6], dtype=float) if len(dimensions) >= 6 else np.full(3, 90.0)
    if np.all(angles == 90.0):
        return vectors - lengths * np.round(vectors / lengths)
    al, be, ga = np.deg2rad(angles)
    a = np.array([lengths[0], 0.0, 0.0])
    b = lengths[1] * np.array([np.cos(ga), np.sin(ga), 0.0])
    cx, cy = np.cos(be), (np.cos(al) - np.cos(be) * np.cos(ga)) / np.sin(ga)
    c = lengths[2] * np.array([cx, cy, np.sqrt(max(1.0 - cx * cx - cy * cy, 0.0))])
    cell = np.stack([a, b, c])
    frac = vectors @ np.linalg.inv(cell)
    base = (frac - np.round(frac)) @ cell
    shifts = np.array([[i, j, k] for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1)],
                      dtype=float) @ cell
    cand = base[..., None, :] + shifts
    best = np.argmin((cand ** 2).sum(axis=-1), axis=-1)
    return np.take_along_axis(cand, best[..., None, None], axis=-2)[..., 0, :]


def unwrap_edge(*, group=None, positions=None, bonds=None, dimensions=None, thresholds=None,
                masses=None):
    """
    Locally unwrap molecules that straddle the cell boundary (reference topology.py:385-529):
    every atom is placed at the minimum image relative to a bonded atom that has already been
    placed (starting from the first atom of each molecule), then each molecule is shifted so
    that its centre of mass lies inside the cell.

    positions : float[N, 3] (modified in place and returned); bonds : int[N_bonds, 2] indices
    into ``positions``; dimensions : float[3] or float[6]; masses : float[N] or per molecule.
    ``group=`` (an AtomGroup carrying bond information) is not supported by this build's
    universe shim — pass ``positions`` and ``bonds``.
    """
    if group is not None:
        raise NotImplementedError("unwrap_edge(group=...) needs bond topology, which the "
                                  "array universes of this build do not carry; pass "
                                  "'positions', 'bonds' and 'dimensions'.")
    if positions is None:
        raise ValueError("Either 'group' or 'positions' must be specified.")
    if bonds is None:
        raise ValueError("Bond information must be specified in 'bonds'.")
    if dimensions is None:
        raise ValueError("System dimensions must be specified in 'dimensions'.")
    dimensions = np.asarray(dimensions, dtype=float)
    if len(dimensions) == 3:
        dimensions = np.concatenate((dimensions, (90.0, 90.0, 90.0)))

    n = len(positions)
    neighbours = [[] for _ in range(n)]
    for a, b in np.asarray(bonds, dtype=int).reshape(-1, 2):
        neighbours[a].append(b)
        neighbours[b].append(a)
    # molecules = connected components of the bonded atoms, each walked from its first atom
    seen = np.zeros(n, dtype=bool)
    molecules = []
    for root in range(n):
        if seen[root] or not neighbours[root]:
            continue
        seen[root] = True
        members, parents, children = [root], [], []
        queue = [root]
        while queue:
            nxt = []
            for p in queue:
                for q in neighbours[p]:
                    if not seen[q]:
                        seen[q] = True
                        members.append(q)
                        parents.append(p)
                        children.append(q)
                        nxt.append(q)
            # one generation at a time: all parents of this generation are already placed
            if nxt:
                par = np.asarray(parents[-len(nxt):])
                chi = np.asarray(children[-len(nxt):])
                positions[chi] = positions[par] + _minimum_image(positions[chi] - positions[par],
                                                                 dimensions)
            queue = nxt
        molecules.append(np.sort(np.asarray(members)))

    if masses is None:
        warnings.warn("No masses specified. All atoms are assumed to have a mass of 1.")
        masses = np.ones(n)
    else:
        masses = np.asarray(masses, dtype=float) if np.ndim(masses[0]) == 0 else np.concatenate(masses)
        if len(masses) != n:
            raise ValueError("The number of masses must be equal to the number of atoms or the "
                             "number of molecules.")
    for mol in molecules:
        com = (masses[mol, None] * positions[mol]).sum(axis=0) / masses[mol].sum()
        positions[mol] += wrap(com, dimensions[:3], in_place=False) - com
    return positions


def wrap(positions, dimensions, *, in_place: bool = True):
    """Wrap positions back into the primary cell ``[0, L]``."""
    if not in_place:
        positions = positions.copy()
    outside = (positions < 0) | (positions > dimensions)
    shift = np.floor(positions / dimensions) * dimensions
    positions[outside] -= shift[outside]
    return None if in_place else positions
