"""
Frame-preparation helpers ``unwrap`` and ``wrap`` (reference
``src/mdhelper/algorithm/topology.py:294-383`` and ``:531-578``), used by
``Onsager`` before the positions are stored (reference analysis/transport.py:979-981,
997-1004).  O(N) per frame, host NumPy.
"""

from __future__ import annotations

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


def wrap(positions, dimensions, *, in_place: bool = True):
    """Wrap positions back into the primary cell ``[0, L]``."""
    if not in_place:
        positions = positions.copy()
    outside = (positions < 0) | (positions > dimensions)
    shift = np.floor(positions / dimensions) * dimensions
    positions[outside] -= shift[outside]
    return None if in_place else positions
