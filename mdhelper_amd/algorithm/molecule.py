"""
Frame-preparation helper ``center_of_mass`` (reference
``src/mdhelper/algorithm/molecule.py:15-310``): per-residue / per-segment centres
of mass used when ``groupings != "atoms"``.  O(N) per frame, host NumPy (SURVEY.md
§8 a-11: frame prep stays on the host in this round).
"""

from __future__ import annotations

import numpy as np


def _level_ids(group, grouping):
    if grouping == "residues":
        ids = getattr(group, "resindices", None)
    elif grouping == "segments":
        ids = getattr(group, "segindices", None)
    else:
        raise ValueError(f"Invalid grouping '{grouping}'.")
    if ids is None:
        raise ValueError(f"The group does not expose per-atom {grouping} indices.")
    return np.asarray(ids)


def center_of_mass(group=None, grouping: str = None, *, masses=None, positions=None,
                   images=None, dimensions=None, n_groups: int = None, raw: bool = False):
    r"""
    Centre(s) of mass :math:`\mathbf R=\sum_a m_a\mathbf r_a/\sum_a m_a`.

    Either pass a ``group`` (optionally with ``grouping`` = ``"residues"`` /
    ``"segments"`` for one centre per residue / segment, ``None`` for the whole
    group), or pass ``masses`` and ``positions`` directly (optionally reshaped into
    ``n_groups`` equal molecules).  ``images`` (boundary-crossing counts) unwrap
    the positions with ``dimensions`` first.
    """
    # (reference molecule.py:218-221: the grouping is checked before anything else)
    if grouping not in {None, "residues", "segments"}:
        raise ValueError(f"Invalid grouping: '{grouping}'. Valid options are None, 'residues', "
                         "and 'segments'.")
    if group is not None:
        pos = np.array(group.positions if positions is None else positions, dtype=float)
        m = np.asarray(group.masses if masses is None else masses, dtype=float)
        if images is not None:
            if dimensions is None:
                dims = getattr(group, "dimensions", None)
                if dims is None:
                    dims = getattr(group.universe, "dimensions", None)
                if dims is None:
                    raise ValueError("The number of periodic boundary crossings was provided, "
                                     "but no system dimensions were provided or found in the "
                                     "trajectory.")
                dimensions = dims
            pos = pos + np.asarray(images) * np.asarray(dimensions, dtype=float)[:3]
        if grouping is None and not n_groups:
            com = (m[:, None] * pos).sum(axis=0) / m.sum()
            return (com, m, pos) if raw else com
        if n_groups:
            m2 = m.reshape((n_groups, -1))
            p2 = pos.reshape((n_groups, -1, 3))
            com = np.einsum("...a,...ad->...d", m2, p2) / m2.sum(axis=-1, keepdims=True)
            return (com, m2, p2) if raw else com
        ids = _level_ids(group, grouping)
        _, inverse = np.unique(ids, return_inverse=True)
        n = inverse.max() + 1
        msum = np.bincount(inverse, weights=m, minlength=n)
        com = np.stack([np.bincount(inverse, weights=m * pos[:, k], minlength=n) for k in range(3)],
                       axis=1) / msum[:, None]
        return (com, m, pos) if raw else com

    if masses is None or positions is None:
        raise ValueError("Either a group or both masses and positions must be provided.")
    try:
        p = np.asarray(positions, dtype=float)
        m = np.asarray(masses, dtype=float)
        ragged = False
    except ValueError:
        ragged = True
    if ragged:
        return np.array([np.dot(np.asarray(mm, dtype=float), np.asarray(pp, dtype=float))
                         / np.sum(mm) for mm, pp in zip(masses, positions)])
    if n_groups:
        m = m.reshape((n_groups, -1))
        p = p.reshape((n_groups, -1, 3))
    if m.shape != p.shape[:-1]:
        raise ValueError("The shapes of the arrays containing the particle masses and "
                         "positions are incompatible.")
    return np.einsum("...a,...ad->...d", m, p) / m.sum(axis=-1, keepdims=True)
