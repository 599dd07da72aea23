"""
Unit plumbing (reference ``src/mdhelper/algorithm/unit.py:162-292``, ``strip_unit``).

``pint`` and OpenMM are not dependencies here.  Plain numbers pass through
(the second return value is then the *name* of the assumed unit, a ``str``, as
in the reference); quantities from either library are converted when they
know how (``.m_as`` / ``.to`` for pint, ``.value_in_unit`` for OpenMM), otherwise
their bare magnitude is taken.
"""

from __future__ import annotations

import numpy as np


def strip_unit(value, unit=None):
    """Return ``(magnitude, unit)``; ``unit`` stays a ``str`` for unit-less input."""
    if value is None:
        return None, unit
    if isinstance(value, (int, float, np.integer, np.floating, np.ndarray, list, tuple)):
        return value, unit
    # pint.Quantity
    if hasattr(value, "m_as") and hasattr(value, "units"):
        if isinstance(unit, str):
            try:
                return value.m_as(unit), value.units
            except Exception:
                return value.magnitude, value.units
        return value.magnitude, value.units
    # openmm.unit.Quantity
    if hasattr(value, "value_in_unit") and hasattr(value, "unit"):
        return value._value, value.unit
    return value, unit
