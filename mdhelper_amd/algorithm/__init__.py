"""Algorithms of the hot path (mirrors ``mdhelper.algorithm``)."""

from . import accelerated, correlation, molecule, topology, unit, utility  # noqa: F401
