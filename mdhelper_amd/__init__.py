"""
mdhelper_amd — MI355X-native drop-in for the per-frame hot path of
bbye98/mdhelper: ``analysis.structure`` (radial distribution function, static /
partial structure factor) and ``analysis.transport`` (MSD / Onsager
coefficients) with ``algorithm.correlation`` underneath.

Python host code over hand-written gfx950 HIP kernels (``libmdx.so``, C-ABI in
``include/mdx.h``), reached through ctypes.  No CPU fallback: the analysis
classes raise when the library or a GPU is missing.
"""

__version__ = "0.1.0"

from . import algorithm, analysis  # noqa: E402,F401
from .universe import ArrayUniverse  # noqa: E402,F401
from .io import FileUniverse  # noqa: E402,F401
