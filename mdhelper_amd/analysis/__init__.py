"""Analysis classes of the hot path (mirrors ``mdhelper.analysis``)."""

from . import base, structure, transport  # noqa: F401
from .structure import RadialDistributionFunction, StructureFactor  # noqa: F401
from .transport import Onsager  # noqa: F401
