"""Analysis classes of the hot path (mirrors ``mdhelper.analysis``)."""

from . import base, structure, transport  # noqa: F401
from .structure import (IntermediateScatteringFunction, RadialDistributionFunction,  # noqa: F401
                        StructureFactor)
from .transport import Onsager  # noqa: F401
