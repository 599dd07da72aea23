"""Analysis classes of the hot path (mirrors ``mdhelper.analysis``)."""

from . import base, polymer, structure, transport  # noqa: F401
from .structure import (IntermediateScatteringFunction, RadialDistributionFunction,  # noqa: F401
                        StructureFactor)
from .polymer import EndToEndVector  # noqa: F401
from .transport import Onsager  # noqa: F401
