"""plastid_amd -- MI355X-native per-position read counting behind plastid's
``BAMGenomeArray`` / MapFactory / ``SegmentChain.get_counts`` interfaces.

Only the hot path is here: packed alignment records are
staged to HBM and counted by hand-written HIP kernels (``csrc/``) through the C
ABI in ``include/plastid_counts.h``.  There is no CPU counting fallback.
"""
from .exceptions import DataWarning, EngineError, MalformedFileError  # noqa: F401
from .packing import PackedAlignments  # noqa: F401
from .roitools import GenomicSegment, SegmentChain  # noqa: F401
from .map_factories import (CenterMapFactory, FivePrimeMapFactory, FlagFilterFactory, SizeFilterFactory,  # noqa: F401
                            StratifiedVariableFivePrimeMapFactory, ThreePrimeMapFactory,
                            VariableFivePrimeMapFactory)
from .genome_array import BAMGenomeArray, DenseGenomeArray  # noqa: F401

__version__ = "0.1.0"
