"""Warning/exception types used on the counting path.

Mirrors the names the reference's hot path raises
(plastid/util/services/exceptions.py:124 ``DataWarning``), so user code that
filters on them keeps working.  The reference's "once per family" warning
machinery (exceptions.py:146-235) is out of scope: the path only has to *raise*
the warning once per call, which plain :func:`warnings.warn` does.
"""
import warnings


class DataWarning(Warning):
    """Raised when data has attributes that are unexpected, but that may not
    be cause for alarm (e.g. read alignments too short for a mapping rule)."""


class ArgumentWarning(Warning):
    """Raised when arguments are nonsensical but recoverable."""


class EngineError(RuntimeError):
    """The HIP counting engine reported an error (or is not available)."""


def warn(message, category=DataWarning, stacklevel=3):
    warnings.warn(message, category, stacklevel=stacklevel)
