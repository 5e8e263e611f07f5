"""Warning/exception types used on the counting path.

Mirrors the names the reference's hot path raises
(plastid/util/services/exceptions.py:124 ``DataWarning``), so user code that
filters on them keeps working.  The reference's "once per family" warning
machinery (exceptions.py:146-235) is out of scope: the path only has to *raise*
the warning once per call, which plain :func:`warnings.warn` does.
"""
import warnings


class DataWarning(UserWarning):
    """Raised when data has attributes that are unexpected, but that may not
    be cause for alarm (e.g. read alignments too short for a mapping rule).

    The reference declares ``DataWarning(Warning)`` but its mapping functions call
    ``warn_onceperfamily(msg, DataWarning)`` with the class in the *pattern*
    slot (map_factories.pyx:259-263; signature exceptions.py:203), so what users
    actually receive there is a ``UserWarning`` [observed on the scratch build].
    Deriving from ``UserWarning`` satisfies filters written against either."""


class MalformedFileError(Exception):
    """A file cannot be parsed as expected (plastid/util/services/exceptions.py)."""

    def __init__(self, filename, message, line_num=None):
        self.filename = filename
        self.msg = message
        self.line_num = line_num
        Exception.__init__(self, filename, message, line_num)

    def __str__(self):
        if self.line_num is None:
            return "Error opening file '%s': %s" % (self.filename, self.msg)
        return "Error opening file '%s' at line %s: %s" % (self.filename, self.line_num, self.msg)


class FileFormatWarning(Warning):
    """A file is not formatted as expected but can still be read
    (plastid/util/services/exceptions.py:119), e.g. a repeated attribute key in a GTF2 line."""


class ArgumentWarning(Warning):
    """Raised when arguments are nonsensical but recoverable."""


class EngineError(RuntimeError):
    """The HIP counting engine reported an error (or is not available)."""


def warn(message, category=DataWarning, stacklevel=3):
    warnings.warn(message, category, stacklevel=stacklevel)
