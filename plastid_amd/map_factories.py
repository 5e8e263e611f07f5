"""The five BAM mapping rules + the size filter, behind the reference's plugin API.

Host-side mirror of ``plastid/genomics/map_factories.pyx`` (same class names,
constructor signatures, properties, ``__call__(reads, segment)`` contract and
error behaviour):

=========================================  ==========================
``CenterMapFactory(nibble=0)``             map_factories.pyx:167-276
``FivePrimeMapFactory(offset=0)``          :278-374
``ThreePrimeMapFactory(offset=0)``         :377-474
``VariableFivePrimeMapFactory(dict)``      :477-650  (+ ``from_file`` :545-582)
``StratifiedVariableFivePrimeMapFactory``  :653-791  (+ ``row_keys``, ``shape``)
``SizeFilterFactory(min=1, max=-1)``       :794-839
=========================================  ==========================

A *mapping function* is any callable ``f(alignments, segment) -> (reads_out,
ndarray)``, last axis = positions of `segment` in genome order
(docs/source/concepts/mapping_rules.rst:201-231).  The instances here are such
callables, but they hold only *parameters*: the arithmetic runs in the HIP
kernels of ``csrc/pc_kernels.hip.h``.  Called directly on a read list they pack
the reads, stage them to the GPU and count there; installed in a
:class:`~plastid_amd.genome_array.BAMGenomeArray` they configure its engine so
whole batches of intervals are counted per launch.  There is no CPU counting
path: without the HIP library the call raises
:class:`~plastid_amd.exceptions.EngineError`.
"""
import numpy as np

from . import _lib
from .exceptions import DataWarning, MalformedFileError, warn
from .packing import PackedAlignments, positions_to_runs
from .roitools import GenomicSegment

_BAD_OFFSET = -1
TABLE_LEN = _lib.OFFSET_TABLE_LEN  # map_factories.pxd:10-12


def _as_c_int(value, name):
    """Cython ``int`` argument conversion: ints (and ``__index__``) only."""
    if isinstance(value, (bool, np.bool_)):
        return int(value)
    if isinstance(value, (int, np.integer)):
        return int(value)
    raise TypeError("%s: an integer is required, got %s" % (name, type(value).__name__))


def _parse_variable_offset_file(fh):
    """Two-column, tab-delimited ``length<TAB>offset`` text (``default`` allowed as
    length; header line starting with ``length`` skipped) -> dict
    (plastid/util/scriptlib/argparsers.py:2507-2561)."""
    my_dict = {}
    for line in fh:
        if line.startswith("length"):
            continue
        items = line.strip("\n").split("\t")
        name = getattr(fh, "__name__", "Variable offset file")
        if len(items) != 2:
            raise MalformedFileError(name, "More or fewer than two columns on line:\n\t%s" % line.strip("\n"))
        if items[0] == "length":
            continue
        key = items[0]
        try:
            key = key if key == "default" else int(key)
        except ValueError:
            raise MalformedFileError(name, "Non integer value for key '%s' on line:\n\t%s" % (key, line.strip("\n")))
        if key in my_dict:
            raise MalformedFileError(name, "multiple offsets defined for read length %s" % key)
        try:
            my_dict[key] = int(items[1])
        except ValueError:
            raise MalformedFileError(
                name, "Non integer value for value '%s' on line:\n\t%s" % (items[1], line.strip("\n")))
    return my_dict


def _skip_comments(stream):
    """``CommentReader``: drop lines whose first non-blank char is '#' (util/io/filters.py:249-295)."""
    for line in stream:
        ltmp = line.lstrip()
        if len(ltmp) > 1 and ltmp[0] == "#":
            continue
        yield line


class _EngineMapFactory(object):
    """Shared machinery: configure an engine, and run a direct
    ``factory(reads, segment)`` call on the GPU."""

    _kind = None
    shape_rows = 1

    def _engine_args(self):
        """-> dict(kind, param, fw, rc, min_len, max_len) for ``Engine.set_mapping``."""
        raise NotImplementedError

    def _configure(self, engine):
        engine.set_mapping(**self._engine_args())

    def _out_dtype(self):
        return np.float64 if self._kind == _lib.MAP_CENTER else np.int64

    def _warning_text(self, packed_files, mask_unmappable=None):
        return None

    def __call__(self, reads, seg):
        if reads is None or not isinstance(reads, list):
            raise TypeError("Argument 'reads' has incorrect type (expected list, got %s)" % type(reads).__name__)
        if seg is None or not isinstance(seg, GenomicSegment):
            raise TypeError("Argument 'seg' has incorrect type (expected GenomicSegment, got %s)"
                            % type(seg).__name__)
        from .engine import Plan, default_engine
        engine = default_engine()
        files, index_lists = _pack_read_list(reads, order_matters=self._kind == _lib.MAP_CENTER)
        engine.set_alignments(files, ntid=1)
        self._configure(engine)
        engine.set_size_filter(None)
        engine.set_normalize(False)
        rows = engine.rows
        seg_len = len(seg)
        # direct call: no fetch, no strand filter -- the rule only looks at seg.strand (:345-346)
        strand = seg.c_strand | _lib.STRAND_NOFILTER
        plan = Plan(engine, [0], [seg.start], [seg.end], [strand], [0], [1], [seg_len], rows * seg_len, rows)
        counts = plan.count(self._out_dtype())
        if rows > 1 or self._kind == _lib.MAP_STRAT5:
            counts = counts.reshape(rows, seg_len)
        mapped = np.zeros(len(reads), bool)
        for fi, idx in enumerate(index_lists):
            # every read of the list is a candidate here (no fetch): widen the overlap window
            m = engine.mapped_reads(fi, 0, len(idx), 0, seg.start, seg.end, strand)
            if self._kind == _lib.MAP_CENTER:
                # CenterMapFactory returns every read with map_length > 0, overlapping or not (:249-256)
                m = (files[fi].alen.astype(np.int64) - 2 * self.nibble) > 0
            mapped[idx] = m.astype(bool)
        reads_out = [r for r, keep in zip(reads, mapped) if keep]
        msg = self._direct_warning(files)
        if msg:
            warn(msg, DataWarning)
        plan.close()
        return reads_out, counts

    def _direct_warning(self, files):
        return None


def _pack_read_list(reads, order_matters):
    """Pack an arbitrary read list (all assumed on the segment's chromosome, as in
    the reference, which never looks at the read's chromosome inside a map
    function).  The engine wants coordinate-sorted files: for order-insensitive
    rules the list is stably sorted; for the center rule (float64 sums in list
    order) the list is cut into its maximal sorted runs, staged as consecutive
    pseudo-files, which reproduces list order exactly (file-major replay)."""
    runs = [positions_to_runs(r.positions) for r in reads]
    rev = [bool(r.is_reverse) for r in reads]
    n = len(reads)
    if n == 0:
        return [PackedAlignments.from_runs(0, [], [], references=["_"], lengths=[0])], [np.zeros(0, np.int64)]
    starts = np.array([r[0][0] if r else 0 for r in runs], np.int64)
    if not order_matters:
        order = np.argsort(starts, kind="stable")
        f = PackedAlignments.from_runs(0, [rev[i] for i in order], [runs[i] for i in order],
                                       references=["_"], lengths=[0])
        return [f], [order]
    cuts = [0] + (np.nonzero(np.diff(starts) < 0)[0] + 1).tolist() + [n]
    if len(cuts) - 1 > 256:
        raise ValueError("CenterMapFactory: read list is too disordered (%d unsorted runs); "
                         "pass reads in coordinate order" % (len(cuts) - 1))
    files, index_lists = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        files.append(PackedAlignments.from_runs(0, rev[a:b], runs[a:b], references=["_"], lengths=[0]))
        index_lists.append(np.arange(a, b))
    return files, index_lists


class CenterMapFactory(_EngineMapFactory):
    """CenterMapFactory(nibble=0)

    `nibble` positions are removed from each side of each read alignment, and the
    `N` remaining positions are each apportioned `1/N` of the read count
    (map_factories.pyx:167-276).  float64, summed in read order."""
    _kind = _lib.MAP_CENTER

    def __init__(self, nibble=0):
        self.nibble = nibble

    @property
    def nibble(self):
        """Number of positions to trim from each side of read alignment"""
        return self._nibble

    @nibble.setter
    def nibble(self, val):
        val = _as_c_int(val, "nibble")
        if val < 0:  # `unsigned int nibble`: conversion fails before the ValueError branch (:182)
            raise OverflowError("can't convert negative value to unsigned int")
        self._nibble = val

    def _engine_args(self):
        return dict(kind=self._kind, param=self._nibble)

    def _direct_warning(self, files):
        if any(np.any(f.alen.astype(np.int64) - 2 * self._nibble < 0) for f in files):
            return ("Data contains read alignments shorter than `2*nibble` value of '%s' nt. Ignoring these."
                    % (2 * self._nibble))

    def _warn_message(self):
        return ("Data contains read alignments shorter than `2*nibble` value of '%s' nt. Ignoring these."
                % (2 * self._nibble))


class _OffsetMapFactory(_EngineMapFactory):
    _name = ""

    def __init__(self, offset=0):
        offset = _as_c_int(offset, "offset")
        if offset < 0:  # :304-305, :403-404 (message text as in the reference)
            raise ValueError("%s: `offset` must be <= 0. Got %s." % (self._name, offset))
        self._offset = offset

    @property
    def offset(self):
        return self._offset

    @offset.setter
    def offset(self, val):
        val = _as_c_int(val, "offset")
        if val < 0:
            raise ValueError("%s: `offset` must be >= 0 for the HIP engine. Got %s." % (self._name, val))
        self._offset = val

    def _engine_args(self):
        return dict(kind=self._kind, param=self._offset)

    def _warn_message(self):
        return "Data contains read alignments shorter than offset (%s nt). Ignoring." % self._offset

    def _direct_warning(self, files):
        if any(np.any(f.alen.astype(np.int64) <= self._offset) for f in files):
            return self._warn_message()


class FivePrimeMapFactory(_OffsetMapFactory):
    """FivePrimeMapFactory(offset=0)

    Reads are mapped at `offset` nucleotides from the fiveprime end of their
    alignment (map_factories.pyx:278-374)."""
    _kind = _lib.MAP_FIVE
    _name = "FivePrimeMapFactory"


class ThreePrimeMapFactory(_OffsetMapFactory):
    """ThreePrimeMapFactory(offset=0)

    Reads are mapped `offset` nucleotides from the threeprime end of their
    alignments (map_factories.pyx:377-474)."""
    _kind = _lib.MAP_THREE
    _name = "ThreePrimeMapFactory"


def build_offset_tables(offset_dict):
    """``VariableFivePrimeMapFactory.__cinit__`` (map_factories.pyx:494-543):
    ``forward_offsets`` / ``reverse_offsets`` int32[10000], -1 = no usable offset."""
    fw = np.full(TABLE_LEN, _BAD_OFFSET, np.int32)   # :511-512
    rc = np.full(TABLE_LEN, _BAD_OFFSET, np.int32)
    if offset_dict is None:                           # :517-518
        offset_dict = {"default": 0}
    elif not isinstance(offset_dict, dict):
        raise TypeError("Argument 'offset_dict' has incorrect type (expected dict, got %s)"
                        % type(offset_dict).__name__)
    have_default = "default" in offset_dict
    default = None
    if have_default:                                  # :520-526
        default = int(offset_dict["default"])
        if default < 0:
            raise ValueError("VariableFivePrimeMapFactory: default offset must be >= 0, got %s" % default)
        if default + 1 < TABLE_LEN:
            fw[default + 1:] = default
            rc[default + 1:] = np.arange(default + 1, TABLE_LEN) - default - 1
    for read_length, offset in offset_dict.items():   # :530-543
        if read_length == "default":
            continue
        read_length = _as_c_int(read_length, "read length")
        offset = _as_c_int(offset, "offset")
        if offset >= read_length:
            if not have_default:
                # the reference reads the unbound local `default` at :533
                raise UnboundLocalError("local variable 'default' referenced before assignment")
            if read_length >= default:
                warn("Given offset '%s' longer than read length '%s'. Falling back to default '%s'."
                     % (offset, read_length, default), DataWarning)
            else:
                warn("Given offset '%s' and default '%s' are longer than read length '%s'. Ignoring %s-mers."
                     % (offset, default, read_length, read_length), DataWarning)
            continue                                   # :540 -- the entry is dropped in both branches
        if offset < 0 or read_length >= TABLE_LEN:
            # the reference would index read.positions / its tables out of bounds
            raise ValueError("VariableFivePrimeMapFactory: offset %s for read length %s is out of range"
                             % (offset, read_length))
        fw[read_length] = offset
        rc[read_length] = read_length - offset - 1
    return fw, rc


class VariableFivePrimeMapFactory(_EngineMapFactory):
    """VariableFivePrimeMapFactory(offset_dict)

    Reads are mapped at ``offset_dict[read length]`` (or ``offset_dict['default']``)
    nucleotides from their fiveprime ends (map_factories.pyx:477-650)."""
    _kind = _lib.MAP_VAR5

    def __init__(self, offset_dict, *args):
        self.forward_offsets, self.reverse_offsets = build_offset_tables(offset_dict)
        self.offset_dict = None if offset_dict is None else dict(offset_dict)

    @classmethod
    def from_file(cls, fn_or_fh):
        """Create the factory from a two-column text file as written by the `psite`
        script (map_factories.pyx:545-582)."""
        if isinstance(fn_or_fh, str):
            with open(fn_or_fh) as my_fh:
                return cls(_parse_variable_offset_file(my_fh))
        return cls(_parse_variable_offset_file(_skip_comments(fn_or_fh)))

    def _engine_args(self):
        return dict(kind=self._kind, param=0, fw=self.forward_offsets, rc=self.reverse_offsets)

    def _unmappable_lengths(self, lengths):
        lengths = np.asarray(lengths, np.int64)
        bad = lengths >= TABLE_LEN
        ok = ~bad
        bad[ok] = self.forward_offsets[lengths[ok]] == _BAD_OFFSET
        return bad

    def _warn_message(self, length="(some)"):
        return "No usable offset for reads of length %s nt in offset dict. Ignoring these." % length

    def _direct_warning(self, files):
        for f in reversed(files):
            bad = self._unmappable_lengths(f.alen)
            if bad.any():
                return self._warn_message(int(f.alen[np.nonzero(bad)[0][-1]]))


class StratifiedVariableFivePrimeMapFactory(VariableFivePrimeMapFactory):
    """StratifiedVariableFivePrimeMapFactory(offset_dict, min=25, max=35)

    As :class:`VariableFivePrimeMapFactory`, but counts go into a 2D array: one
    row per read length ``min..max`` (inclusive), one column per position
    (map_factories.pyx:653-791).  Lengths without a usable offset map to the
    read's last aligned position (no bad-offset check at :773-774); never warns."""
    _kind = _lib.MAP_STRAT5

    def __init__(self, offset_dict, min=25, max=35):
        VariableFivePrimeMapFactory.__init__(self, offset_dict)
        min = _as_c_int(min, "min")
        max = _as_c_int(max, "max")
        if max <= min:  # :716-717 (so max == min is rejected, despite the message)
            raise ValueError("Max length '%s' must be >= min length '%s'. " % (max, min))
        if max >= TABLE_LEN:
            raise ValueError("Max length '%s' beyond the offset table (%s)" % (max, TABLE_LEN))
        self.min_length = min
        self.max_length = max
        self._numlengths = max - min + 1

    @property
    def row_keys(self):
        """numpy array of read lengths corresponding to each row of mapped data."""
        return np.arange(self.min_length, self.max_length + 1)

    @property
    def shape(self):
        """size of each axis of output, excluding the final (position) axis."""
        return [self._numlengths]

    def _engine_args(self):
        return dict(kind=self._kind, param=0, fw=self.forward_offsets, rc=self.reverse_offsets,
                    min_len=self.min_length, max_len=self.max_length)

    def _direct_warning(self, files):
        return None


class SizeFilterFactory(object):
    """SizeFilterFactory(min=1, max=-1)

    Read-length filter for :meth:`BAMGenomeArray.add_filter`
    (map_factories.pyx:794-839).  ``max == -1``: no maximum.  Recognised by
    :class:`~plastid_amd.genome_array.BAMGenomeArray`, which applies it inside the
    HIP kernels; as a plain callable it answers for a single read."""

    def __init__(self, min=1, max=-1):
        min = _as_c_int(min, "min")
        max = _as_c_int(max, "max")
        if max != -1 and max < min:
            raise ValueError("Alignment size filter: max read length must be >= min read length")
        if min < 1:
            raise ValueError("Alignment size filter: min read length must be >= 1. Got %s" % min)
        self.min_ = min
        self.max_ = max

    def __call__(self, read):
        if read is None:
            raise TypeError("Argument 'read' must not be None")
        my_length = len(read.positions)
        return my_length >= self.min_ and (my_length <= self.max_ or self.max_ == -1)


#: SAM FLAG bits by pysam's property names (kent/src/htslib/htslib/sam.h:110-132)
FLAG_BITS = {"is_paired": 0x1, "is_proper_pair": 0x2, "is_unmapped": 0x4, "mate_is_unmapped": 0x8, "is_reverse": 0x10,
             "mate_is_reverse": 0x20, "is_read1": 0x40, "is_read2": 0x80, "is_secondary": 0x100, "is_qcfail": 0x200,
             "is_duplicate": 0x400, "is_supplementary": 0x800}


class FlagFilterFactory(object):
    """FlagFilterFactory(require=0, exclude=0, min_mapq=0, max_nh=0)

    Read filter on the SAM FLAG word, MAPQ and the ``NH:i`` tag for :meth:`BAMGenomeArray.add_filter`: a read is kept iff
    ``(read.flag & require) == require and (read.flag & exclude) == 0 and read.mapping_quality >= min_mapq`` and --
    with `max_nh` > 0 -- ``read.has_tag("NH") and read.get_tag("NH") <= max_nh`` (``max_nh=1``: the unique mappers).
    `require` / `exclude` are bit masks or iterables of pysam property names (``"is_secondary"``,
    ``"is_duplicate"``, ``"is_qcfail"``, ``"is_proper_pair"`` ...).

    The reference has no such class: there one writes ``lambda read: not read.is_secondary and
    read.mapping_quality >= 10`` and it is called on every fetched read (genome_array.py:697-722, 819-820).
    That callable works here too (on the host, read by read); this class says the same thing in a form
    :class:`~plastid_amd.genome_array.BAMGenomeArray` evaluates on the GPU (``pc_set_flag_filter``: one
    pass over 3 bytes per record), also for files opened with ``keep_reads=False``.  As a plain callable
    it answers for a single read."""

    def __init__(self, require=0, exclude=0, min_mapq=0, max_nh=0):
        self.require = self._mask(require)
        self.exclude = self._mask(exclude)
        self.min_mapq = _as_c_int(min_mapq, "min_mapq")
        if not 0 <= self.min_mapq <= 255:
            raise ValueError("FlagFilterFactory: min_mapq must be in 0 .. 255. Got %s" % min_mapq)
        self.max_nh = _as_c_int(max_nh, "max_nh")
        if not 0 <= self.max_nh <= 65535:
            raise ValueError("FlagFilterFactory: max_nh must be in 0 (no test) .. 65535. Got %s" % max_nh)
        if self.require & self.exclude:
            raise ValueError("FlagFilterFactory: the same FLAG bit is both required and excluded")

    @staticmethod
    def _mask(spec):
        if isinstance(spec, str):
            spec = [spec]
        if isinstance(spec, (list, tuple, set, frozenset)):
            m = 0
            for name in spec:
                if name not in FLAG_BITS:
                    raise ValueError("FlagFilterFactory: unknown FLAG property %r" % (name,))
                m |= FLAG_BITS[name]
            return m
        m = _as_c_int(spec, "flag mask")
        if not 0 <= m <= 0xffff:
            raise ValueError("FlagFilterFactory: FLAG masks are 16-bit. Got %s" % spec)
        return m

    def __call__(self, read):
        if read is None:
            raise TypeError("Argument 'read' must not be None")
        flag = int(read.flag)
        if not ((flag & self.require) == self.require and (flag & self.exclude) == 0 and int(read.mapping_quality) >= self.min_mapq):
            return False
        return not self.max_nh or (bool(read.has_tag("NH")) and int(read.get_tag("NH")) <= self.max_nh)
