"""Interval model of the counting path: |GenomicSegment| and the hot subset of
|SegmentChain|.

Host-side mirror of the reference interface (same names, argument meaning and
error behaviour) for the part of ``plastid/genomics/roitools.pyx`` that is on the
counting path:

* ``GenomicSegment``                      roitools.pyx:941-1249
* ``merge_segments`` / ``positions_to_segments``  :257-334, :336-390
* ``SegmentChain.__cinit__/_set_segments``  :1333-1448 (sort + merge of overlapping
  *and adjacent* segments, :296)
* ``_get_position_hash``, ``get_position_list/set``, ``get_masked_position_set``
  :1450-1484, :2059-2135
* ``add_masks/_set_masks/reset_masks``     :2213-2366
* ``get_counts`` / ``get_masked_counts``   :3221-3315

``get_counts`` keeps the reference's duck-typed contract -- it calls
``ga.get(segment, roi_order=False)`` per segment on *any* array object -- but
when `ga` is this package's :class:`~plastid_amd.genome_array.BAMGenomeArray`
the whole chain is counted in one batched HIP launch instead.

Out of scope (not on the path): BED/GTF/PSL export, comparison operators beyond
equality/ordering of segments, ``Transcript``.
"""
import re
import warnings

import numpy as np

from .exceptions import DataWarning

_STRANDS = ("+", "-", ".", "\x00")
STRAND_CODE = {"\x00": 0, "+": 1, "-": 2, ".": 3}  # plastid/genomics/c_common.pxd:1-6
segpat = re.compile(r"([^:]*):([0-9]+)-([0-9]+)\(([+-.])\)")
ivcpat = re.compile(r"([^:]*):([^(]+)\(([+-.])\)")


class GenomicSegment(object):
    """A continuous span of a chromosome on a strand: ``chrom:start-end(strand)``,
    0-indexed, half-open (roitools.pyx:941-1249)."""

    __slots__ = ("_chrom", "_start", "_end", "_strand")

    def __init__(self, chrom, start, end, strand):
        if not isinstance(chrom, str):
            raise TypeError("GenomicSegment: chrom must be str")
        start = int(start)
        end = int(end)
        if end < start:  # roitools.pyx:1022-1023
            raise ValueError("GenomicSegment: start coordinate (%s) must be >= end (%s)." % (start, end))
        if strand not in _STRANDS:  # c_common.pyx str_to_strand
            raise ValueError("Strand must be '+', '-', '.', or '\\x00' (undefined). Got '%s'." % strand)
        self._chrom = chrom
        self._start = start
        self._end = end
        self._strand = strand

    chrom = property(lambda self: self._chrom)
    start = property(lambda self: self._start)
    end = property(lambda self: self._end)

    @property
    def strand(self):
        return "strand undefined" if self._strand == "\x00" else self._strand

    @property
    def c_strand(self):
        return STRAND_CODE[self._strand]

    def __reduce__(self):
        return (GenomicSegment, (self._chrom, self._start, self._end, self._strand))

    def __repr__(self):
        return "<%s %s:%s-%s strand='%s'>" % ("GenomicSegment", self.chrom, self.start, self.end, self.strand)

    def __str__(self):
        return "%s:%s-%s(%s)" % (self.chrom, self.start, self.end, self.strand)

    def __hash__(self):
        return hash(("GenomicSegment", self.chrom, self.start, self.end, self.strand))

    def __len__(self):
        return self._end - self._start

    def _key(self):
        return (self._chrom, self._start, self._end, STRAND_CODE[self._strand])

    def __eq__(self, other):
        if other is None or not isinstance(other, GenomicSegment):
            return False
        return self._key() == other._key()

    def __ne__(self, other):
        return not self.__eq__(other)

    def __lt__(self, other):
        return self._key() < other._key()

    def __gt__(self, other):
        return other._key() < self._key()

    def __le__(self, other):
        return self._key() <= other._key()

    def __ge__(self, other):
        return other._key() <= self._key()

    def __contains__(self, other):
        return self.contains(other)

    def contains(self, other):
        return (self.chrom == other.chrom and self.c_strand == other.c_strand
                and other.start >= self.start and other.end <= self.end and other.end >= other.start)

    def overlaps(self, other):
        if self.chrom == other.chrom and self.c_strand == other.c_strand:
            if (self.start >= other.start and self.start < other.end) \
                    or (other.start >= self.start and other.start < self.end):
                return True
        return False

    @staticmethod
    def from_str(inp):
        chrom, s_start, s_end, strand = segpat.search(inp).groups()
        return GenomicSegment(chrom, int(s_start), int(s_end), strand)

    def as_igv_str(self):
        return "%s:%s-%s" % (self.chrom, self.start + 1, self.end + 1)


NullSegment = GenomicSegment("NullChromosome", 0, 0, "\x00")


def merge_segments(segments):
    """Merge overlapping or adjacent |GenomicSegments| (assumed same chrom and
    strand); result sorted and non-overlapping (roitools.pyx:257-306)."""
    if len(segments) < 2:
        return segments
    ssegments = sorted(segments)
    left = ssegments[0]
    chrom, strand = left._chrom, left._strand
    new_segments = []
    for right in ssegments[1:]:
        if right.start > left.end:  # :296 -- adjacent segments (== end) are merged too
            new_segments.append(left)
            left = right
        else:
            left = GenomicSegment(chrom, left.start, max(left.end, right.end), strand)
    new_segments.append(left)
    return new_segments


def positionlist_to_segments(chrom, strand, positions):
    """|GenomicSegments| covering a SORTED, UNIQUE list of positions (roitools.pyx:336-390)."""
    pos = np.asarray(positions, dtype=np.int64)
    if len(pos) == 0:
        return []
    breaks = np.nonzero(np.diff(pos) != 1)[0]
    starts = np.concatenate([[pos[0]], pos[breaks + 1]])
    ends = np.concatenate([pos[breaks], [pos[-1]]]) + 1
    return [GenomicSegment(chrom, int(s), int(e), strand) for s, e in zip(starts, ends)]


def positions_to_segments(chrom, strand, positions):
    """|GenomicSegments| covering any iterable of positions (roitools.pyx:308-334)."""
    return positionlist_to_segments(chrom, strand, sorted(set(positions)))


class SegmentChain(object):
    """A feature made of zero or more |GenomicSegments| on one chromosome and
    strand, kept sorted left-to-right and merged (roitools.pyx:1251-3553, hot subset)."""

    def __init__(self, *segments, **attr):
        if "type" not in attr:
            attr["type"] = "exon"
        self.attr = attr
        self._mask_segments = None
        self._segments = []
        self.length = 0
        self.masked_length = 0
        self.spanning_segment = NullSegment
        self._position_mask = None
        self._position_hash = None
        if len(segments) == 1:
            self._set_segments(list(segments))
        elif len(segments) > 1:
            self._check_segments(segments)
            self._set_segments(merge_segments(list(segments)))

    # ------------------------------------------------------------- structure
    def _check_segments(self, segments):
        """roitools.pyx:749-784"""
        if len(segments) == 0:
            return
        span = self.spanning_segment
        my_chrom, my_strand = span.chrom, span.c_strand
        if len(self._segments) == 0:
            my_chrom, my_strand = segments[0].chrom, segments[0].c_strand
        msg = "SegmentChain.add_segments: incoming segments (%s) mismatch chain '%s'"
        for seg in segments:
            if seg.chrom != my_chrom:
                raise ValueError((msg + "; wrong and/or multiple chromosomes")
                                 % (", ".join(str(X) for X in segments), self))
            if seg.c_strand != my_strand:
                raise ValueError((msg + "; wrong and/or multiple strands")
                                 % (", ".join(str(X) for X in segments), self))

    def _set_segments(self, segments):
        """roitools.pyx:1388-1448 (no merging/sorting here; callers guarantee it)"""
        self._segments = segments
        self.length = sum(len(x) for x in segments)
        self._position_hash = None
        self.reset_masks()
        if len(segments) == 0:
            self.spanning_segment = NullSegment
        elif len(segments) == 1:
            self.spanning_segment = segments[0]
        else:
            seg0 = segments[0]
            self.spanning_segment = GenomicSegment(seg0._chrom, seg0.start, segments[-1].end, seg0._strand)
        return True

    def add_segments(self, *segments):
        if len(segments) > 0:
            if self._mask_segments is not None and len(self._mask_segments) > 0:
                warnings.warn("Segmentchain: adding segments to %s will reset its masks!" % self, UserWarning)
            self._check_segments(segments)
            self._set_segments(merge_segments(list(segments) + self._segments))

    chrom = property(lambda self: self.spanning_segment.chrom)
    strand = property(lambda self: self.spanning_segment.strand)
    c_strand = property(lambda self: self.spanning_segment.c_strand)

    @property
    def segments(self):
        return list(self._segments)

    @property
    def mask_segments(self):
        return [] if self._mask_segments is None else list(self._mask_segments)

    def __len__(self):
        return len(self._segments)

    def __iter__(self):
        return iter(self._segments)

    def __getitem__(self, index):
        return self._segments[index]

    def get_length(self):
        return self.length

    def get_masked_length(self):
        return self.masked_length

    def get_name(self):
        return self.attr.get("ID", self.attr.get("Name", self.attr.get("name", str(self))))

    def __str__(self):
        if len(self) > 0:
            return "%s:%s(%s)" % (self.chrom, "^".join("%s-%s" % (s.start, s.end) for s in self), self.strand)
        return "na"

    def __repr__(self):
        sout = "<%s segments=%s" % (self.__class__.__name__, len(self._segments))
        if len(self) > 0:
            span = self.spanning_segment
            sout += " bounds=%s:%s-%s(%s)" % (span.chrom, span.start, span.end, span.strand)
            sout += " name=%s" % self.get_name()
        return sout + ">"

    def __eq__(self, other):
        return isinstance(other, SegmentChain) and self._segments == other._segments

    def __hash__(self):
        return hash(str(self))

    @staticmethod
    def from_str(inp):
        """Inverse of ``str(chain)``: ``chrom:start-end^start-end(strand)`` (roitools.pyx:3379-3417)."""
        if inp in ("na", "nan", "None:(None)", "None", "none", None) or \
                (isinstance(inp, float) and np.isnan(inp)):
            return SegmentChain()
        chrom, middle, strand = ivcpat.search(inp).groups()
        segs = []
        for piece in middle.split("^"):
            sstart, send = piece.split("-")
            segs.append(GenomicSegment(chrom, int(sstart), int(send), strand))
        return SegmentChain(*segs)

    @staticmethod
    def from_bed(line, extra_columns=0):
        """|SegmentChain| from a BED line (4-12 columns).  As in the reference
        (roitools.pyx:3465-3470) the blocks are taken as given: no merging."""
        from .annotation import bed_line_to_chain
        return bed_line_to_chain(line, SegmentChain)

    # ------------------------------------------------------------- positions
    def _get_position_hash(self):
        """Genomic coordinate of every chain position, leftmost first regardless
        of strand (roitools.pyx:1450-1484)."""
        if self._position_hash is None:
            if self._segments:
                self._position_hash = np.concatenate(
                    [np.arange(s.start, s.end, dtype=np.int64) for s in self._segments])
            else:
                self._position_hash = np.zeros(0, np.int64)
        return self._position_hash

    def get_position_list(self):
        return self._get_position_hash().tolist()

    def get_position_set(self):
        return set(self._get_position_hash().tolist())

    def get_masked_position_set(self):
        """Genomic coordinates of `self` NOT masked (roitools.pyx:2117-2135)."""
        if self._position_mask is None:
            return self.get_position_set()
        return set(self._get_position_hash()[self._position_mask == 0].tolist())

    def get_segmentchain_coordinate(self, genomic_x, stranded=True):
        """roitools.pyx:2957-3012"""
        msg = "SegmentChain.get_segmentchain_coordinate: genomic position '%s' is not in chain '%s'.\n" % (
            genomic_x, self)
        if genomic_x < self.spanning_segment.start:
            raise KeyError(msg)
        cumlength = 0
        for seg in self._segments:
            cumlength += len(seg)
            if genomic_x < seg.end:
                if genomic_x >= seg.start:
                    retval = cumlength - seg.end + genomic_x
                    if self.c_strand == 2 and stranded is True:
                        retval = self.length - retval - 1
                    return retval
                raise KeyError(msg)
        raise KeyError(msg)

    # ----------------------------------------------------------------- masks
    def add_masks(self, *mask_segments):
        """Mask positions (union with existing masks, trimmed to the chain) --
        roitools.pyx:2213-2256, done on sorted interval arrays instead of Python sets."""
        if len(mask_segments) > 0:
            self._check_segments(mask_segments)
            seg = mask_segments[0]
            my_chrom, my_strand = seg._chrom, seg._strand
            segs = list(mask_segments)
            if self._mask_segments is not None:
                segs += self._mask_segments
            pos = self._get_position_hash()
            flag = np.zeros(len(pos), bool)
            for m in segs:
                lo, hi = np.searchsorted(pos, [m.start, m.end], side="left")
                flag[lo:hi] = True
            new_segments = positionlist_to_segments(my_chrom, my_strand, pos[flag])
            self._set_masks(new_segments)

    def _set_masks(self, segments):
        """roitools.pyx:2258-2301"""
        pos = self._get_position_hash()
        pmask = np.zeros(self.length, np.intc)
        tmpsum = 0
        for seg in segments:
            lo, hi = np.searchsorted(pos, [seg.start, seg.end], side="left")
            if hi - lo != len(seg):
                raise KeyError("SegmentChain._set_masks: mask %s is not within chain %s" % (seg, self))
            pmask[lo:hi] = 1
            tmpsum += len(seg)
        self._mask_segments = segments
        self.masked_length = self.length - tmpsum
        self._position_mask = pmask
        return True

    def reset_masks(self):
        """roitools.pyx:2349-2366"""
        self._position_mask = None
        self._mask_segments = None
        self.masked_length = self.length

    def get_masks_as_segmentchain(self):
        if self._mask_segments is None:
            return SegmentChain()
        return SegmentChain(*self._mask_segments)

    # ---------------------------------------------------------------- counts
    def get_counts(self, ga, stranded=True):
        """Counts from `ga` at each position of `self`, 5'->3' (roitools.pyx:3221-3273)."""
        if len(self) == 0:  # :3248-3253
            warnings.warn("%s is a zero-length SegmentChain. Returning 0-length count vector."
                          % self.get_name(), DataWarning)
            return np.array([], dtype=float)
        batched = getattr(ga, "_get_chain_counts", None)
        if batched is not None:
            return batched(self, stranded)
        # generic duck-typed array: the reference's own per-segment loop (:3259-3271)
        count_arrays = [ga.get(X, roi_order=False) for X in self._segments]
        dims = list(count_arrays[0].shape)
        dims[-1] = self.length
        count_array = np.empty(dims, dtype=float)
        i = 0
        for n, seg in enumerate(self._segments):
            j = i + len(seg)
            count_array[..., i:j] = count_arrays[n]
            i = j
        if self.c_strand == 2 and stranded is True:
            count_array = count_array[..., ::-1]
        return count_array

    def get_masked_counts(self, ga, stranded=True, copy=False):
        """Masked array of counts (roitools.pyx:3275-3315; as there, `stranded` is
        not forwarded to ``get_counts`` and the mask is always flipped for '-')."""
        counts = self.get_counts(ga)  # :3301
        if self._position_mask is None:
            mask = np.zeros_like(counts)
        else:
            m = self._position_mask
            if self.c_strand == 2:  # :3309-3310
                m = m[::-1]
            mask = np.empty_like(counts)
            mask[..., :] = m
        return np.ma.MaskedArray(counts, mask=mask.astype(bool), copy=copy)
