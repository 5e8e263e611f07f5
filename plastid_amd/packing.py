"""Packed alignment records: the host-side staging format of the counting engine.

The reference consumes, per read, ``read.positions`` (pysam
``get_reference_positions()``: ascending reference coordinates of the CIGAR
M/=/X bases) and ``read.is_reverse`` (plastid/genomics/map_factories.pyx:243,
349, 448, 629, 769, 838; plastid/genomics/genome_array.py:812-815).  A
:class:`PackedAlignments` holds exactly that information for a whole
coordinate-sorted BAM file as flat numpy arrays (structure of arrays), which is
what gets bulk-staged to HBM:

==============  ========  ====================================================
``tid``         int32     reference (contig) index
``pos``         int32     leftmost aligned reference coordinate (0-based)
``alen``        uint16    ``L = len(read.positions)`` ("read length" in plastid)
``flags``       uint8     bit0 = ``read.is_reverse``; bit7 = excluded by a
                          host-side (arbitrary Python) read filter
``nblk``        uint8     number of maximal runs of contiguous aligned
                          reference positions (0 iff ``L == 0``)
``blk_start``   int32     runs of every record with ``nblk >= 2``, record after
``blk_len``     int32     record (a ``nblk == 1`` record has the single implicit
                          run ``[pos, pos + L)``)
``flag16``      uint16    (optional) the SAM FLAG word of every record
``mapq``        uint8     (optional) MAPQ
``qlen``        int32     (optional) ``l_seq``, pysam's ``query_length``
``nh``          uint16    (optional) the ``NH:i`` tag (reported alignments of the query), 0 where a record has none
==============  ========  ====================================================

The three optional columns are what read *filters* may look at beyond strand and length (the reference's filters
take the ``pysam.AlignedSegment``, genome_array.py:697-722): :class:`PackedRead` serves ``flag``,
``mapping_quality``, ``query_length`` and the ``is_*`` properties from them, and the engine evaluates
:class:`~plastid_amd.map_factories.FlagFilterFactory` on ``flag16`` / ``mapq`` in HBM.

Records must be in BAM order: sorted by ``(tid, pos)``, ties in file order.

WIDE records -- more than 65 535 aligned positions or more than 255 aligned runs (long reads; the
reference has no such limit: ``read.positions`` is a Python list) -- carry the markers ``alen = 65535`` and
``nblk = 255`` in the packed arrays and their true values in three short side arrays: ``wide_idx`` (ascending
record indices), ``wide_alen``, ``wide_nblk`` (int32).  :meth:`PackedAlignments.true_alen` /
:meth:`true_nblk` give the true values of every record.

A :class:`PackedAlignments` also duck-types the small part of
``pysam.AlignmentFile`` that ``BAMGenomeArray`` uses (``fetch``, ``references``,
``lengths``, ``mapped``, ``close`` -- genome_array.py:669, 679, 690, 802-807), so
it can be handed to anything that expects an open alignment file.
"""
import numpy as np

FLAG_REVERSE = 0x01
FLAG_EXCLUDED = 0x80

MAX_ALIGNED_LEN = 65535
MAX_RUNS = 255

# BAM CIGAR op codes (SAM spec section 4.2): MIDNSHP=X
_CIGAR_CHARS = "MIDNSHP=X"
_OP_CODE = {c: i for i, c in enumerate(_CIGAR_CHARS)}


def positions_to_runs(positions):
    """Group an ascending list of reference positions into maximal contiguous
    runs ``[(start, length), ...]``."""
    runs = []
    start = prev = None
    for p in positions:
        p = int(p)
        if start is None:
            start = prev = p
        elif p == prev + 1:
            prev = p
        else:
            runs.append((start, prev - start + 1))
            start = prev = p
    if start is not None:
        runs.append((start, prev - start + 1))
    return runs


def parse_cigar_string(cigar):
    """``"10M2D5M"`` -> ``[(0, 10), (2, 2), (0, 5)]`` (BAM op codes)."""
    out = []
    num = ""
    for ch in cigar:
        if ch.isdigit():
            num += ch
        else:
            if ch not in _OP_CODE or not num:
                raise ValueError("Malformed CIGAR string '%s'" % cigar)
            out.append((_OP_CODE[ch], int(num)))
            num = ""
    if num:
        raise ValueError("Malformed CIGAR string '%s'" % cigar)
    return out


def cigar_to_runs(pos, cigartuples):
    """Aligned reference runs of one alignment (SAM spec: M/=/X consume query and
    reference and yield aligned positions; D/N advance the reference only;
    I/S/H/P touch nothing on the reference).

    Returns ``(runs, aligned_len)`` with ``runs = [(start, length), ...]``.
    """
    runs = []
    ref = int(pos)
    L = 0
    for op, n in cigartuples:
        n = int(n)
        if op in (0, 7, 8):
            if n > 0:
                if runs and runs[-1][0] + runs[-1][1] == ref:
                    runs[-1] = (runs[-1][0], runs[-1][1] + n)
                else:
                    runs.append((ref, n))
                ref += n
                L += n
        elif op in (2, 3):
            ref += n
        elif op in (1, 4, 5, 6):
            pass
        else:
            raise ValueError("Unknown CIGAR op code %r" % (op,))
    return runs, L


class PackedRead(object):
    """Minimal read object yielded by :meth:`PackedAlignments.fetch`: carries
    what the reference's mapping functions and filters consume."""

    __slots__ = ("source", "index", "reference_id", "reference_start", "is_reverse", "_runs", "flag", "mapping_quality",
                 "query_length", "_nh")

    def __init__(self, source, index, tid, pos, is_reverse, runs, flag=None, mapq=None, qlen=None, nh=None):
        self.source = source
        self.index = index
        self.reference_id = tid
        self.reference_start = pos
        self.is_reverse = is_reverse
        self._runs = runs
        # a source without the SAM columns (synthetic arrays, stub reads): the strand bit is all the FLAG word is
        # known to hold, MAPQ is 255 ("not available", SAM spec 1.4), the query is as long as its aligned part
        self.flag = int(flag) if flag is not None else (0x10 if is_reverse else 0)
        self.mapping_quality = int(mapq) if mapq is not None else 255
        self.query_length = int(qlen) if qlen is not None else sum(n for _, n in runs)
        self._nh = None if nh is None else int(nh)   # the NH:i tag (None: the source has no such column; 0: this record has no tag)

    # the FLAG bits by pysam's names (kent/src/htslib/htslib/sam.h:110-132)
    is_paired = property(lambda self: bool(self.flag & 0x1))
    is_proper_pair = property(lambda self: bool(self.flag & 0x2))
    is_unmapped = property(lambda self: bool(self.flag & 0x4))
    mate_is_unmapped = property(lambda self: bool(self.flag & 0x8))
    mate_is_reverse = property(lambda self: bool(self.flag & 0x20))
    is_read1 = property(lambda self: bool(self.flag & 0x40))
    is_read2 = property(lambda self: bool(self.flag & 0x80))
    is_secondary = property(lambda self: bool(self.flag & 0x100))
    is_qcfail = property(lambda self: bool(self.flag & 0x200))
    is_duplicate = property(lambda self: bool(self.flag & 0x400))
    is_supplementary = property(lambda self: bool(self.flag & 0x800))
    mapq = property(lambda self: self.mapping_quality)

    # the one auxiliary field the packed columns carry: NH:i, what the usual unique-mapper filter of the reference's users
    # reads (`lambda read: read.get_tag("NH") == 1`, genome_array.py:697-722); pysam's semantics: KeyError without the tag
    def has_tag(self, tag):
        return tag == "NH" and bool(self._nh)

    def get_tag(self, tag, with_value_type=False):
        if tag != "NH" or not self._nh:
            raise KeyError("tag '%s' not present" % tag)
        return (self._nh, "i") if with_value_type else self._nh

    def get_tags(self, with_value_type=False):
        return [("NH",) + ((self._nh, "i") if with_value_type else (self._nh,))] if self._nh else []

    tags = property(lambda self: self.get_tags())

    @property
    def cigartuples(self):
        """``[(op, length), ...]`` in BAM op codes: the aligned runs as ``M`` (0) joined by ``N`` (3) -- or ``D`` (2) for a
        gap of one position, as :func:`tests.bam_writer.write_bam_realistic` writes them; insertions, clips and the
        distinction = / X are not kept by the packed columns."""
        out, prev = [], None
        for s, n in self._runs:
            if prev is not None:
                out.append((2 if s - prev == 1 else 3, s - prev))
            out.append((0, n))
            prev = s + n
        return out

    cigar = property(lambda self: self.cigartuples)

    @property
    def positions(self):
        out = []
        for s, n in self._runs:
            out.extend(range(s, s + n))
        return out

    def get_reference_positions(self):
        return self.positions

    @property
    def reference_end(self):
        if not self._runs:
            return None
        s, n = self._runs[-1]
        return s + n

    def __repr__(self):
        return "<PackedRead #%d tid=%d pos=%d %s runs=%r>" % (
            self.index, self.reference_id, self.reference_start,
            "-" if self.is_reverse else "+", self._runs)


class PackedAlignments(object):
    """One coordinate-sorted alignment file as flat arrays (see module doc)."""

    def __init__(self, tid, pos, alen, flags, nblk, blk_start=None, blk_len=None,
                 references=None, lengths=None, mapped=None, read_objects=None, validate=True,
                 wide_idx=None, wide_alen=None, wide_nblk=None, flag16=None, mapq=None, qlen=None, nh=None):
        self.tid = np.ascontiguousarray(tid, dtype=np.int32)
        self.pos = np.ascontiguousarray(pos, dtype=np.int32)
        self.alen = np.ascontiguousarray(alen, dtype=np.uint16)
        self.flags = np.ascontiguousarray(flags, dtype=np.uint8)
        self.nblk = np.ascontiguousarray(nblk, dtype=np.uint8)
        self.blk_start = np.ascontiguousarray(
            np.zeros(0, np.int32) if blk_start is None else blk_start, dtype=np.int32)
        self.blk_len = np.ascontiguousarray(
            np.zeros(0, np.int32) if blk_len is None else blk_len, dtype=np.int32)
        self.wide_idx = np.ascontiguousarray(np.zeros(0, np.int64) if wide_idx is None else wide_idx, dtype=np.int64)
        self.wide_alen = np.ascontiguousarray(np.zeros(0, np.int32) if wide_alen is None else wide_alen, dtype=np.int32)
        self.wide_nblk = np.ascontiguousarray(np.zeros(0, np.int32) if wide_nblk is None else wide_nblk, dtype=np.int32)
        self.flag16 = None if flag16 is None else np.ascontiguousarray(flag16, dtype=np.uint16)
        self.mapq = None if mapq is None else np.ascontiguousarray(mapq, dtype=np.uint8)
        self.qlen = None if qlen is None else np.ascontiguousarray(qlen, dtype=np.int32)
        self.nh = None if nh is None else np.ascontiguousarray(nh, dtype=np.uint16)
        n = len(self.tid)
        for name in ("flag16", "mapq", "qlen", "nh"):
            col = getattr(self, name)
            if col is not None and len(col) != n:
                raise ValueError("PackedAlignments: array '%s' has wrong length" % name)
        if references is None:
            ntid = int(self.tid.max()) + 1 if n else 1
            references = ["chr%d" % i for i in range(ntid)]
        self.references = tuple(references)
        if lengths is None:
            lengths = [0] * len(self.references)
        self.lengths = tuple(int(x) for x in lengths)
        self.mapped = int(n if mapped is None else mapped)
        self._read_objects = read_objects
        self._blk_off = None
        self._tid_bounds = None
        self._max_span = None
        if validate:
            self.validate()

    # ------------------------------------------------------------------ basic
    def __len__(self):
        return len(self.tid)

    @property
    def n(self):
        return len(self.tid)

    @property
    def n_wide(self):
        return len(self.wide_idx)

    def true_alen(self):
        """Aligned length of every record as int64 (the side arrays patched in for wide records)."""
        a = self.alen.astype(np.int64)
        if self.n_wide:
            a[self.wide_idx] = self.wide_alen
        return a

    def true_nblk(self):
        """Run count of every record as int64 (the side arrays patched in for wide records)."""
        a = self.nblk.astype(np.int64)
        if self.n_wide:
            a[self.wide_idx] = self.wide_nblk
        return a

    def validate(self):
        n = self.n
        for name in ("pos", "alen", "flags", "nblk"):
            if len(getattr(self, name)) != n:
                raise ValueError("PackedAlignments: array '%s' has wrong length" % name)
        if self.n_wide:
            w = self.wide_idx
            if len(self.wide_alen) != len(w) or len(self.wide_nblk) != len(w) or np.any(np.diff(w) <= 0) or w[0] < 0 or w[-1] >= n:
                raise ValueError("PackedAlignments: wide_idx must be ascending record indices with one wide_alen / wide_nblk each")
            if np.any(self.alen[w] != MAX_ALIGNED_LEN) or np.any(self.nblk[w] != MAX_RUNS):
                raise ValueError("PackedAlignments: wide records carry alen 65535 and nblk 255 in the packed arrays")
            return self._validate_general()
        if n:
            if self.tid.min() < 0 or self.tid.max() >= len(self.references):
                raise ValueError("PackedAlignments: tid out of range of `references`")
            key = (self.tid.astype(np.int64) << 32) | self.pos.astype(np.int64)
            if np.any(key[1:] < key[:-1]):
                raise ValueError(
                    "PackedAlignments: records are not sorted by (tid, pos); "
                    "alignment files must be coordinate sorted")
            if self.pos.min() < 0:
                raise ValueError("PackedAlignments: negative alignment start")
        multi = self.nblk >= 2
        if int(self.nblk[multi].astype(np.int64).sum()) != len(self.blk_start) or \
                len(self.blk_start) != len(self.blk_len):
            raise ValueError("PackedAlignments: run arrays do not match `nblk`")
        single = self.nblk == 1
        if np.any(self.alen[single] == 0) or np.any((self.nblk == 0) != (self.alen == 0)):
            raise ValueError("PackedAlignments: nblk/alen mismatch")
        if len(self.blk_start):
            off = self.block_offsets()
            idx = np.nonzero(multi)[0]
            # first run starts at pos; runs ascending and non-adjacent; lengths sum to alen
            if np.any(self.blk_start[off[idx]] != self.pos[idx]):
                raise ValueError("PackedAlignments: first run of a record must start at pos")
            if np.any(self.blk_len <= 0):
                raise ValueError("PackedAlignments: empty run")
            sums = np.add.reduceat(self.blk_len.astype(np.int64), off[idx])
            if np.any(sums != self.alen[idx]):
                raise ValueError("PackedAlignments: run lengths do not sum to alen")
            ends = self.blk_start.astype(np.int64) + self.blk_len
            inner = np.ones(len(self.blk_start), bool)
            inner[off[idx]] = False  # first run of each record has no predecessor
            if np.any(self.blk_start[inner] <= ends[np.nonzero(inner)[0] - 1]):
                raise ValueError("PackedAlignments: runs must be ascending and non-adjacent")

    def _validate_general(self):
        """validate() for a file with wide records: the same checks on the true lengths / run counts."""
        n = self.n
        L, nb = self.true_alen(), self.true_nblk()
        if n:
            if self.tid.min() < 0 or self.tid.max() >= len(self.references):
                raise ValueError("PackedAlignments: tid out of range of `references`")
            key = (self.tid.astype(np.int64) << 32) | self.pos.astype(np.int64)
            if np.any(key[1:] < key[:-1]):
                raise ValueError("PackedAlignments: records are not sorted by (tid, pos); alignment files must be coordinate sorted")
            if self.pos.min() < 0:
                raise ValueError("PackedAlignments: negative alignment start")
        multi = nb >= 2
        if int(nb[multi].sum()) != len(self.blk_start) or len(self.blk_start) != len(self.blk_len):
            raise ValueError("PackedAlignments: run arrays do not match `nblk`")
        if np.any(L[nb == 1] == 0) or np.any((nb == 0) != (L == 0)):
            raise ValueError("PackedAlignments: nblk/alen mismatch")
        if len(self.blk_start):
            off = self.block_offsets()
            idx = np.nonzero(multi)[0]
            if np.any(self.blk_start[off[idx]] != self.pos[idx]):
                raise ValueError("PackedAlignments: first run of a record must start at pos")
            if np.any(self.blk_len <= 0):
                raise ValueError("PackedAlignments: empty run")
            if np.any(np.add.reduceat(self.blk_len.astype(np.int64), off[idx]) != L[idx]):
                raise ValueError("PackedAlignments: run lengths do not sum to alen")
            ends = self.blk_start.astype(np.int64) + self.blk_len
            inner = np.ones(len(self.blk_start), bool)
            inner[off[idx]] = False
            if np.any(self.blk_start[inner] <= ends[np.nonzero(inner)[0] - 1]):
                raise ValueError("PackedAlignments: runs must be ascending and non-adjacent")

    def block_offsets(self):
        """``off[i]`` = index of record ``i``'s first run in ``blk_*`` (meaningful
        for ``nblk >= 2`` records)."""
        if self._blk_off is None:
            nb = self.true_nblk() if self.n_wide else self.nblk
            cnt = np.where(nb >= 2, nb, 0).astype(np.int64)
            off = np.zeros(self.n + 1, np.int64)
            np.cumsum(cnt, out=off[1:])
            self._blk_off = off
        return self._blk_off

    def tid_bounds(self):
        """``b[t]:b[t+1]`` = record range of contig ``t``."""
        if self._tid_bounds is None:
            self._tid_bounds = np.searchsorted(
                self.tid, np.arange(len(self.references) + 1), side="left").astype(np.int64)
        return self._tid_bounds

    def ref_end(self):
        """htslib ``bam_endpos``: one past the last aligned reference position
        (``pos + 1`` for records without aligned bases)."""
        L = self.true_alen() if self.n_wide else self.alen.astype(np.int64)
        nb = self.true_nblk() if self.n_wide else self.nblk
        end = self.pos.astype(np.int64) + np.maximum(L, 1)
        multi = np.nonzero(nb >= 2)[0]
        if len(multi):
            off = self.block_offsets()
            last = off[multi] + nb[multi] - 1
            end[multi] = self.blk_start[last].astype(np.int64) + self.blk_len[last]
        return end

    def _true_of(self, i):
        """``(aligned length, run count)`` of record `i`."""
        if self.n_wide and self.alen[i] == MAX_ALIGNED_LEN and self.nblk[i] == MAX_RUNS:
            k = int(np.searchsorted(self.wide_idx, i))
            if k < self.n_wide and self.wide_idx[k] == i:
                return int(self.wide_alen[k]), int(self.wide_nblk[k])
        return int(self.alen[i]), int(self.nblk[i])

    def runs_of(self, i):
        L, nb = self._true_of(i)
        if nb >= 2:
            o = int(self.block_offsets()[i])
            return [(int(self.blk_start[o + b]), int(self.blk_len[o + b]))
                    for b in range(nb)]
        if L == 0:
            return []
        return [(int(self.pos[i]), L)]

    def read(self, i):
        """Read object for record ``i`` (the original object if this file was
        packed from read objects, else a :class:`PackedRead`)."""
        if self._read_objects is not None:
            return self._read_objects[i]
        return PackedRead(self, int(i), int(self.tid[i]), int(self.pos[i]),
                          bool(self.flags[i] & FLAG_REVERSE), self.runs_of(i),
                          None if self.flag16 is None else self.flag16[i], None if self.mapq is None else self.mapq[i],
                          None if self.qlen is None else self.qlen[i], None if self.nh is None else self.nh[i])

    def sam_columns(self, idx):
        """The optional SAM columns of the records `idx` (index array or slice) as constructor keywords."""
        return {name: None if getattr(self, name) is None else getattr(self, name)[idx] for name in ("flag16", "mapq", "qlen", "nh")}

    # ------------------------------------------- pysam.AlignmentFile duck type
    def fetch(self, reference=None, start=None, end=None, **kwargs):
        """Records overlapping ``reference:start-end`` in file order (htslib rule:
        ``pos < end and endpos > start``)."""
        for i in self.fetch_indices(reference, start, end):
            yield self.read(i)

    def fetch_indices(self, reference=None, start=None, end=None):
        if reference is None:
            return np.arange(self.n)
        try:
            t = self.references.index(reference)
        except ValueError:
            raise ValueError("invalid reference `%s`" % reference)
        b = self.tid_bounds()
        lo, hi = int(b[t]), int(b[t + 1])
        if start is None:
            start = 0
        if end is None:
            end = np.iinfo(np.int64).max
        hi = lo + int(np.searchsorted(self.pos[lo:hi], end, side="left"))
        if self._max_span is None:
            self._max_span = int((self.ref_end() - self.pos).max()) if self.n else 1
        lo2 = lo + int(np.searchsorted(self.pos[lo:hi], start - self._max_span, side="right"))
        idx = np.arange(lo2, hi)
        if len(idx):
            idx = idx[self.ref_end()[lo2:hi] > start]
        return idx

    def close(self):
        pass

    # ------------------------------------------------------------ constructors
    @classmethod
    def from_runs(cls, tids, is_reverse, runs_per_read, references=None, lengths=None,
                  mapped=None, read_objects=None, sort=False, positions=None, flag16=None, mapq=None, qlen=None, nh=None):
        """Build from per-read lists of aligned runs ``[(start, len), ...]``.  `positions`
        (optional) places the records that have no aligned base at all."""
        n = len(runs_per_read)
        tid = np.asarray(tids, dtype=np.int32) if np.ndim(tids) else np.full(n, tids, np.int32)
        pos = np.zeros(n, np.int32)
        alen = np.zeros(n, np.int64)
        nblk = np.zeros(n, np.int64)
        flags = np.zeros(n, np.uint8)
        flags[np.asarray(is_reverse, dtype=bool)] = FLAG_REVERSE
        order = None
        for i, runs in enumerate(runs_per_read):
            nblk[i] = len(runs)
            if runs:
                pos[i] = runs[0][0]
                alen[i] = sum(r[1] for r in runs)
            elif positions is not None:
                pos[i] = positions[i]
        if alen.max(initial=0) > 0x7fffffff:
            raise ValueError("alignments with more than 2^31 - 1 aligned positions are not supported")
        if sort and n:
            key = (tid.astype(np.int64) << 32) | pos.astype(np.int64)
            order = np.argsort(key, kind="stable")
            tid, pos, alen, nblk, flags = tid[order], pos[order], alen[order], nblk[order], flags[order]
            runs_per_read = [runs_per_read[j] for j in order]
            if read_objects is not None:
                read_objects = [read_objects[j] for j in order]
        bs, bl = [], []
        for runs in runs_per_read:
            if len(runs) >= 2:
                for s, ln in runs:
                    bs.append(s)
                    bl.append(ln)
        # records beyond the 16-bit / 8-bit fields keep markers there and their true values aside (module doc)
        wide = np.nonzero((alen > MAX_ALIGNED_LEN) | (nblk > MAX_RUNS) | ((alen == MAX_ALIGNED_LEN) & (nblk == MAX_RUNS)))[0]
        a16, n8 = alen.copy(), nblk.copy()
        a16[wide] = MAX_ALIGNED_LEN
        n8[wide] = MAX_RUNS
        sam = {}
        for name, col in (("flag16", flag16), ("mapq", mapq), ("qlen", qlen), ("nh", nh)):
            if col is not None:
                col = np.asarray(col)
                sam[name] = col[order] if order is not None else col
        out = cls(tid, pos, a16.astype(np.uint16), flags, n8.astype(np.uint8), bs, bl,
                  references=references, lengths=lengths, mapped=mapped, read_objects=read_objects,
                  wide_idx=wide, wide_alen=alen[wide], wide_nblk=nblk[wide], **sam)
        out.sort_order = order
        return out

    @classmethod
    def from_reads(cls, reads, tids=None, **kwargs):
        """Build from read objects exposing ``positions`` and ``is_reverse`` (and
        ``reference_id`` unless `tids` is given) -- e.g. ``pysam.AlignedSegment``."""
        reads = list(reads)
        runs = [positions_to_runs(r.positions) for r in reads]
        rev = [bool(r.is_reverse) for r in reads]
        if tids is None:
            tids = [int(getattr(r, "reference_id", 0)) for r in reads]
            tids = [t if t >= 0 else 0 for t in tids]
        # reads that know their FLAG word / MAPQ / query length (pysam.AlignedSegment does) keep them
        if reads and all(hasattr(r, "flag") and hasattr(r, "mapping_quality") for r in reads):
            kwargs.setdefault("flag16", np.array([int(r.flag) & 0xffff for r in reads], np.uint16))
            kwargs.setdefault("mapq", np.array([int(r.mapping_quality) & 0xff for r in reads], np.uint8))
            if all(getattr(r, "query_length", None) is not None for r in reads):
                kwargs.setdefault("qlen", np.array([int(r.query_length) for r in reads], np.int32))
        if reads and all(hasattr(r, "has_tag") and hasattr(r, "get_tag") for r in reads):   # (pysam reads: NH:i where present)
            kwargs.setdefault("nh", np.array([min(max(int(r.get_tag("NH")), 0), 65535) if r.has_tag("NH") else 0 for r in reads], np.uint16))
        return cls.from_runs(tids, rev, runs, read_objects=reads, **kwargs)

    @classmethod
    def from_cigars(cls, tids, pos, cigars, is_reverse, **kwargs):
        """Build from ``(pos, CIGAR)`` pairs; `cigars` are strings or lists of
        ``(op, length)`` tuples in BAM op codes."""
        runs = []
        for p, cg in zip(pos, cigars):
            if isinstance(cg, str):
                cg = parse_cigar_string(cg)
            r, _ = cigar_to_runs(p, cg)
            if not r:
                # no aligned bases: keep the placement so fetch() semantics survive
                runs.append([])
            else:
                runs.append(r)
        # records without aligned bases keep their stated position
        return cls.from_runs(tids, is_reverse, runs, positions=pos, **kwargs)

    @classmethod
    def from_ungapped(cls, tid, pos, alen, is_reverse, **kwargs):
        """Fast path: every read is a single ``<L>M`` run (arrays in, no Python loop)."""
        pos = np.asarray(pos, dtype=np.int32)
        alen = np.asarray(alen)
        n = len(pos)
        tid = np.asarray(tid, dtype=np.int32) if np.ndim(tid) else np.full(n, tid, np.int32)
        flags = np.where(np.asarray(is_reverse, dtype=bool), FLAG_REVERSE, 0).astype(np.uint8)
        nblk = (alen > 0).astype(np.uint8)
        return cls(tid, pos, alen.astype(np.uint16), flags, nblk, **kwargs)

    def subset(self, mask_or_index, validate=True):
        """New :class:`PackedAlignments` with the selected records, in the order given (``validate=False``: the
        caller re-orders the records and re-labels the contigs itself)."""
        idx = np.asarray(mask_or_index)
        if idx.dtype == bool:
            idx = np.nonzero(idx)[0]
        off = self.block_offsets()
        nb = self.true_nblk() if self.n_wide else self.nblk
        multi = idx[nb[idx] >= 2]
        if len(multi):
            sel = np.concatenate([np.arange(off[i], off[i] + nb[i]) for i in multi])
        else:
            sel = np.zeros(0, np.int64)
        ro = None if self._read_objects is None else [self._read_objects[i] for i in idx]
        wide = {}
        if self.n_wide:   # the selected wide records, at their new indices
            is_w = np.zeros(self.n, bool)
            is_w[self.wide_idx] = True
            new_w = np.nonzero(is_w[idx])[0]
            src = np.searchsorted(self.wide_idx, idx[new_w])
            wide = dict(wide_idx=new_w, wide_alen=self.wide_alen[src], wide_nblk=self.wide_nblk[src])
        return PackedAlignments(self.tid[idx], self.pos[idx], self.alen[idx], self.flags[idx],
                                self.nblk[idx], self.blk_start[sel], self.blk_len[sel],
                                references=self.references, lengths=self.lengths,
                                mapped=len(idx), read_objects=ro, validate=validate, **dict(wide, **self.sam_columns(idx)))


    def slice(self, i0, i1):
        """Records ``[i0, i1)`` as a new :class:`PackedAlignments` (views, no Python loop): a genome
        range of a coordinate-sorted file is a contiguous record range."""
        i0, i1 = int(i0), int(i1)
        off = self.block_offsets()
        b0, b1 = int(off[i0]), int(off[i1])
        ro = None if self._read_objects is None else self._read_objects[i0:i1]
        wide = {}
        if self.n_wide:
            k0, k1 = np.searchsorted(self.wide_idx, [i0, i1])
            wide = dict(wide_idx=self.wide_idx[k0:k1] - i0, wide_alen=self.wide_alen[k0:k1], wide_nblk=self.wide_nblk[k0:k1])
        return PackedAlignments(self.tid[i0:i1], self.pos[i0:i1], self.alen[i0:i1], self.flags[i0:i1],
                                self.nblk[i0:i1], self.blk_start[b0:b1], self.blk_len[b0:b1],
                                references=self.references, lengths=self.lengths, mapped=i1 - i0,
                                read_objects=ro, validate=False, **dict(wide, **self.sam_columns(slice(i0, i1))))


def concat_file_major(files):
    """Concatenate packed files into the single file-major record list the
    reference iterates (``itertools.chain`` over ``bamfile.fetch``,
    genome_array.py:800-809).  Returns a dict of arrays incl. ``file_id``."""
    if not files:
        raise ValueError("no alignment files")
    base = np.cumsum([0] + [f.n for f in files])
    wide = {}
    if any(getattr(f, "n_wide", 0) for f in files):   # the wide records of all files, at their indices in the concatenation
        wide = {"wide_idx": np.concatenate([f.wide_idx + base[k] for k, f in enumerate(files)]),
                "wide_alen": np.concatenate([f.wide_alen for f in files]), "wide_nblk": np.concatenate([f.wide_nblk for f in files])}
    if all(getattr(f, "flag16", None) is not None and getattr(f, "mapq", None) is not None for f in files):
        wide["flag16"] = np.concatenate([f.flag16 for f in files])
        wide["mapq"] = np.concatenate([f.mapq for f in files])
    if all(getattr(f, "nh", None) is not None for f in files):
        wide["nh"] = np.concatenate([f.nh for f in files])
    return dict(wide, **{
        "tid": np.concatenate([f.tid for f in files]),
        "pos": np.concatenate([f.pos for f in files]),
        "alen": np.concatenate([f.alen for f in files]),
        "flags": np.concatenate([f.flags for f in files]),
        "nblk": np.concatenate([f.nblk for f in files]),
        "file_id": np.concatenate([np.full(f.n, k, np.uint8) for k, f in enumerate(files)]),
        "blk_start": np.concatenate([f.blk_start for f in files]),
        "blk_len": np.concatenate([f.blk_len for f in files]),
    })
