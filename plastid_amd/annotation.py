"""Annotation -> interval table: the step on the *other* side of the counting path.

The reference builds one Python object per transcript (``BED_Reader`` ->
``SegmentChain.from_bed``, plastid/readers/bed.py:88-356, plastid/genomics/roitools.pyx:534-741,
3421-3470) and then loops over them.  For the GPU path the useful form is a flat table --
``(tid, start, end, strand, chain, spliced offset)`` per exon -- that becomes ONE counting
plan for the whole annotation.  :class:`IntervalTable` is that table; it can be built from
BED text, from ``SegmentChain`` objects, or directly from arrays, and hands back per-chain
views of the batched result.
"""
import numpy as np

from .exceptions import DataWarning, warn

STRAND_CODE = {"\x00": 0, "+": 1, "-": 2, ".": 3}
STRAND_CHAR = {0: "\x00", 1: "+", 2: "-", 3: "."}


def _bed_fields(line):
    """chrom, strand, name and exon list of one BED3-BED12 line (standard columns,
    roitools.pyx:534-617 + 711-741).  As in the reference, blocks are taken as given."""
    items = line.strip("\n").split("\t")
    n = len(items)
    if n < 3:
        raise ValueError("BED format requires at least 3 columns. Found only %s.\n\t    %s" % (n, items))
    chrom = items[0]
    chrom_start, chrom_end = int(items[1]), int(items[2])
    strand = "." if n < 6 else items[5]
    name = items[3] if n > 3 else "%s:%s-%s(%s)" % (chrom, chrom_start, chrom_end, strand)
    if n >= 12:
        try:
            nblocks = int(items[9])
            sizes = items[10].strip(",").split(",")
            starts = items[11].strip(",").split(",")
            exons = [(chrom_start + int(starts[i]), chrom_start + int(starts[i]) + int(sizes[i]))
                     for i in range(nblocks)]
        except (ValueError, IndexError):
            raise ValueError("Could not parse BED line:\n\t    '%s'" % line)
    else:
        exons = [(chrom_start, chrom_end)]
    attr = {"ID": name}
    if n > 4:
        try:
            attr["score"] = float(items[4])
        except ValueError:
            warn("get_standard_bed_attr: Could not format column %s with 'float'. Falling back to default value 'nan'."
                 % items[4], DataWarning)
            attr["score"] = float("nan")
    if n > 7:
        try:
            ts, te = int(items[6]), int(items[7])
        except ValueError:
            ts = te = -1
        if ts == te or ts < 0 or te < 0:
            ts = te = chrom_start
        attr["thickstart"], attr["thickend"] = ts, te
    return chrom, strand, exons, attr


def bed_line_to_chain(line, cls):
    """``SegmentChain.from_bed`` (roitools.pyx:3421-3470): no sorting/merging of the blocks."""
    from .roitools import GenomicSegment
    chrom, strand, exons, attr = _bed_fields(line)
    chain = cls()
    chain._set_segments([GenomicSegment(chrom, s, e, strand) for s, e in exons])
    chain.attr.update(attr)
    return chain


def iter_bed_lines(stream):
    """Data lines of a BED stream (``BED_Reader._assemble``, readers/bed.py:322-337:
    ``browser``/``track``/``#`` lines and blank lines are not features)."""
    for line in stream:
        if not line.strip() or line.startswith(("browser", "track", "#")):
            continue
        yield line


def read_bed(path_or_stream, cls=None):
    """List of |SegmentChains| from a BED file (one object per line, as ``BED_Reader`` yields)."""
    from .roitools import SegmentChain
    cls = SegmentChain if cls is None else cls
    if isinstance(path_or_stream, str):
        with open(path_or_stream) as fh:
            return [bed_line_to_chain(l, cls) for l in iter_bed_lines(fh)]
    return [bed_line_to_chain(l, cls) for l in iter_bed_lines(path_or_stream)]


class IntervalTable(object):
    """CSR table of chains: exons ``ex_start/ex_end`` (genomic, ascending, non-overlapping) of
    chain ``c`` are ``ex_off[c]:ex_off[c+1]``; ``tid`` indexes ``references``;
    ``strand`` uses plastid's codes (1 '+', 2 '-', 3 '.')."""

    def __init__(self, names, lengths, tid, strand, ex_off, ex_start, ex_end, ids=None):
        self.references = list(names)
        self.ref_lengths = list(lengths) if lengths is not None else [0] * len(self.references)
        self.tid = np.asarray(tid, np.int32)
        self.strand = np.asarray(strand, np.uint8)
        self.ex_off = np.asarray(ex_off, np.int64)
        self.ex_start = np.asarray(ex_start, np.int64)
        self.ex_end = np.asarray(ex_end, np.int64)
        self.ids = ids
        self.n = len(self.tid)
        ex_len = self.ex_end - self.ex_start
        self.ex_cum = np.zeros(len(ex_len) + 1, np.int64)  # running spliced offset over all exons
        np.cumsum(ex_len, out=self.ex_cum[1:])
        self.length = self.ex_cum[self.ex_off[1:]] - self.ex_cum[self.ex_off[:-1]]
        self.ex_tx = np.repeat(np.arange(self.n), np.diff(self.ex_off))

    # ------------------------------------------------------------ constructors
    @classmethod
    def from_chains(cls, chains, references):
        """From |SegmentChain| objects; chains on contigs not in `references` get tid -1
        (counted as zeros, genome_array.py:795-798)."""
        index = {r: i for i, r in enumerate(references)}
        tid, strand, ex_off, s, e, ids = [], [], [0], [], [], []
        for c in chains:
            tid.append(index.get(c.chrom, -1))
            strand.append(c.c_strand)
            for seg in c:
                s.append(seg.start)
                e.append(seg.end)
            ex_off.append(len(s))
            ids.append(c.get_name())
        return cls(references, None, tid, strand, ex_off, s, e, ids=ids)

    @classmethod
    def from_bed(cls, path_or_stream, references):
        """Straight from BED text, no per-feature Python objects.  Exons are sorted within a
        chain; overlapping/adjacent blocks are left as given (``from_bed`` does not merge)."""
        index = {r: i for i, r in enumerate(references)}
        tid, strand, ex_off, s, e, ids = [], [], [0], [], [], []
        opened = isinstance(path_or_stream, str)
        fh = open(path_or_stream) if opened else path_or_stream
        try:
            for line in iter_bed_lines(fh):
                chrom, st, exons, attr = _bed_fields(line)
                tid.append(index.get(chrom, -1))
                strand.append(STRAND_CODE.get(st, 3))
                for a, b in sorted(exons):
                    s.append(a)
                    e.append(b)
                ex_off.append(len(s))
                ids.append(attr["ID"])
        finally:
            if opened:
                fh.close()
        return cls(references, None, tid, strand, ex_off, s, e, ids=ids)

    # ------------------------------------------------------------------ views
    @property
    def n_segments(self):
        return len(self.ex_start)

    @property
    def n_positions(self):
        return int(self.length.sum())

    def subset(self, idx):
        idx = np.asarray(idx)
        cnt = np.diff(self.ex_off)[idx]
        sel = np.concatenate([np.arange(self.ex_off[i], self.ex_off[i + 1]) for i in idx]) if len(idx) else \
            np.zeros(0, np.int64)
        off = np.zeros(len(idx) + 1, np.int64)
        np.cumsum(cnt, out=off[1:])
        ids = None if self.ids is None else [self.ids[i] for i in idx]
        return type(self)(self.references, self.ref_lengths, self.tid[idx], self.strand[idx], off,
                          self.ex_start[sel], self.ex_end[sel], ids=ids)

    def plan_arrays(self, rows=1, stranded=True):
        """Segment table + output layout of ``chain.get_counts`` for every chain: each chain is a
        ``[rows, length]`` block; '-' chains are laid out 5'->3' (roitools.pyx:3259-3271)."""
        seg_tx = self.ex_tx
        seg_len = self.ex_end - self.ex_start
        off_in_tx = self.ex_cum[:-1] - self.ex_cum[self.ex_off[:-1]][seg_tx]  # spliced offset within the chain
        chain_base = np.zeros(self.n + 1, np.int64)
        np.cumsum(self.length * rows, out=chain_base[1:])
        tx_len = self.length[seg_tx]
        rev = (self.strand[seg_tx] == 2) & bool(stranded)
        out_off = np.where(rev, chain_base[:-1][seg_tx] + tx_len - 1 - off_in_tx,
                           chain_base[:-1][seg_tx] + off_in_tx)
        out_step = np.where(rev, -1, 1).astype(np.int8)
        return dict(tid=self.tid[seg_tx].astype(np.int32), start=self.ex_start.copy(), end=self.ex_end.copy(),
                    strand=self.strand[seg_tx].astype(np.uint8), out_off=out_off.astype(np.int64),
                    out_step=out_step, row_stride=tx_len.astype(np.int64), out_elems=int(chain_base[-1]),
                    chain_base=chain_base, seg_len=seg_len)

    def split_counts(self, flat, rows=1):
        """Per-chain views (``[length]`` or ``[rows, length]``) of a batched result."""
        base = np.zeros(self.n + 1, np.int64)
        np.cumsum(self.length * rows, out=base[1:])
        if rows == 1:
            return [flat[base[c]:base[c + 1]] for c in range(self.n)]
        return [flat[base[c]:base[c + 1]].reshape(rows, int(self.length[c])) for c in range(self.n)]

    def chains(self, limit=None):
        """|SegmentChain| objects (Python objects: use for small sets only)."""
        from .roitools import GenomicSegment, SegmentChain
        out = []
        for t in range(self.n if limit is None else min(limit, self.n)):
            s = STRAND_CHAR[int(self.strand[t])]
            chrom = self.references[self.tid[t]] if self.tid[t] >= 0 else "?"
            segs = [GenomicSegment(chrom, int(self.ex_start[j]), int(self.ex_end[j]), s)
                    for j in range(self.ex_off[t], self.ex_off[t + 1])]
            c = SegmentChain()
            c._set_segments(segs)
            if self.ids is not None:
                c.attr["ID"] = self.ids[t]
            out.append(c)
        return out
