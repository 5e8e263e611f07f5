"""Annotation -> interval table: the step on the *other* side of the counting path.

The reference builds one Python object per transcript (``BED_Reader`` ->
``SegmentChain.from_bed``, plastid/readers/bed.py:88-356, plastid/genomics/roitools.pyx:534-741,
3421-3470) and then loops over them.  For the GPU path the useful form is a flat table --
``(tid, start, end, strand, chain, spliced offset)`` per exon -- that becomes ONE counting
plan for the whole annotation.  :class:`IntervalTable` is that table; it can be built from
BED text, from ``SegmentChain`` objects, or directly from arrays, and hands back per-chain
views of the batched result.
"""
import re

import numpy as np

from .exceptions import DataWarning, FileFormatWarning, warn

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


_GTF2_EXON_LIKE = ("exon", "5UTR", "3UTR", "CDS", "start_codon", "stop_codon")
_GTF2_CDS_LIKE = ("CDS", "start_codon", "stop_codon")
_GTF2_TOKEN = re.compile(r'\s*([^\s";]+)\s+(?:"((?:[^"\\]|\\.)*)"|([^\s;]+))\s*(?:;|$)')


def _gtf2_attributes(text):
    """Ninth GTF2 column -> dict (``parse_GTF2_tokens``, readers/gff_tokens.py:537-600):
    ``key "value";`` pairs, values may hold semicolons inside the quotes, repeated keys
    are joined with a comma."""
    out = {}
    text = text.strip()
    pairs = None
    if "\\" not in text:
        # fast path (no backslashes): split at the semicolons; a token with an odd number of quotes means a
        # semicolon sat inside a quoted value -- then the general tokenizer below takes over
        ok = True
        pairs = []
        for tok in text.split(";"):
            tok = tok.strip()
            if not tok:
                continue
            if tok.count('"') & 1:
                ok = False
                break
            key, sep, val = tok.partition(" ")
            val = val.strip()
            if not sep or not val or (val[0] == '"') != (val[-1] == '"') or (val[0] != '"' and (" " in val or "\t" in val)) \
                    or '"' in key or "\t" in key or (val[0] == '"' and ('"' in val[1:-1] or len(val) < 2)):
                ok = False
                break
            if val[0] == '"':
                val = val[1:-1]
            pairs.append((key, val))
        if not ok:
            pairs = None
    if pairs is None:
        pairs = []
        for m in _GTF2_TOKEN.finditer(text):
            pairs.append((m.group(1), m.group(2) if m.group(2) is not None else m.group(3)))
    esc = "%" in text
    for key, val in pairs:
        if esc:   # unescape_GTF2 on keys and values (gff_tokens.py:582-599)
            key, val = _unescape(key, True), _unescape(val, True)
        if key in out:
            warn("Found duplicate attribute key '%s' in GTF2 line. Catenating value with previous value for key in attr dict:\n    %s"
                 % (key, text), FileFormatWarning)
            out[key] = "%s,%s" % (out[key], val)
        else:
            out[key] = val
    return out


def iter_gtf2_features(stream):
    """(chrom, type, start, end, strand, attr) of every data line; coordinates converted
    from GTF2's 1-based closed to 0-based half-open (readers/gff.py:305-340)."""
    for line in stream:
        if not line.strip() or line.startswith("#"):
            continue
        items = line.rstrip("\n").split("\t")
        if len(items) < 9:
            raise ValueError("GTF2 format requires 9 columns. Found only %s.\n\t    %s" % (len(items), items))
        yield (items[0], items[2], int(items[3]) - 1, int(items[4]), items[6],
               _gtf2_attributes(items[8]))


def _assemble_gtf2(stream):
    """Group exon-like features by ``transcript_id`` (``GTF2_TranscriptAssembler``,
    readers/gff.py:1089-1206): ``exon``/UTR/CDS/codon features build the exon chain (CDS
    features alone imply the exons), blocks are sorted and overlapping/adjacent ones merged as
    the ``Transcript`` constructor does (roitools.pyx:1450-1498), transcripts with blocks on
    several chromosomes or strands are rejected with a ``DataWarning``, and the result is
    sorted the way |SegmentChains| compare (roitools.pyx:835-853).

    Returns a list of ``(chrom, strand, [(start, end), ...], attr)``."""
    exons, cds, attrs = {}, {}, {}
    for chrom, ftype, start, end, strand, attr in stream:
        if ftype not in _GTF2_EXON_LIKE:
            continue
        tname = attr.get("transcript_id")
        exons.setdefault(tname, []).append((chrom, start, end, strand))
        if ftype in _GTF2_CDS_LIKE:
            cds.setdefault(tname, []).append((start, end))
        common = attrs.get(tname)
        if common is None:
            attrs[tname] = dict(attr)
        else:  # keep what every component agrees on (get_identical_attributes, gff.py:1008-1045)
            for k in [k for k in common if attr.get(k, None) != common[k]]:
                del common[k]
    out = []
    for tname, blocks in exons.items():
        if len(set((b[0], b[3]) for b in blocks)) > 1:
            warn("Rejecting transcript '%s' because it contains exons on multiple chromosomes or strands."
                 % tname, DataWarning)
            continue
        merged = []
        for _, s, e, _ in sorted(blocks):
            if merged and s <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], e)
            else:
                merged.append([s, e])
        attr = attrs[tname]
        attr["type"] = "mRNA"
        if tname in cds:
            ordered = sorted(cds[tname])   # first start / last end of the sorted CDS features (gff.py:1175-1178)
            attr["cds_genome_start"], attr["cds_genome_end"] = ordered[0][0], ordered[-1][1]
        out.append((blocks[0][0], blocks[0][3], [tuple(m) for m in merged], attr))
    out.sort(key=lambda t: (t[0], t[2][0][0], t[2][-1][1], STRAND_CODE.get(t[1], 3),
                            sum(e - s for s, e in t[2]), str(t[3].get("transcript_id"))))
    return out


# ---------------------------------------------------------------------------- GFF3
# feature types of the reference's default schema (Sequence Ontology 2.5.3 subset, readers/gff.py:203-345)
def _load_terms(name):
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", name)
    with open(path) as fh:
        return frozenset(line.strip() for line in fh if line.strip() and not line.startswith("#"))


GFF3_TRANSCRIPT_TYPES = _load_terms("gff3_transcript_types.txt")
GFF3_EXON_TYPES = frozenset(["exon", "coding_exon", "noncoding_exon", "exon_of_single_exon_gene", "interior_exon",
                             "interior_coding_exon", "five_prime_coding_exon",
                             "three_prime_coding_exonfive_prime_noncoding_exon",   # (sic: the reference's list lacks a comma)
                             "three_prime_noncoding_exon", "pseudogenic_exon"])
GFF3_CDS_TYPES = frozenset(["CDS", "CDS_fragment", "CDS_indpendently_known", "CDS_predicted"])
_GFF3_LIST_KEYS = ("Parent", "Alias", "Note", "Dbxref", "Ontology_term", "dbxref")


# percent escapes the reference undoes (readers/gff_tokens.py:44-125): "%;,=&", the control characters
# 0x00-0x1f, 0x7f and 0x80-0x9f -- upper-case hex only -- and, in GTF2, the double quote.  Every other
# "%XX" stays as it is ("%25" -> "%" is applied last there and nothing is rescanned, so one pass equals it).
_ESCAPED_CODES = frozenset("%%%02X" % c for c in ([ord(x) for x in "%;,=&"] + list(range(0x20)) + [0x7f] + list(range(0x80, 0xa0))))
_ESCAPE_RE = re.compile(r"%[0-9A-F]{2}")


def _unescape(text, gtf2=False):
    if "%" not in text:
        return text
    def repl(m):
        code = m.group(0)
        if code in _ESCAPED_CODES or (gtf2 and code == "%22"):
            return chr(int(code[1:], 16))
        return code
    return _ESCAPE_RE.sub(repl, text)


def _gff3_unescape(text):
    return _unescape(text)


def _gff3_attributes(text):
    """Ninth GFF3 column -> dict (``parse_GFF3_tokens``, readers/gff_tokens.py:460-535): ``key=value``
    pairs, URL-unescaped; Parent / Alias / Note / Dbxref / Ontology_term are comma-separated lists."""
    out = {}
    for item in text.strip("\n").strip(";").split(";"):
        if not item:
            continue
        key, val = item.split("=")
        key = _gff3_unescape(key.strip(" "))
        if key in _GFF3_LIST_KEYS:
            val = [_gff3_unescape(x) for x in val.strip(" ").split(",")]
        else:
            val = _gff3_unescape(val.strip(" "))
        if key in out:
            val = "%s,%s" % (out[key], val)
        out[key] = val
    return out


def _gff3_batches(stream):
    """Lists of (chrom, type, start, end, strand, attr) between ``###`` lines (the signal that
    everything read so far may be assembled, readers/gff.py:925-988); ``##FASTA`` ends the features."""
    batch = []
    for line in stream:
        if line.startswith("##FASTA") or line.startswith("###FASTA"):
            break
        if line.startswith("###"):
            yield batch
            batch = []
            continue
        if not line.strip() or line.startswith("#"):
            continue
        items = line.rstrip("\n").split("\t")
        if len(items) < 9:
            raise ValueError("GFF3 format requires 9 columns. Found only %s.\n\t    %s" % (len(items), items))
        attr = _gff3_attributes(items[8])
        batch.append((items[0], items[2], int(items[3]) - 1, int(items[4]), items[6], attr))
    yield batch


def _identical_attributes(dicts):
    out = None
    for d in dicts:
        if out is None:
            out = dict(d)
        else:
            for k in [k for k in out if d.get(k, None) != out[k]]:
                del out[k]
    return out or {}


def _assemble_gff3(features, transcript_types=GFF3_TRANSCRIPT_TYPES, exon_types=GFF3_EXON_TYPES,
                   cds_types=GFF3_CDS_TYPES):
    """One batch of ``GFF3_TranscriptAssembler`` (readers/gff.py:1417-1565): exon- and CDS-type
    features are grouped under every ``Parent`` they name (or, lacking one, under their shared ``ID``);
    a group becomes a transcript whose blocks are the merged exon and CDS spans; its attributes come
    from the transcript-type feature of that ID when there is one, else from what all components
    agree on.  Transcript features without exon/CDS children yield nothing, groups on several
    chromosomes or strands are rejected with a ``DataWarning``.  Sorted like |SegmentChains|."""
    tx_features, comp, cds_of = {}, {}, {}
    for chrom, ftype, start, end, strand, attr in features:
        name = attr.get("ID", attr.get("Name", attr.get("name", "%s:%s-%s(%s)" % (chrom, start, end, strand))))
        if ftype in transcript_types:
            tx_features.setdefault(name, []).append(dict(attr, type=ftype))
        elif ftype in exon_types or ftype in cds_types:
            parents = attr.get("Parent")
            if parents is None:
                if "ID" not in attr:
                    warn("Found %s at %s:%s-%s(%s) with no `Parent` or `ID`. Ignoring." % (ftype, chrom, start, end, strand),
                         DataWarning)
                    continue
                parents = [attr["ID"]]
            full = dict(attr, type=ftype)
            for tname in parents:
                comp.setdefault(tname, []).append((chrom, start, end, strand, full))
                if ftype in cds_types:
                    cds_of.setdefault(tname, []).append((chrom, start, end, STRAND_CODE.get(strand, 3)))
    out = []
    for tname, blocks in comp.items():
        if tname in tx_features:
            attr = dict(tx_features[tname][0])
            attr["gene_id"] = ",".join(sorted(attr.get("Parent", [tname])))
        else:
            attr = _identical_attributes([b[4] for b in blocks])
            attr["type"] = "mRNA"
        attr["ID"] = attr["transcript_id"] = tname
        if tname in cds_of:
            ordered = sorted(cds_of[tname])
            attr["cds_genome_start"], attr["cds_genome_end"] = ordered[0][1], ordered[-1][2]
        if len(set((b[0], b[3]) for b in blocks)) > 1:
            warn("Rejecting transcript '%s' because it contains exons  on multiple strands." % tname, DataWarning)
            continue
        merged = []
        for _, s, e, _, _ in sorted(blocks, key=lambda b: (b[1], b[2])):
            if merged and s <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], e)
            else:
                merged.append([s, e])
        out.append((blocks[0][0], blocks[0][3], [tuple(m) for m in merged], attr))
    out.sort(key=lambda t: (t[0], t[2][0][0], t[2][-1][1], STRAND_CODE.get(t[1], 3),
                            sum(e - s for s, e in t[2]), str(t[3]["transcript_id"])))
    return out


def _gff3_rows(path_or_stream, **types):
    fh, opened = _open_text(path_or_stream)
    try:
        rows = []
        for batch in _gff3_batches(fh):
            rows.extend(_assemble_gff3(batch, **types))
        return rows
    finally:
        if opened:
            fh.close()


def read_gff3(path_or_stream, cls=None, **types):
    """Transcripts of a GFF3 file as |SegmentChains|, in the order ``GFF3_TranscriptAssembler``
    yields them (every ``###``-delimited batch sorted)."""
    from .roitools import GenomicSegment, SegmentChain
    cls = SegmentChain if cls is None else cls
    chains = []
    for chrom, strand, blocks, attr in _gff3_rows(path_or_stream, **types):
        chain = cls()
        chain._set_segments([GenomicSegment(chrom, s, e, strand) for s, e in blocks])
        chain.attr.update(attr)
        chains.append(chain)
    return chains


def _open_text(path_or_stream):
    if isinstance(path_or_stream, str):
        if path_or_stream.endswith(".gz"):
            import gzip
            return gzip.open(path_or_stream, "rt"), True
        return open(path_or_stream), True
    return path_or_stream, False


def read_gtf2(path_or_stream, cls=None):
    """Transcripts of a GTF2 file as |SegmentChains| named by ``transcript_id``, in the order
    ``GTF2_TranscriptAssembler`` returns them for an unsorted file (readers/gff.py:1150-1206)."""
    from .roitools import GenomicSegment, SegmentChain
    cls = SegmentChain if cls is None else cls
    fh, opened = _open_text(path_or_stream)
    try:
        rows = _assemble_gtf2(iter_gtf2_features(fh))
    finally:
        if opened:
            fh.close()
    chains = []
    for chrom, strand, blocks, attr in rows:
        chain = cls()
        chain._set_segments([GenomicSegment(chrom, s, e, strand) for s, e in blocks])
        chain.attr.update(attr)
        chain.attr.setdefault("ID", attr.get("transcript_id"))
        chains.append(chain)
    return chains


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

    @classmethod
    def from_gtf2(cls, path_or_stream, references):
        """Straight from GTF2 text: exon-like features grouped by ``transcript_id`` and merged
        (see :func:`_assemble_gtf2`), no per-transcript Python objects."""
        index = {r: i for i, r in enumerate(references)}
        fh, opened = _open_text(path_or_stream)
        try:
            rows = _assemble_gtf2(iter_gtf2_features(fh))
        finally:
            if opened:
                fh.close()
        tid, strand, ex_off, s, e, ids = [], [], [0], [], [], []
        for chrom, st, blocks, attr in rows:
            tid.append(index.get(chrom, -1))
            strand.append(STRAND_CODE.get(st, 3))
            for a, b in blocks:
                s.append(a)
                e.append(b)
            ex_off.append(len(s))
            ids.append(attr.get("transcript_id"))
        return cls(references, None, tid, strand, ex_off, s, e, ids=ids)

    @classmethod
    def from_gff3(cls, path_or_stream, references, **types):
        """Straight from GFF3 text (see :func:`_assemble_gff3`), no per-transcript Python objects."""
        index = {r: i for i, r in enumerate(references)}
        tid, strand, ex_off, s, e, ids = [], [], [0], [], [], []
        for chrom, st, blocks, attr in _gff3_rows(path_or_stream, **types):
            tid.append(index.get(chrom, -1))
            strand.append(STRAND_CODE.get(st, 3))
            for a, b in blocks:
                s.append(a)
                e.append(b)
            ex_off.append(len(s))
            ids.append(attr["transcript_id"])
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

    # -------------------------------------------------------------- positions
    def position_arrays(self, engine=None):
        """``SegmentChain.get_position_list`` for EVERY chain at once (roitools.pyx:1450-1484, 2059-2080):
        ``(positions, chain_off)`` -- the genomic coordinate of every chain position, ascending within a
        chain regardless of strand, chains back to back; chain ``c`` owns
        ``positions[chain_off[c]:chain_off[c+1]]``.  With an :class:`~plastid_amd.engine.Engine` the array
        is filled by one HIP kernel over the plan's segments (``pc_plan_coordinates``); without one, by a
        single vectorised numpy pass.  Both give the same array."""
        chain_off = np.zeros(self.n + 1, np.int64)
        np.cumsum(self.length, out=chain_off[1:])
        if engine is not None:
            p = self.plan_arrays(rows=1, stranded=False)        # ascending layout: the position hash
            plan = engine.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"],
                               p["out_elems"], 1)
            try:
                return plan.coordinates(), chain_off
            finally:
                plan.close()
        ex_len = self.ex_end - self.ex_start
        total = int(self.ex_cum[-1])
        pos = np.repeat(self.ex_start - self.ex_cum[:-1], ex_len) + np.arange(total, dtype=np.int64)
        return pos, chain_off

    def masked_position_arrays(self, masks, engine=None):
        """``SegmentChain.get_masked_position_set`` for every chain (roitools.pyx:2117-2135, masks as
        ``add_masks`` defines them, :2213-2256): the chain positions NOT covered by a mask segment of the
        same chain.  `masks` is an :class:`IntervalTable` with one (possibly empty) chain of mask segments
        per chain of `self` -- mask segments may overlap each other and need not lie inside the chain.
        Returns ``(positions, chain_off)`` like :meth:`position_arrays`."""
        if masks.n != self.n:
            raise ValueError("mask table has %d chains, expected %d" % (masks.n, self.n))
        pos, off = self.position_arrays(engine)
        chain_of = np.repeat(np.arange(self.n, dtype=np.int64), self.length)
        keep = np.ones(len(pos), bool)
        if masks.n_segments:
            # a position is masked iff it lies in [start, end) of some mask segment of its chain:
            # count starts <= x minus ends <= x over the chain's (sorted) mask edges
            shift = np.int64(1) << 40
            mchain = masks.ex_tx.astype(np.int64)
            ks = np.sort(mchain * shift + masks.ex_start)
            ke = np.sort(mchain * shift + masks.ex_end)
            kx = chain_of * shift + pos
            opened = np.searchsorted(ks, kx, side="right") - np.searchsorted(ks, chain_of * shift, side="left")
            closed = np.searchsorted(ke, kx, side="right") - np.searchsorted(ke, chain_of * shift, side="left")
            keep = opened == closed
        moff = np.zeros(self.n + 1, np.int64)
        np.cumsum(np.bincount(chain_of[keep], minlength=self.n), out=moff[1:])
        return pos[keep], moff

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
        flat, off = self.position_arrays()
        for t in range(self.n if limit is None else min(limit, self.n)):
            s = STRAND_CHAR[int(self.strand[t])]
            chrom = self.references[self.tid[t]] if self.tid[t] >= 0 else "?"
            segs = [GenomicSegment(chrom, int(self.ex_start[j]), int(self.ex_end[j]), s)
                    for j in range(self.ex_off[t], self.ex_off[t + 1])]
            c = SegmentChain()
            c._set_segments(segs)
            c._position_hash = flat[off[t]:off[t + 1]]    # get_position_list / get_position_set: views of the batch array
            if self.ids is not None:
                c.attr["ID"] = self.ids[t]
            out.append(c)
        return out
