"""BAM files -> :class:`~plastid_amd.packing.PackedAlignments` with the package's own
native reader (``csrc/bam_stager.cpp``: BGZF inflate on a thread pool + BAM record parse).

Replaces what the reference gets from pysam on this path (``pysam.AlignmentFile(X, "rb")``,
``.references/.lengths/.mapped``, ``read.positions``, ``read.is_reverse`` --
plastid/genomics/genome_array.py:660-690, 800-815).  The file must be coordinate sorted
(as for pysam ``fetch``); no index file is needed because the whole file is staged.
"""
import ctypes
import os

import numpy as np

from .build import BAM_LIB, build_bam_library
from .packing import PackedAlignments

_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(BAM_LIB):
            build_bam_library()
        L = ctypes.CDLL(BAM_LIB)
        vp = ctypes.c_void_p
        L.pb_last_error.restype = ctypes.c_char_p
        L.pb_open.restype = vp
        L.pb_open.argtypes = [ctypes.c_char_p]
        L.pb_close.argtypes = [vp]
        L.pb_load.argtypes = [vp, ctypes.c_int]
        L.pb_load_regions.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), vp, vp]
        L.pb_nref.argtypes = [vp]
        L.pb_ref_name.restype = ctypes.c_char_p
        L.pb_ref_name.argtypes = [vp, ctypes.c_int]
        L.pb_ref_length.restype = ctypes.c_int32
        L.pb_ref_length.argtypes = [vp, ctypes.c_int]
        L.pb_counts.argtypes = [vp, vp]
        L.pb_fill.argtypes = [vp] * 8
        L.pb_wide_count.restype = ctypes.c_int64
        L.pb_wide_count.argtypes = [vp]
        L.pb_fill_wide.argtypes = [vp] * 4
        L.pb_fill_sam.argtypes = [vp] * 4
        L.pb_fill_nh.argtypes = [vp] * 2
        L.pb_resolve_regions.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p)] + [vp] * 9
        _lib = L
    return _lib


def bam_header(path):
    """``(references, lengths)`` of a BAM file from its header alone: the leading BGZF members are inflated (zlib) until
    the reference list is complete -- what ``pysam.AlignmentFile(path).references / .lengths`` give
    (genome_array.py:667-670)."""
    import struct
    import zlib
    buf = b""
    with open(path, "rb") as fh:
        def more():
            head = fh.read(12)
            if len(head) < 12:
                return False
            if head[0] != 31 or head[1] != 139 or head[2] != 8 or not (head[3] & 4):
                raise ValueError("not a BGZF file (bad gzip member header)")
            xlen = struct.unpack("<H", head[10:12])[0]
            extra = fh.read(xlen)
            bsize, x = -1, 0
            while x + 4 <= len(extra):
                slen = struct.unpack("<H", extra[x + 2:x + 4])[0]
                if extra[x:x + 2] == b"BC" and slen == 2:
                    bsize = struct.unpack("<H", extra[x + 4:x + 6])[0]
                x += 4 + slen
            if bsize < 0:
                raise ValueError("BGZF member without BC subfield")
            payload = fh.read(bsize + 1 - 12 - xlen)
            if len(payload) < bsize + 1 - 12 - xlen:
                raise ValueError("truncated BGZF member")
            nonlocal buf
            buf += zlib.decompressobj(-15).decompress(payload[:-8])
            return True

        def need(n):
            while len(buf) < n:
                if not more():
                    raise ValueError("truncated BAM header")

        need(12)
        if buf[:4] != b"BAM\x01":
            raise ValueError("not a BAM file (bad magic)")
        l_text = struct.unpack("<i", buf[4:8])[0]
        need(12 + l_text)
        n_ref = struct.unpack("<i", buf[8 + l_text:12 + l_text])[0]
        at = 12 + l_text
        refs, lens = [], []
        for _ in range(n_ref):
            need(at + 4)
            l_name = struct.unpack("<i", buf[at:at + 4])[0]
            need(at + 4 + l_name + 4)
            refs.append(buf[at + 4:at + 4 + l_name - 1].decode())
            lens.append(struct.unpack("<i", buf[at + 4 + l_name:at + 8 + l_name])[0])
            at += 8 + l_name
    return refs, lens


def _region_tuples(regions):
    return [(r.chrom, r.start, r.end) if hasattr(r, "chrom") else tuple(r) for r in regions]


def resolve_regions(path, regions):
    """Resolve `regions` (``(chrom, start, end)`` or |GenomicSegments|) through the BAI index of `path` for a decoder
    that reads the file itself (:func:`read_bam_gpu`, :meth:`Engine.add_bam`): returns a dict with the span of virtual
    offsets ``voff_begin``, ``voff_end`` that holds every chunk of every region (bins + 16 kb linear index, SAM
    specification section 5 -- what ``AlignmentFile.fetch`` walks per region, genome_array.py:800-809), the merged regions
    by reference id (``tid``, ``beg``, ``end`` arrays), the index's whole-file ``mapped`` count (-1: none) and the
    file's ``references`` / ``lengths``."""
    L = _load()
    h = L.pb_open(os.fsencode(path))
    if not h:
        raise IOError(L.pb_last_error().decode())
    try:
        regs = _region_tuples(regions)
        n = len(regs)
        names = (ctypes.c_char_p * max(n, 1))(*[os.fsencode(str(c)) for c, _, _ in regs])
        starts = np.array([int(s) for _, s, _ in regs], np.int64)
        ends = np.array([int(e) for _, _, e in regs], np.int64)
        vb, ve = ctypes.c_uint64(0), ctypes.c_uint64(0)
        mapped, nm = ctypes.c_int64(0), ctypes.c_int(0)
        tid, beg, end = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.int64)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        rc = L.pb_resolve_regions(h, n, names, p(starts), p(ends), ctypes.byref(vb), ctypes.byref(ve), ctypes.byref(mapped), ctypes.byref(nm),
                                  p(tid), p(beg), p(end))
        if rc != 0:
            raise ValueError(L.pb_last_error().decode())
        nref = L.pb_nref(h)
        refs = [L.pb_ref_name(h, i).decode() for i in range(nref)]
        lens = [int(L.pb_ref_length(h, i)) for i in range(nref)]
    finally:
        L.pb_close(h)
    k = int(nm.value)
    return dict(voff_begin=int(vb.value), voff_end=int(ve.value), tid=tid[:k].copy(), beg=beg[:k].copy(), end=end[:k].copy(),
                mapped=int(mapped.value), references=refs, lengths=lens)


def read_bam_gpu(path, engine, timing=None, regions=None):
    """The same :class:`PackedAlignments` as :func:`read_bam` gives for a whole file, decoded ON THE GPU: the file image
    goes to HBM as it is, the BGZF members are inflated there (one wave per member) and the BAM records decoded
    (``pc_bam_open``, ``csrc/bam_kernels.hip.h``); only the packed columns -- 13 bytes per record instead of the ~120 of
    an aligner's record -- come back.  `engine`: a :class:`plastid_amd.engine.Engine` (its device and stream are used).
    `timing`: optional dict that receives the phase times in ms and the member / byte counts.
    `regions`: as for :func:`read_bam` -- only the alignments that overlap one of them, through the BAI index: only the
    BGZF members of ONE span of the file -- from the first to the last index chunk of the regions, whatever lies between two far-apart regions included -- are uploaded and inflated (``pc_bam_open_span``; the overlap test then drops what no region wants: regions that sit together, like one rank's genome range, read little else); ``mapped`` is then the index's
    whole-file count, as pysam's."""
    import time
    from . import _lib as clib
    L = clib.load()
    if not os.path.isfile(path):
        raise IOError("No such file: %r" % (path,))
    t_0 = time.perf_counter()
    h = ctypes.c_void_p()
    span = None
    if regions is not None:
        span = resolve_regions(path, regions)
        pv = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        clib.check(L.pc_bam_open_span(engine._h, os.fsencode(path), span["voff_begin"], span["voff_end"], len(span["tid"]),
                                      pv(span["tid"]), pv(span["beg"]), pv(span["end"]), ctypes.byref(h)))
    else:
        # (the library maps the file itself: pages touched by all host threads at once, unmapped on a thread of its own)
        clib.check(L.pc_bam_open_path(engine._h, os.fsencode(path), ctypes.byref(h)))
    t_open = time.perf_counter()
    size = os.path.getsize(path)
    try:
        counts = np.zeros(8, np.int64)
        clib.check(L.pc_bam_counts(h, counts.ctypes.data_as(ctypes.c_void_p)))
        n, nrun, mapped, nw = int(counts[0]), int(counts[1]), int(counts[2]), int(counts[4])
        nref = L.pc_bam_nref(h)
        refs = [L.pc_bam_ref_name(h, i).decode() for i in range(nref)]
        lens = [int(L.pc_bam_ref_length(h, i)) for i in range(nref)]
        tid, pos = np.empty(n, np.int32), np.empty(n, np.int32)
        alen, flags, nblk = np.empty(n, np.uint16), np.empty(n, np.uint8), np.empty(n, np.uint8)
        bs, bl = np.empty(nrun, np.int32), np.empty(nrun, np.int32)
        wi, wa, wn = np.empty(nw, np.int64), np.empty(nw, np.int32), np.empty(nw, np.int32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        clib.check(L.pc_bam_read(h, p(tid), p(pos), p(alen), p(flags), p(nblk), p(bs), p(bl), p(wi), p(wa), p(wn)))
        flag16, mapq, qlen = np.empty(n, np.uint16), np.empty(n, np.uint8), np.empty(n, np.int32)
        clib.check(L.pc_bam_read_sam(h, p(flag16), p(mapq), p(qlen)))
        nh = np.empty(n, np.uint16)
        clib.check(L.pc_bam_read_nh(h, p(nh)))
        if timing is not None:
            timing.update(open_wall_ms=(t_open - t_0) * 1e3, read_wall_ms=(time.perf_counter() - t_open) * 1e3)
            ms = np.zeros(4, np.float64)
            clib.check(L.pc_bam_timing(h, p(ms)))
            timing.update(upload_ms=float(ms[0]), inflate_ms=float(ms[1]), chain_ms=float(ms[2]), decode_ms=float(ms[3]),
                          members=int(counts[5]), inflated_bytes=int(counts[6]), compressed_bytes=size, chain_restarts=int(counts[7]),
                          records=int(counts[3]))
    finally:
        t_c = time.perf_counter()
        L.pc_bam_close(h)
        if timing is not None:
            timing["close_ms"] = (time.perf_counter() - t_c) * 1e3
    wide = dict(wide_idx=wi, wide_alen=wa, wide_nblk=wn) if nw else {}
    if span is not None:   # `mapped` as pysam reports it: from the index, for the whole file
        if span["mapped"] < 0:
            import warnings
            warnings.warn("the BAI index of %s carries no mapped-read counts; using the number of alignments read" % path)
            mapped = n
        else:
            mapped = span["mapped"]
    out = PackedAlignments(tid, pos, alen, flags, nblk, bs, bl, references=refs, lengths=lens, mapped=mapped,
                           validate=False, flag16=flag16, mapq=mapq, qlen=qlen, nh=nh, **wide)   # the device decoder has checked every invariant validate() checks
    out.filename = path
    return out


def read_bam(path, threads=0, regions=None):
    """Read a coordinate-sorted BAM file into a :class:`PackedAlignments`.

    `regions`: iterable of ``(chrom, start, end)`` (0-based, half-open) or objects with those
    attributes (|GenomicSegments|): only the alignments that overlap one of them are read, through
    the file's BAI index (``path + ".bai"`` or ``.bai`` in place of ``.bam``) -- what the reference
    does region by region with ``AlignmentFile.fetch`` (genome_array.py:800-809), here for a whole
    query set at once.  Counts over positions inside the regions equal those of the whole file.

    ``mapped`` is the number of records with flag 0x4 unset (what ``pysam
    AlignmentFile.mapped`` reports from the index); unplaced reads are not staged
    (``fetch`` never returns them).  Raises ``ValueError`` for unsorted input, as pysam does."""
    L = _load()
    h = L.pb_open(os.fsencode(path))
    if not h:
        raise IOError(L.pb_last_error().decode())
    try:
        if regions is None:
            rc = L.pb_load(h, int(threads))
        else:
            regs = _region_tuples(regions)
            names = (ctypes.c_char_p * max(len(regs), 1))(*[os.fsencode(str(c)) for c, _, _ in regs])
            starts = np.array([int(s) for _, s, _ in regs], np.int64)
            ends = np.array([int(e) for _, _, e in regs], np.int64)
            rc = L.pb_load_regions(h, int(threads), len(regs), names, starts.ctypes.data_as(ctypes.c_void_p),
                                   ends.ctypes.data_as(ctypes.c_void_p))
        if rc != 0:
            msg = L.pb_last_error().decode()
            raise ValueError(msg)
        counts = np.zeros(4, np.int64)
        L.pb_counts(h, counts.ctypes.data_as(ctypes.c_void_p))
        n, nrun, mapped = int(counts[0]), int(counts[1]), int(counts[2])
        nref = L.pb_nref(h)
        refs = [L.pb_ref_name(h, i).decode() for i in range(nref)]
        lens = [int(L.pb_ref_length(h, i)) for i in range(nref)]
        tid = np.empty(n, np.int32)
        pos = np.empty(n, np.int32)
        alen = np.empty(n, np.uint16)
        flags = np.empty(n, np.uint8)
        nblk = np.empty(n, np.uint8)
        bs = np.empty(nrun, np.int32)
        bl = np.empty(nrun, np.int32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        L.pb_fill(h, p(tid), p(pos), p(alen), p(flags), p(nblk), p(bs), p(bl))
        wide = {}
        nw = int(L.pb_wide_count(h))
        if nw > 0:   # reads beyond the 16-bit / 8-bit columns: true lengths / run counts aside (packing.py)
            wi, wa, wn = np.empty(nw, np.int64), np.empty(nw, np.int32), np.empty(nw, np.int32)
            L.pb_fill_wide(h, p(wi), p(wa), p(wn))
            wide = dict(wide_idx=wi, wide_alen=wa, wide_nblk=wn)
        # the SAM FLAG word, MAPQ and l_seq of every record: what read filters may look at (genome_array.py:697-722)
        flag16, mapq, qlen = np.empty(n, np.uint16), np.empty(n, np.uint8), np.empty(n, np.int32)
        L.pb_fill_sam(h, p(flag16), p(mapq), p(qlen))
        nh = np.empty(n, np.uint16)   # ... and the NH:i tag (0: none): read.get_tag("NH") / has_tag("NH")
        L.pb_fill_nh(h, p(nh))
    finally:
        L.pb_close(h)
    if mapped < 0:   # an index without the per-reference counts samtools writes
        import warnings
        warnings.warn("the BAI index of %s carries no mapped-read counts; using the number of alignments read" % path)
        mapped = n
    out = PackedAlignments(tid, pos, alen, flags, nblk, bs, bl, references=refs, lengths=lens, mapped=mapped,
                           validate=False, flag16=flag16, mapq=mapq, qlen=qlen, nh=nh, **wide)   # the native reader has checked every invariant validate() checks
    out.filename = path
    return out
