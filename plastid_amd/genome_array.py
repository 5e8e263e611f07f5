"""|BAMGenomeArray| -- the drop-in boundary of the counting path.

Host-side mirror of ``plastid/genomics/genome_array.py:582-1111`` (same method
names, argument meaning and error behaviour).  The per-segment work of the
reference's ``get_reads_and_counts`` (:760-832) -- fetch, strand filter, read
filters, ``map_fn``, normalisation, strand flip -- runs on the GPU through the C
ABI in ``include/plastid_counts.h``:

* alignment files are packed once (:mod:`plastid_amd.packing`) and staged to HBM;
* ``get(segment)`` / ``__getitem__`` count one interval;
* ``get(chain)`` / ``chain.get_counts(ga)`` count a whole |SegmentChain| in one
  launch; :meth:`BAMGenomeArray.get_counts_batch` counts thousands of chains in
  one launch (what the counting scripts of the reference do in a Python loop,
  e.g. ``bin/counts_in_region.py:113-124``).

Arbitrary Python callables keep working exactly as in the reference: a custom
*mapping function* is simply called (``map_fn(list(reads), roi)``,
genome_array.py:823) on the reads fetched from the packed records; a custom
*filter* is evaluated once per read on the host and staged as an exclusion bit so
the HIP kernels honour it.  Only the five built-in factories and
``SizeFilterFactory`` are computed on the GPU -- and they are *always* computed
there (no CPU fallback).
"""
import itertools
import os
import warnings
from collections import OrderedDict

import numpy as np

from . import _lib
from .engine import STRAND_CODE, Engine, chain_layout
from .exceptions import DataWarning, warn
from .map_factories import CenterMapFactory, FlagFilterFactory, SizeFilterFactory, _EngineMapFactory
from .packing import FLAG_EXCLUDED, PackedAlignments
from .roitools import GenomicSegment, SegmentChain


#: files of at least this many bytes are decoded on the GPU when ``decode="auto"`` (BGZF inflate + record decode as
#: HIP kernels, bam.read_bam_gpu): below it the host reader's latency wins
GPU_DECODE_MIN_BYTES = 32 << 20


def _open_alignment_source(src, regions=None, engine=None, decode="auto"):
    """Filenames are read with the package's own BAM readers; objects are used as
    given (``multiopen`` passes non-str objects through, util/io/openers.py:90-94).
    `regions`: stage only the alignments that overlap them (through the BAI index).
    `decode`: ``"host"`` (threads + zlib / libdeflate), ``"gpu"`` (the file image goes to HBM, pc_bam_open) or
    ``"auto"`` (the GPU for whole files of GPU_DECODE_MIN_BYTES and more); both give the same arrays."""
    if isinstance(src, PackedAlignments):
        return src
    if isinstance(src, str):
        from .bam import read_bam, read_bam_gpu
        if decode not in ("auto", "host", "gpu"):
            raise ValueError("decode must be 'auto', 'host' or 'gpu', got %r" % (decode,))
        on_gpu = decode == "gpu" or (decode == "auto" and regions is None and os.path.exists(src) and os.path.getsize(src) >= GPU_DECODE_MIN_BYTES)
        if on_gpu and engine is not None:
            if decode == "gpu":   # (also with `regions`: the members the index points to are inflated on the GPU)
                aln = read_bam_gpu(src, engine, regions=regions)
                aln.decoder = "gpu"
                return aln
            # "auto": a file the device decoder rejects gets a second opinion from the host decoder (whose verdict -- the
            # arrays or the exception -- is the one the caller sees); ``decoder`` records which of the two read the file
            try:
                aln = read_bam_gpu(src, engine)
                aln.decoder = "gpu"
                return aln
            except (ValueError, IOError, OSError, RuntimeError):
                pass
        aln = read_bam(src, regions=regions)
        aln.decoder = "host"
        return aln
    return src


class _DeviceAlignments(object):
    """Stand-in for a BAM file whose records live in HBM only (``BAMGenomeArray(path, keep_reads=False)``): it knows
    the header, the record and mapped-read counts; read OBJECTS are made on demand from the staged records
    (``pc_read_records``: position, aligned runs, strand, FLAG, MAPQ -- what the mapping functions and filters look at)."""

    def __init__(self, path, references, lengths):
        self.filename = path
        self.references = list(references)
        self.lengths = list(lengths)
        self.mapped = 0
        self.n = 0
        self._engine = None
        self._file_index = 0

    def close(self):
        pass

    def reads(self, indices):
        """:class:`~plastid_amd.packing.PackedRead` objects for the staged records `indices` (one gather on the GPU)."""
        from .packing import PackedRead
        idx = np.asarray(indices, np.int64)
        if not len(idx):
            return []
        d = self._engine.read_records(self._file_index, idx)
        out = []
        off, rs, rl = d["run_off"], d["run_start"], d["run_len"]
        for k in range(len(idx)):
            runs = [(int(rs[j]), int(rl[j])) for j in range(int(off[k]), int(off[k + 1]))]
            out.append(PackedRead(self, int(idx[k]), int(d["tid"][k]), int(d["pos"][k]), bool(d["reverse"][k]), runs,
                                  int(d["flag16"][k]), int(d["mapq"][k]), None))
        return out

    def read(self, i):
        return self.reads([int(i)])[0]

    def _host_only(self, *args, **kwargs):
        raise NotImplementedError(
            "this BAMGenomeArray was opened with keep_reads=False: its reads live on the GPU only (count vectors, read "
            "objects through get_reads / get_reads_batch, size and FLAG / MAPQ filters); open it with keep_reads=True "
            "for arbitrary filter functions and fetch()")

    fetch = fetch_indices = _host_only


def _pack_source(src, chroms, chrom_index):
    """Pack any alignment source into a :class:`PackedAlignments` whose ``tid``
    indexes `chroms` (the array's sorted chromosome list)."""
    if isinstance(src, PackedAlignments):
        remap = np.array([chrom_index[r] for r in src.references], np.int32)
        if len(remap) == len(chroms) and np.array_equal(remap, np.arange(len(chroms))):
            return src
        tid = remap[src.tid] if src.n else src.tid
        if src.n and np.any(np.diff(tid) < 0):
            order = np.argsort(tid, kind="stable")  # keep BAM order within a contig
            sub = src.subset(order, validate=False)   # (the source's own contig order is not this array's)
            tid = tid[order]
        else:
            sub = src
        return PackedAlignments(tid, sub.pos, sub.alen, sub.flags, sub.nblk, sub.blk_start, sub.blk_len,
                                references=chroms, lengths=[0] * len(chroms), mapped=src.mapped,
                                read_objects=sub._read_objects, wide_idx=sub.wide_idx, wide_alen=sub.wide_alen,
                                wide_nblk=sub.wide_nblk, flag16=sub.flag16, mapq=sub.mapq, qlen=sub.qlen, nh=sub.nh)
    # duck-typed pysam.AlignmentFile: walk every contig in coordinate order
    reads, tids = [], []
    for ref, length in zip(src.references, src.lengths):
        for r in src.fetch(reference=ref, start=0, end=max(int(length), 1) + 2**29):
            reads.append(r)
            tids.append(chrom_index[ref])
    order = np.argsort(np.asarray(tids, np.int64), kind="stable") if reads else []
    reads = [reads[i] for i in order]
    tids = [tids[i] for i in order]
    return PackedAlignments.from_reads(reads, tids=tids, references=chroms, lengths=[0] * len(chroms),
                                       mapped=getattr(src, "mapped", len(reads)))


class BAMGenomeArray(object):
    """BAMGenomeArray(*bamfiles, mapping=CenterMapFactory())

    A GenomeArray for read alignments in BAM files (genome_array.py:582-1111).
    `bamfiles`: filenames, :class:`~plastid_amd.packing.PackedAlignments`, or any
    object with ``fetch/references/lengths/mapped/close``; or one list of those.

    Extra keyword: ``device`` (GPU index, default 0).
    """

    def __init__(self, *bamfiles, **kwargs):
        if len(bamfiles) == 1 and isinstance(bamfiles[0], list):  # :657-658
            bamfiles = bamfiles[0]
        # (extension) regions=[(chrom, start, end) | GenomicSegment, ...]: files named by path are staged
        # only where they overlap the regions, via their BAI index -- for a few loci of a large file
        # (the engine exists before the files are opened: large BAM files are inflated and decoded on its GPU)
        self._engine = Engine(kwargs.get("device", 0))
        # (extension) keep_reads=False: files named by path go from their bytes to staged alignments entirely on the GPU
        # (Engine.add_bam: inflate, record decode and staging as kernels) -- the fastest way to count vectors; the reads
        # themselves (reads_out as objects, arbitrary filter callables) then stay out of reach
        self._device_only = kwargs.get("keep_reads", True) is False
        if self._device_only:
            from .bam import bam_header
            if not bamfiles or not all(isinstance(x, str) for x in bamfiles):
                raise ValueError("keep_reads=False takes BAM files named by path")
            self.bamfiles = [_DeviceAlignments(x, *bam_header(x)) for x in bamfiles]
            if any(b.references != self.bamfiles[0].references for b in self.bamfiles):
                raise ValueError("keep_reads=False needs the same reference list in every file")
        else:
            self.bamfiles = [_open_alignment_source(x, kwargs.get("regions"), self._engine, kwargs.get("decode", "auto")) for x in bamfiles]
        self._strands = ("+", "-", ".")
        self._normalize = False
        self._sum = None
        self._chr_lengths = {}
        for bamfile in self.bamfiles:  # :667-670
            for k, v in zip(bamfile.references, bamfile.lengths):
                self._chr_lengths[k] = max(self._chr_lengths.get(k, 0), v)
        self._chroms = sorted(list(self._chr_lengths.keys()))  # :672
        # engine contig order: the files' own order when they all agree (no re-sorting of
        # records), else the sorted union
        ref_lists = [tuple(b.references) for b in self.bamfiles]
        if ref_lists and all(r == ref_lists[0] for r in ref_lists):
            self._tid_names = list(ref_lists[0])
        else:
            self._tid_names = list(self._chroms)
        self._chrom_index = {c: i for i, c in enumerate(self._tid_names)}
        self._filters = OrderedDict()
        if self._device_only:
            self._packed = self.bamfiles
            self._engine.clear_alignments()
            for fi, b in enumerate(self.bamfiles):
                if kwargs.get("regions") is not None:   # only what overlaps the regions is staged; `mapped` is the index's whole-file count, as pysam's
                    from .bam import resolve_regions
                    kept = self._engine.add_bam(b.filename, regions=kwargs["regions"])
                    whole = resolve_regions(b.filename, [])["mapped"]
                    b.mapped = whole if whole >= 0 else kept
                else:
                    b.mapped = self._engine.add_bam(b.filename)
                b.n = self._engine.num_records(fi)
                b._engine, b._file_index = self._engine, fi
            self._base_flags = None
        else:
            self._packed = [_pack_source(x, self._tid_names, self._chrom_index) for x in self.bamfiles]
            self._engine.set_alignments(self._packed, ntid=max(len(self._tid_names), 1))
            self._base_flags = [p.flags.copy() for p in self._packed]
        self._filters_dirty = False
        self._file_offsets = np.cumsum([0] + [p.n for p in self._packed])
        self.map_fn = None
        self.set_mapping(kwargs.get("mapping", CenterMapFactory()))  # :663

    def __del__(self):  # :677-679
        for bamfile in getattr(self, "bamfiles", []):
            try:
                bamfile.close()
            except Exception:
                pass

    # ------------------------------------------------------------ bookkeeping
    def reset_sum(self):
        """Sum = total mapped reads of all files; filters are not applied (:681-690)."""
        self._sum = sum([X.mapped for X in self.bamfiles])

    def _update(self):
        self.reset_sum()

    def set_sum(self, val):
        self._sum = val

    def sum(self):
        if self._sum is None:
            self.reset_sum()
        return self._sum

    def set_normalize(self, value=True):
        assert value in (True, False)
        self._normalize = value

    def chroms(self):
        return self._chroms

    def strands(self):
        return self._strands

    def lengths(self):
        return self._chr_lengths

    def __contains__(self, chrom):
        return chrom in self._chr_lengths

    def __len__(self):
        return len(self._strands) * sum(self.lengths().values())

    def __repr__(self):
        return "<%s len=%s sum=%s chroms=%s strands=%s>" % (
            self.__class__.__name__, len(self), self.sum(), ",".join(self.chroms()), ",".join(self.strands()))

    __str__ = __repr__

    def add_filter(self, name, func):
        """Filter reads before mapping and counting (:697-722)."""
        self._filters[name] = func
        self._filters_dirty = True

    def remove_filter(self, name):
        retval = self._filters.pop(name)
        self._filters_dirty = True
        return retval

    def get_mapping(self):
        return self.map_fn.__doc__

    def set_mapping(self, mapping_function):
        """Change the mapping rule (:935-963)."""
        self.map_fn = mapping_function
        self._update()

    # ------------------------------------------------------- engine plumbing
    def _native(self):
        return isinstance(self.map_fn, _EngineMapFactory)

    def _sync_engine(self, queries=None):
        """Push mapping rule, filters and normalisation to the engine.

        `queries`: the ``(chrom, start, end, strand)`` regions about to be counted.  Arbitrary filter
        callables are evaluated lazily, as the reference does (genome_array.py:800-820): only on the
        reads ``fetch`` returns for those regions, only on the strand the region keeps, each read at
        most once (verdicts are cached until the filter set changes); ``None`` evaluates every read
        of every file (whole-annotation batch calls)."""
        self.map_fn._configure(self._engine)
        if self._filters_dirty or not hasattr(self, "_size_filter_state"):
            size = None
            custom = []
            # filters on FLAG / MAPQ run on the GPU when every file carries the two columns (files read by the
            # package's own BAM decoders do; device-only files always): combined, they are one mask test
            sam_ok = self._device_only or all(p.flag16 is not None and p.mapq is not None for p in self._packed)
            nh_ok = self._device_only or all(getattr(p, "nh", None) is not None for p in self._packed)
            flagf = None
            max_nh = 0
            for f in self._filters.values():
                if isinstance(f, SizeFilterFactory) and size is None:
                    size = f
                elif isinstance(f, FlagFilterFactory) and sam_ok and (nh_ok or not f.max_nh) and \
                        not (flagf is not None and ((flagf[0] | f.require) & (flagf[1] | f.exclude))):
                    flagf = (f.require, f.exclude, f.min_mapq) if flagf is None else \
                        (flagf[0] | f.require, flagf[1] | f.exclude, max(flagf[2], f.min_mapq))
                    if f.max_nh:   # several NH limits: the tightest
                        max_nh = f.max_nh if not max_nh else min(max_nh, f.max_nh)
                else:
                    custom.append(f)
            self._flag_filter_state = flagf
            self._nh_filter_state = max_nh
            if custom and self._device_only:
                self.bamfiles[0]._host_only()   # (arbitrary callables are evaluated on read objects the host does not have)
            self._custom_filters = custom
            if not self._device_only:
                # verdicts of the callables per record: -1 not asked yet, 0 dropped, 1 kept
                self._verdict = [np.full(packed.n, -1, np.int8) for packed in self._packed]
                for fi, packed in enumerate(self._packed):
                    if not np.array_equal(self._base_flags[fi], packed.flags):
                        packed.flags = self._base_flags[fi].copy()
                        self._engine.update_flags(fi, packed.flags)
            self._size_filter_state = size
            self._filters_dirty = False
        if self._custom_filters:
            self._evaluate_custom_filters(queries)
        if self._flag_filter_state is None:
            self._engine.set_flag_filter(enabled=False)
        else:
            self._engine.set_flag_filter(*self._flag_filter_state)
        self._engine.set_nh_filter(getattr(self, "_nh_filter_state", 0))
        size = self._size_filter_state
        if size is None:
            self._engine.set_size_filter(None)
        else:
            self._engine.set_size_filter(size.min_, size.max_)
        if self._normalize is True:
            self._engine.set_normalize(True, float(self.sum()))  # count / float(sum) * 1e6, :826-827
        else:
            self._engine.set_normalize(False)

    def _evaluate_custom_filters(self, queries):
        """Arbitrary callables (and any size filter beyond the first): evaluated per read on the host,
        staged as exclusion bits the kernels honour."""
        for fi, packed in enumerate(self._packed):
            if not packed.n:
                continue
            verdict = self._verdict[fi]
            if queries is None:
                todo = np.nonzero(verdict < 0)[0]
            else:
                sel = np.zeros(packed.n, bool)
                rev = (self._base_flags[fi] & 1) != 0
                for chrom, start, end, strand in queries:
                    if chrom not in self._chrom_index or chrom not in packed.references:
                        continue
                    idx = packed.fetch_indices(chrom, start, end)
                    if strand == "+":
                        idx = idx[~rev[idx]]
                    elif strand == "-":
                        idx = idx[rev[idx]]
                    sel[idx] = True
                todo = np.nonzero(sel & (verdict < 0))[0]
            if not len(todo):
                continue
            dropped = False
            for i in todo:
                read = packed.read(int(i))
                keep = True
                for f in self._custom_filters:
                    if not f(read):
                        keep = False
                        break
                verdict[i] = 1 if keep else 0
                dropped |= not keep
            if dropped:
                flags = self._base_flags[fi].copy()
                flags[verdict == 0] |= FLAG_EXCLUDED
                packed.flags = flags
                self._engine.update_flags(fi, flags)

    def _out_dtype(self):
        if self._normalize is True or self.map_fn._kind == _lib.MAP_CENTER:
            return np.float64
        return np.int64

    def _warn_if_unmappable(self, plan):
        if self.map_fn._kind == _lib.MAP_STRAT5:
            return
        if self.map_fn._kind == _lib.MAP_VAR5:
            # the reference names the length of the last offending read of the (last warning) call (:633-648)
            flags, lens = plan.warn_details()
            hit = np.nonzero(flags)[0]
            if len(hit):
                warn(self.map_fn._warn_message(int(lens[hit[-1]])), DataWarning, stacklevel=4)
            return
        if plan.warn_flags().any():
            msg = self.map_fn._warn_message()
            warn(msg, DataWarning, stacklevel=4)

    # --------------------------------------------------------------- queries
    def _fetch_filtered(self, roi):
        """Host-side read objects for `roi`: fetch, strand filter, filters (:800-820)."""
        reads = itertools.chain.from_iterable(
            (X.fetch(reference=roi.chrom, start=roi.start, end=roi.end) for X in self._packed_sources()))
        if roi.strand == "+":
            reads = filter(lambda x: x.is_reverse is False, reads)
        elif roi.strand == "-":
            reads = filter(lambda x: x.is_reverse is True, reads)
        for my_filter in self._filters.values():
            reads = filter(my_filter, reads)
        return list(reads)

    def _packed_sources(self):
        return self._packed

    def get_reads_and_counts(self, roi, roi_order=True):
        """Reads covering a |GenomicSegment| and the count vector under the current
        mapping rule (genome_array.py:760-832)."""
        chrom, strand = roi.chrom, roi.strand
        if chrom not in self._chr_lengths:  # :795-798
            shape = [1] + getattr(self.map_fn, "shape", [])
            return [], np.zeros(shape)

        if not self._native():
            # plugin mapping function: same call as the reference (:823)
            reads, count_array = self.map_fn(self._fetch_filtered(roi), roi)
            if self._normalize is True:
                count_array = count_array / float(self.sum()) * 1e6
            if roi_order is True and strand == "-":
                count_array = count_array[..., ::-1]
            return reads, count_array

        count_array, plan = self._count_segments([roi], roi_order=roi_order, keep_plan=True)
        tid = self._chrom_index[chrom]
        code = roi.c_strand
        reads = []
        if self._device_only:   # the reads the rule kept, file by file, found and gathered on the GPU
            reads = self.get_reads_batch([roi])[0]
        for fi, packed in enumerate(self._packed if not self._device_only else []):
            idx = packed.fetch_indices(chrom, roi.start, roi.end)
            if len(idx) == 0:
                continue
            lo, hi = int(idx[0]), int(idx[-1]) + 1
            mask = self._engine.mapped_reads(fi, lo, hi, tid, roi.start, roi.end, code)
            reads.extend(packed.read(i) for i in (lo + np.nonzero(mask)[0]))
        self._warn_if_unmappable(plan)
        plan.close()
        rows = self._engine.rows
        if self.map_fn._kind == _lib.MAP_STRAT5:
            count_array = count_array.reshape(rows, len(roi))
        return reads, count_array

    def get_reads(self, roi):
        reads, _ = self.get_reads_and_counts(roi)
        return reads

    def get_reads_batch(self, segments, as_indices=False):
        """``[self.get_reads(seg) for seg in segments]`` in ONE pass over the staged alignments
        (``pc_mapped_reads_batch``): what ``bin/psite.py:182`` / ``bin/phase_by_size.py:187`` ask region by
        region.  `as_indices`: lists of ``(file index, record index)`` arrays instead of read objects (no Python
        object per read).  Segments on unknown chromosomes give empty lists (genome_array.py:795-798)."""
        segs = list(segments)
        if not self._native():
            return [self.get_reads(s) for s in segs]
        self._sync_engine([(s.chrom, s.start, s.end, s.strand) for s in segs])
        n = len(segs)
        tid = [self._chrom_index.get(s.chrom, -1) for s in segs]
        plan = self._engine.plan(tid, [s.start for s in segs], [s.end for s in segs], [s.c_strand for s in segs],
                                 np.zeros(n, np.int64), np.zeros(n, np.int8), np.ones(n, np.int64), max(self._engine.rows, 1), self._engine.rows)
        try:
            offsets, rec = plan.mapped_reads()
        finally:
            plan.close()
        nfiles = len(self._packed)
        out = []
        for s in range(n):
            parts = [(f, rec[offsets[s * nfiles + f]:offsets[s * nfiles + f + 1]]) for f in range(nfiles)]
            if as_indices:
                out.append([(f, idx.astype(np.int64)) for f, idx in parts if len(idx)])
            elif self._device_only:
                out.append([r for f, idx in parts if len(idx) for r in self._packed[f].reads(idx)])
            else:
                out.append([self._packed[f].read(int(i)) for f, idx in parts for i in idx])
        return out

    def __getitem__(self, roi):
        return self.get(roi, roi_order=True)

    def get(self, roi, roi_order=True):
        """Count vector over a |GenomicSegment| or |SegmentChain| (:891-928)."""
        if isinstance(roi, SegmentChain):
            return roi.get_counts(self)
        if not self._native() or roi.chrom not in self._chr_lengths:
            _, count_array = self.get_reads_and_counts(roi, roi_order=roi_order)
            return count_array
        if self._one_call_query(roi):
            # one segment, one call: the window travels in the kernel's arguments, the counts come back through page-locked
            # memory (pc_query_segment) -- for arrays whose reads cannot make the map function warn (checked once per rule)
            self._sync_engine([(roi.chrom, roi.start, roi.end, roi.strand)])
            return self._engine.query_segment(self._chrom_index[roi.chrom], roi.start, roi.end, roi.c_strand,
                                              roi_order is True and roi.strand == "-", self._out_dtype())
        count_array, plan = self._count_segments([roi], roi_order=roi_order, keep_plan=True)
        self._warn_if_unmappable(plan)
        plan.close()
        if self.map_fn._kind == _lib.MAP_STRAT5:
            count_array = count_array.reshape(self._engine.rows, len(roi))
        return count_array

    #: segments up to this many positions are counted by ``pc_query_segment``
    ONE_CALL_MAX = 4096

    def _one_call_query(self, roi):
        if (self._device_only or len(self._packed) != 1 or not 0 < len(roi) <= self.ONE_CALL_MAX or roi.start < 0 or
                self.map_fn._kind not in (_lib.MAP_FIVE, _lib.MAP_THREE, _lib.MAP_VAR5) or os.environ.get("PC_NO_SINGLE")):
            return False
        args = self.map_fn._engine_args()
        # the verdict depends on the rule (an offset table can be mutated in place: its bytes are part of the key) and on
        # the reads staged -- the key holds the objects themselves, so a freed map function's id cannot come back as another
        fw = args.get("fw")
        key = (self.map_fn, args.get("kind"), args.get("param"), None if fw is None else np.asarray(fw).tobytes(),
               tuple(self._packed))
        prev = getattr(self, "_no_warn_key", None)
        if prev is None or len(prev) != len(key) or prev[0] is not key[0] or prev[1:4] != key[1:4] or \
                len(prev[4]) != len(key[4]) or any(a is not b for a, b in zip(prev[4], key[4])):
            # can any read of the file make this rule emit its DataWarning (an offset beyond the read, a length without an
            # offset)?  One pass over the aligned lengths, once per rule; if so, queries keep to the plan path, which reports it
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                self._no_warn = self.map_fn._direct_warning(self._packed) is None
            self._no_warn_key = key
        return self._no_warn

    def _count_segments(self, segs, roi_order, keep_plan=False):
        """One launch over independent segments, each laid out like ``get(seg, roi_order)``."""
        self._sync_engine([(s.chrom, s.start, s.end, s.strand) for s in segs])
        rows = self._engine.rows
        tid = [self._chrom_index.get(s.chrom, -1) for s in segs]
        start = [s.start for s in segs]
        end = [s.end for s in segs]
        strand = [s.c_strand for s in segs]
        out_off, out_step, row_stride = [], [], []
        base = 0
        for s in segs:
            n = len(s)
            if roi_order is True and s.strand == "-":  # :829-830
                out_off.append(base + n - 1)
                out_step.append(-1)
            else:
                out_off.append(base)
                out_step.append(1)
            row_stride.append(n)
            base += rows * n
        plan = self._engine.plan(tid, start, end, strand, out_off, out_step, row_stride, base, rows)
        out = plan.count(self._out_dtype())
        if keep_plan:
            return out, plan
        plan.close()
        return out

    def _get_chain_counts(self, chain, stranded=True):
        """``SegmentChain.get_counts`` for this array: one launch for the whole chain
        (roitools.pyx:3259-3271 semantics, float64 result)."""
        if not self._native():
            return _generic_chain_counts(self, chain, stranded)
        if chain.chrom not in self._chr_lengths:
            # unknown chromosome: every segment yields zeros([1]+shape), which broadcasts
            # into a 1-D chain vector and cannot into a stratified one (Q9)
            return _generic_chain_counts(self, chain, stranded)
        out = self.get_counts_batch([chain], stranded=stranded)
        return out[0]

    def get_counts_batch(self, chains, stranded=True):
        """Count many |SegmentChains| in ONE launch.  Returns a list of float64
        arrays, each what ``chain.get_counts(self, stranded)`` returns."""
        if not self._native():
            return [c.get_counts(self, stranded) for c in chains]
        self._sync_engine([(c.chrom, s.start, s.end, c.strand) for c in chains for s in c])
        rows = self._engine.rows
        segs, strands = [], []
        for c in chains:
            segs.append([(s.start, s.end) for s in c])
            strands.append(c.strand)
        seg_chain, out_off, out_step, row_stride, chain_base, chain_len, total = chain_layout(
            segs, strands, rows=rows, stranded=stranded is True)
        tid = np.array([self._chrom_index.get(chains[ci].chrom, -1) for ci in seg_chain], np.int32)
        flat = [se for chain_segs in segs for se in chain_segs]
        start = np.array([s for s, _ in flat], np.int64)
        end = np.array([e for _, e in flat], np.int64)
        strand = np.array([chains[ci].c_strand for ci in seg_chain], np.uint8)
        plan = self._engine.plan(tid, start, end, strand, out_off, out_step, row_stride, total, rows)
        out = plan.count(np.float64)  # SegmentChain.get_counts always returns float (roitools.pyx:3262)
        self._warn_if_unmappable(plan)
        plan.close()
        results = []
        strat = self.map_fn._kind == _lib.MAP_STRAT5
        for ci, c in enumerate(chains):
            if len(c) == 0:
                results.append(c.get_counts(self, stranded))
                continue
            block = out[chain_base[ci]:chain_base[ci] + rows * chain_len[ci]]
            results.append(block.reshape(rows, chain_len[ci]) if strat else block)
        return results

    def count_table(self, table, stranded=True, dtype=np.float64):
        """Count every chain of an :class:`~plastid_amd.annotation.IntervalTable` in ONE launch.
        Returns ``(flat, per_chain)``: the flat result buffer and per-chain views, each equal to
        ``chain.get_counts(self, stranded)`` (as `dtype`; float64 is what the reference returns)."""
        if not self._native():
            raise TypeError("count_table needs one of the built-in mapping factories")
        self._sync_engine()
        rows = self._engine.rows
        if list(table.references) != list(self._tid_names):
            raise ValueError("IntervalTable was built against a different reference list")
        p = table.plan_arrays(rows=rows, stranded=stranded)
        plan = self._engine.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"],
                                 p["row_stride"], p["out_elems"], rows)
        want = np.float64 if (self._out_dtype() == np.float64) else np.dtype(dtype)
        flat = plan.count(want)
        self._warn_if_unmappable(plan)
        plan.close()
        return flat, table.split_counts(flat, rows if self.map_fn._kind == _lib.MAP_STRAT5 else 1)

    def count_in_regions(self, chains):
        """Fused region statistics: for every |SegmentChain| the masked count sum, masked length,
        reads per nucleotide and RPKM -- what ``bin/counts_in_region.py:113-124`` computes with
        ``numpy.nansum(chain.get_masked_counts(ga))`` per chain -- in ONE launch; only the sums come
        back from the GPU, no per-position vectors.  Integer mapping rules only.

        Returns a dict of arrays: ``counts`` (int64 ``[n]`` or ``[n, rows]``), ``length``,
        ``counts_per_nucleotide``, ``rpkm`` (float64; NaN where the masked length is 0)."""
        if not self._native() or self.map_fn._kind == _lib.MAP_CENTER:
            raise TypeError("count_in_regions needs an integer-valued built-in mapping factory")
        norm = self._normalize
        self._normalize = False
        try:
            self._sync_engine([(c.chrom, s.start, s.end, c.strand) for c in chains for s in list(c) + list(c.mask_segments)])
        finally:
            self._normalize = norm
        rows = self._engine.rows
        n = len(chains)
        tid, start, end, strand, out_off = [], [], [], [], []
        for ci, c in enumerate(chains):
            t = self._chrom_index.get(c.chrom, -1) if len(c) else -1
            for seg in c:                       # the chain itself -> slot ci
                tid.append(t); start.append(seg.start); end.append(seg.end); strand.append(c.c_strand)
                out_off.append(ci * rows)
            for seg in c.mask_segments:         # its masked positions -> slot n + ci (subtracted below)
                tid.append(t); start.append(seg.start); end.append(seg.end); strand.append(c.c_strand)
                out_off.append((n + ci) * rows)
        nseg = len(tid)
        plan = self._engine.plan(tid, start, end, strand, out_off, np.zeros(nseg, np.int8), np.ones(nseg, np.int64),
                                 2 * n * rows, rows)
        sums = plan.count(np.int64).reshape(2, n, rows)
        self._warn_if_unmappable(plan)
        plan.close()
        counts = sums[0] - sums[1]
        length = np.array([c.masked_length for c in chains], np.int64)
        with np.errstate(divide="ignore", invalid="ignore"):
            denom = np.where(length == 0, np.nan, length).astype(np.float64)
            rpnt = counts.astype(np.float64) / denom[:, None]
            rpkm = rpnt * (1000.0 * 1e6 / self.sum())
        if rows == 1 and self.map_fn._kind != _lib.MAP_STRAT5:
            counts, rpnt, rpkm = counts[:, 0], rpnt[:, 0], rpkm[:, 0]
        return {"counts": counts, "length": length, "counts_per_nucleotide": rpnt, "rpkm": rpkm}

    @staticmethod
    def counts_in_region_lines(chains, stats, names=None):
        """The per-region lines ``bin/counts_in_region.py:113-124`` writes, from the result of
        :meth:`count_in_regions` (one row per chain): name, region, counts, counts per nucleotide,
        RPKM (``%.8e``) and masked length.  A fully masked region prints ``nan`` for all three numbers:
        ``numpy.nansum`` of the reference's all-masked array is ``masked``, which formats as nan."""
        lines = []
        for i, c in enumerate(chains):
            length = int(stats["length"][i])
            counts = float("nan") if length == 0 else float(stats["counts"][i])
            name = names[i] if names is not None else c.get_name()
            lines.append("\t".join([name, str(c), "%.8e" % counts, "%.8e" % float(stats["counts_per_nucleotide"][i]),
                                    "%.8e" % float(stats["rpkm"][i]), "%d" % length]))
        return lines

    def to_genome_array(self, array_type=None):
        """Dense per-chromosome arrays under the current mapping rule (genome_array.py:965-988): one
        whole-contig launch per chromosome and strand.  As in the reference the query stops one
        position short of the contig end (``lengths()[chrom] - 1`` as the half-open end)."""
        if array_type is None:
            array_type = DenseGenomeArray
        ga = array_type(chr_lengths=self.lengths(), strands=self.strands())
        for chrom in self.chroms():
            for strand in self.strands():
                seg = GenomicSegment(chrom, 0, self.lengths()[chrom] - 1, strand)
                ga[seg] = self[seg]
        return ga

    # ---------------------------------------------------------------- export
    def _chromosome_runs(self, chrom, strand, period):
        """Runs of the whole-chromosome vector (genome order) under the current mapping rule: one
        count launch + the GPU run-length encoder; only the runs come back.  Returns
        ``(starts, ends, values)``; runs are also cut at every multiple of `period`."""
        size = self.lengths()[chrom]
        self._sync_engine([(chrom, 0, size, strand)])
        rows = self._engine.rows
        if rows != 1:
            raise TypeError("export needs a mapping rule that returns one row per position")
        plan = self._engine.plan([self._chrom_index[chrom]], [0], [size], [STRAND_CODE[strand]], [0], np.ones(1, np.int8),
                                 [size], size, 1)
        plan.launch(self._out_dtype())
        starts, values = plan.rle(period)
        self._warn_if_unmappable(plan)
        plan.close()
        ends = np.append(starts[1:], size)
        return starts, ends, values

    def to_bedgraph(self, fh, trackname, strand, window_size=100000, printer=None, **kwargs):
        """Write a bedGraph under the current mapping rule (:1041-1111).  Same lines as the
        reference's window loop writes (its runs restart at every window border, windows and runs
        without counts are skipped), from one launch per chromosome."""
        assert strand in self.strands()
        assert window_size > 0
        fh.write("track type=bedGraph name=%s" % trackname)
        for k, v in sorted(kwargs.items(), key=lambda x: x[0]):
            fh.write(" %s=%s" % (k, v))
        fh.write("\n")
        for chrom in sorted(self.chroms()):
            if printer is not None:
                printer.write("Writing chromosome %s..." % chrom)
            if self.lengths()[chrom] <= 0:
                continue
            starts, ends, values = self._chromosome_runs(chrom, strand, window_size)
            keep = values > 0
            for a, b, v in zip(starts[keep], ends[keep], values[keep]):
                fh.write("%s\t%s\t%s\t%s\n" % (chrom, a, b, v))

    def to_variable_step(self, fh, trackname, strand, window_size=100000, printer=None, **kwargs):
        """Write a variableStep wiggle under the current mapping rule (:990-1039): every position
        with a non-zero count, 1-based."""
        assert strand in self.strands()
        fh.write("track type=wiggle_0 name=%s" % trackname)
        for k, v in sorted(kwargs.items(), key=lambda x: x[0]):
            fh.write(" %s=%s" % (k, v))
        fh.write("\n")
        for chrom in sorted(self.chroms()):
            if printer is not None:
                printer.write("Writing chromosome %s..." % chrom)
            fh.write("variableStep chrom=%s span=1\n" % chrom)
            if self.lengths()[chrom] <= 0:
                continue
            starts, ends, values = self._chromosome_runs(chrom, strand, 1)
            keep = values != 0
            for a, v in zip(starts[keep], values[keep]):
                fh.write("%s\t%s\n" % (a + 1, v))


class DenseGenomeArray(object):
    """Read-mostly dense array, the part of the reference's |GenomeArray| that
    ``BAMGenomeArray.to_genome_array`` produces and ``SegmentChain.get_counts`` consumes
    (genome_array.py:1323-1800, hot subset): one float vector per chromosome and strand."""

    def __init__(self, chr_lengths=None, strands=("+", "-")):
        self._chr_lengths = OrderedDict(chr_lengths or {})
        self._strands = tuple(strands)
        self._chroms = {c: {st: np.zeros(int(n)) for st in self._strands} for c, n in self._chr_lengths.items()}
        self._normalize = False
        self._sum = None

    def chroms(self):
        return list(self._chr_lengths.keys())

    def strands(self):
        return self._strands

    def lengths(self):
        return self._chr_lengths

    def sum(self):
        if self._sum is None:
            self._sum = float(sum(v.sum() for d in self._chroms.values() for v in d.values()))
        return self._sum

    def set_normalize(self, value=True):
        assert value in (True, False)
        self._normalize = value

    def __setitem__(self, seg, val):
        """``ga[seg] = val`` with `val` 5'->3' relative to `seg` (genome_array.py:1655-1700)."""
        val = np.asarray(val, float)
        if seg.strand == "-":
            val = val[::-1]
        self._chroms[seg.chrom][seg.strand][seg.start:seg.end] = val
        self._sum = None

    def get(self, roi, roi_order=True):
        if isinstance(roi, SegmentChain):
            return roi.get_counts(self)
        out = np.zeros(len(roi))
        vec = self._chroms.get(roi.chrom, {}).get(roi.strand)
        if vec is not None:
            a, b = max(roi.start, 0), min(roi.end, len(vec))
            if b > a:
                out[a - roi.start:b - roi.start] = vec[a:b]
        if self._normalize is True:
            out = out / float(self.sum()) * 1e6
        if roi_order is True and roi.strand == "-":
            out = out[::-1]
        return out

    def __getitem__(self, roi):
        return self.get(roi, roi_order=True)


def _generic_chain_counts(ga, chain, stranded):
    """The reference's per-segment loop (roitools.pyx:3259-3271) for array objects
    or situations the batched path does not cover."""
    count_arrays = [ga.get(X, roi_order=False) for X in chain]
    dims = list(count_arrays[0].shape)
    dims[-1] = chain.length
    count_array = np.empty(dims, dtype=float)
    i = 0
    for n, seg in enumerate(chain):
        j = i + len(seg)
        count_array[..., i:j] = count_arrays[n]
        i = j
    if chain.c_strand == 2 and stranded is True:
        count_array = count_array[..., ::-1]
    return count_array
