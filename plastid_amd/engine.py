"""Thin Python wrapper over the C ABI (``include/plastid_counts.h``).

:class:`Engine` owns one GPU's staged alignments and mapping state;
:class:`Plan` is a batch of genomic intervals plus the layout of their count
vectors.  All counting happens in HIP kernels; this module only moves numpy
arrays across the ABI.
"""
import ctypes
import weakref

import numpy as np

from . import _lib
from ._lib import (MAP_CENTER, MAP_FIVE, MAP_STRAT5, MAP_THREE, MAP_VAR5, OUT_FLOAT64, OUT_INT64,
                   check)

STRAND_CODE = {"\x00": 0, "+": 1, "-": 2, ".": 3}  # plastid/genomics/c_common.pxd:1-6


def _ptr(a):
    # the address as an int (c_void_p argtypes take it): the caller holds `a` for the (synchronous) call
    return None if a is None else a.ctypes.data


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def release_cached_memory(device=0):
    """Return to the driver the device memory that closed engines left to the process for the next engine
    (``pc_release_cached_memory``) -- for callers that share the GPU with other libraries."""
    check(_lib.load().pc_release_cached_memory(int(device)), "pc_release_cached_memory")


class Engine(object):
    """One GPU's counting engine (``pc_engine``)."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        check(self._lib.pc_create(int(device), ctypes.byref(h)), "pc_create")
        self._h = h
        self.device = int(device)
        self._finalizer = weakref.finalize(self, self._lib.pc_destroy, h)
        self.rows = 1
        self.kind = None
        self.nfiles = 0
        self.ntid = 0
        self._state = {}   # last values pushed through the setters (a repeat is not sent again)

    def close(self):
        self._finalizer()

    def host_buffer(self, n, dtype=np.int64):
        """A page-locked numpy array of `n` elements (``pc_host_alloc``): what a caller that reads count vectors back
        repeatedly hands to :meth:`Plan.read` / :meth:`Plan.count` -- the read-back is then one DMA at the rate of the
        link.  The memory is released when the array (and every view of it) is gone; the engine must still be open then."""
        dtype = np.dtype(dtype)
        nbytes = max(int(n), 0) * dtype.itemsize
        p = ctypes.c_void_p()
        check(self._lib.pc_host_alloc(self._h, nbytes, ctypes.byref(p)), "pc_host_alloc")
        raw = (ctypes.c_uint8 * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(raw, dtype=np.uint8, count=nbytes).view(dtype)
        lib, h, addr = self._lib, self._h, p.value
        fin = self._finalizer

        def release():
            if fin.alive:                      # (an engine that is gone took its page-locked blocks with the process)
                lib.pc_host_free(h, ctypes.c_void_p(addr))
        weakref.finalize(raw, release)
        return arr

    # ------------------------------------------------------------ alignments
    def clear_alignments(self):
        check(self._lib.pc_clear_alignments(self._h))
        self.nfiles = 0

    def add_alignment_file(self, packed, ntid=None):
        """Stage one :class:`~plastid_amd.packing.PackedAlignments` to HBM."""
        ntid = len(packed.references) if ntid is None else int(ntid)
        # the C side trusts the sizes it is given: the array lengths are checked here, always
        for name in ("pos", "alen", "flags", "nblk"):
            if len(getattr(packed, name)) != packed.n:
                raise ValueError("alignment array '%s' has %d entries, expected %d" % (name, len(getattr(packed, name)), packed.n))
        if len(packed.blk_start) != len(packed.blk_len):
            raise ValueError("run arrays blk_start / blk_len differ in length")
        if getattr(packed, "n_wide", 0):   # reads beyond the 16-bit / 8-bit fields: their true lengths / run counts aside
            if len(packed.wide_alen) != packed.n_wide or len(packed.wide_nblk) != packed.n_wide:
                raise ValueError("wide_idx / wide_alen / wide_nblk differ in length")
            check(self._lib.pc_add_alignment_file_wide(
                self._h, packed.n, ntid, _ptr(packed.tid), _ptr(packed.pos), _ptr(packed.alen),
                _ptr(packed.flags), _ptr(packed.nblk), len(packed.blk_start), _ptr(packed.blk_start),
                _ptr(packed.blk_len), packed.n_wide, _ptr(packed.wide_idx), _ptr(packed.wide_alen), _ptr(packed.wide_nblk)))
        else:
            check(self._lib.pc_add_alignment_file(
                self._h, packed.n, ntid, _ptr(packed.tid), _ptr(packed.pos), _ptr(packed.alen),
                _ptr(packed.flags), _ptr(packed.nblk), len(packed.blk_start), _ptr(packed.blk_start),
                _ptr(packed.blk_len)))
        self.nfiles += 1
        self.ntid = ntid
        if getattr(packed, "flag16", None) is not None and getattr(packed, "mapq", None) is not None:
            # the SAM FLAG word and MAPQ of every record: what the vectorised FLAG / MAPQ filter reads (3 bytes per record)
            self.set_alignment_sam(self.nfiles - 1, packed.flag16, packed.mapq)
        if getattr(packed, "nh", None) is not None:
            # the NH:i tag of every record (2 bytes each): what FlagFilterFactory(max_nh=...) tests on the GPU
            nh = _c(packed.nh, np.uint16)
            check(self._lib.pc_set_alignment_nh(self._h, self.nfiles - 1, len(nh), _ptr(nh)))

    def set_alignment_sam(self, file_index, flag16, mapq):
        """Hand the engine the SAM FLAG words and MAPQ values of a staged file (``pc_set_alignment_sam``)."""
        flag16, mapq = _c(flag16, np.uint16), _c(mapq, np.uint8)
        if len(flag16) != len(mapq):
            raise ValueError("flag16 / mapq differ in length")
        check(self._lib.pc_set_alignment_sam(self._h, int(file_index), len(flag16), _ptr(flag16), _ptr(mapq)))

    def set_nh_filter(self, max_nh=0):
        """Keep a read iff it carries an ``NH:i`` tag of at most `max_nh` reported alignments (``pc_set_nh_filter``;
        1: unique mappers); 0 lifts the test."""
        if self._state.get("nhfilter", 0) == int(max_nh):
            return
        check(self._lib.pc_set_nh_filter(self._h, int(max_nh)))
        self._state["nhfilter"] = int(max_nh)

    def set_flag_filter(self, require=0, exclude=0, min_mapq=0, enabled=True):
        """Keep a read iff ``(flag & require) == require and (flag & exclude) == 0 and mapq >= min_mapq``
        (``pc_set_flag_filter``: evaluated on the GPU, folded into the exclusion bit of every staged record);
        ``enabled=False`` lifts the filter."""
        key = (bool(enabled), int(require), int(exclude), int(min_mapq)) if enabled else (False, 0, 0, 0)
        if self._state.get("flagfilter", (False, 0, 0, 0)) == key:
            return
        check(self._lib.pc_set_flag_filter(self._h, 1 if enabled else 0, int(require), int(exclude), int(min_mapq)))
        self._state["flagfilter"] = key

    def set_alignments(self, files, ntid=None):
        self.clear_alignments()
        for f in files:
            self.add_alignment_file(f, ntid)

    def add_bam(self, path, regions=None):
        """Stage a coordinate-sorted BAM file WITHOUT its records ever visiting the host (``pc_add_alignment_bam``): the
        file image goes to HBM, the BGZF members are inflated and the records decoded there, and the packed columns are
        staged by kernels.  Returns the number of mapped reads (pysam's ``AlignmentFile.mapped``).  The engine then
        counts exactly as after ``add_alignment_file(read_bam(path))``; callers that also want the reads themselves
        (``reads_out`` as objects, host-side filters) use :func:`plastid_amd.bam.read_bam_gpu` instead.
        `regions`: stage only the alignments that overlap one of them (``(chrom, start, end)`` or |GenomicSegments|),
        through the file's BAI index -- the BGZF members of ONE span of the file, from the first to the last index chunk of the regions (what lies between far-apart regions included; the overlap test drops it), are uploaded and inflated
        (``pc_add_alignment_bam_span``; what one rank of a multi-GPU job does with its genome range of a shared file);
        the return value is then the number of mapped reads among those staged."""
        import os
        if not os.path.isfile(path):
            raise IOError("No such file: %r" % (path,))
        mapped = ctypes.c_int64(0)
        if regions is not None:
            from .bam import resolve_regions
            sp = resolve_regions(path, regions)
            rc = self._lib.pc_add_alignment_bam_span(self._h, os.fsencode(path), sp["voff_begin"], sp["voff_end"], len(sp["tid"]),
                                                     _ptr(sp["tid"]), _ptr(sp["beg"]), _ptr(sp["end"]), ctypes.byref(mapped))
        else:
            rc = self._lib.pc_add_alignment_bam_path(self._h, os.fsencode(path), ctypes.byref(mapped))
        check(rc)
        self.nfiles += 1
        return int(mapped.value)

    def num_records(self, file_index):
        """Records staged for file `file_index`."""
        return int(self._lib.pc_num_records(self._h, int(file_index)))

    def read_records(self, file_index, indices):
        """Read objects' worth of data for records of a staged file (``pc_read_records`` + ``pc_read_record_runs``):
        dict of arrays ``tid, pos, alen, reverse, nblk, flag16, mapq`` plus ``run_off`` (n + 1), ``run_start``,
        ``run_len`` -- record k's aligned runs are ``run_start/len[run_off[k]:run_off[k + 1]]``."""
        idx = _c(indices, np.int64)
        n = len(idx)
        out = dict(tid=np.zeros(n, np.int32), pos=np.zeros(n, np.int32), alen=np.zeros(n, np.int32), reverse=np.zeros(n, np.uint8),
                   nblk=np.zeros(n, np.int32), flag16=np.zeros(n, np.uint16), mapq=np.zeros(n, np.uint8))
        check(self._lib.pc_read_records(self._h, int(file_index), n, _ptr(idx), _ptr(out["tid"]), _ptr(out["pos"]), _ptr(out["alen"]),
                                        _ptr(out["reverse"]), _ptr(out["nblk"]), _ptr(out["flag16"]), _ptr(out["mapq"])))
        cnt = np.where(out["alen"] > 0, np.maximum(out["nblk"], 1), 0).astype(np.int64)
        off = np.zeros(n + 1, np.int64)
        np.cumsum(cnt, out=off[1:])
        nr = int(off[-1])
        out["run_off"], out["run_start"], out["run_len"] = off, np.zeros(nr, np.int32), np.zeros(nr, np.int32)
        at = np.ascontiguousarray(off[:-1])
        check(self._lib.pc_read_record_runs(self._h, int(file_index), n, _ptr(idx), _ptr(at), nr, _ptr(out["run_start"]), _ptr(out["run_len"])))
        return out

    def update_flags(self, file_index, flags):
        flags = _c(flags, np.uint8)
        check(self._lib.pc_update_flags(self._h, int(file_index), len(flags), _ptr(flags)))

    # --------------------------------------------------------------- mapping
    def set_mapping(self, kind, param=0, fw=None, rc=None, min_len=25, max_len=35):
        if kind in (MAP_VAR5, MAP_STRAT5):
            fw = _c(fw, np.int32)
            rc = _c(rc, np.int32)
            n = len(fw)
            self._state.pop("map", None)
        else:
            fw = rc = None
            n = 0
            key = (int(kind), int(param))
            if self._state.get("map") == key:
                return
        check(self._lib.pc_set_mapping(self._h, int(kind), int(param), _ptr(fw), _ptr(rc), n,
                                       int(min_len), int(max_len)))
        self.kind = int(kind)
        self.rows = self._lib.pc_mapping_rows(self._h)
        if n == 0:
            self._state["map"] = key

    def set_size_filter(self, min_len=None, max_len=-1):
        if self._state.get("size") == (min_len, max_len):
            return
        if min_len is None:
            check(self._lib.pc_set_size_filter(self._h, 0, 0, -1))
        else:
            check(self._lib.pc_set_size_filter(self._h, 1, int(min_len), int(max_len)))
        self._state["size"] = (min_len, max_len)

    def set_normalize(self, enabled, total=1.0):
        if self._state.get("norm") == (bool(enabled), float(total)):
            return
        check(self._lib.pc_set_normalize(self._h, 1 if enabled else 0, float(total)))
        self._state["norm"] = (bool(enabled), float(total))

    # ----------------------------------------------------------------- plans
    def plan(self, tid, start, end, strand, out_off, out_step, row_stride, out_elems, rows=None):
        return Plan(self, tid, start, end, strand, out_off, out_step, row_stride, out_elems,
                    self.rows if rows is None else rows)

    def mapped_reads(self, file_index, rec_lo, rec_hi, tid, start, end, strand_code):
        """``reads_out`` mask of the current mapping rule for one segment over
        records ``[rec_lo, rec_hi)`` of one staged file."""
        n = int(rec_hi) - int(rec_lo)
        mask = np.zeros(max(n, 0), np.uint8)
        if n > 0:
            check(self._lib.pc_mapped_reads(self._h, int(file_index), int(rec_lo), int(rec_hi), int(tid),
                                            int(start), int(end), int(strand_code), _ptr(mask)))
        return mask

    def query_segment(self, tid, start, end, strand_code, reverse, dtype):
        """Counts over ONE segment in ONE call (``pc_query_segment``: no plan object, no upload, no read-back copy)."""
        out = np.empty(int(end) - int(start), dtype)
        check(self._lib.pc_query_segment(self._h, int(tid), int(start), int(end), int(strand_code), 1 if reverse else 0,
                                         OUT_FLOAT64 if dtype == np.float64 else OUT_INT64, out.ctypes.data))
        return out

    def sync(self):
        check(self._lib.pc_sync(self._h))

    def set_profiling(self, level):
        """HIP-event timing of every count: 0 off (default), 1 total + main kernel, 2 all phases."""
        check(self._lib.pc_set_profiling(self._h, int(level)))

    def last_timing(self):
        """ms per phase of the last timed count: dict(total, worklist, hist, long, gather, zero)."""
        ms = np.zeros(6, np.float64)
        k = self._lib.pc_last_timing(self._h, _ptr(ms), 6)
        if k < 0:
            check(k)
        names = ("total", "worklist", "hist", "long", "gather", "zero")
        return dict(zip(names, ms.tolist()))

    def stream_probe(self, nbytes=1 << 30, iters=5):
        """Measured streaming rates of this GPU in GB/s: ``(16-byte loads, 8-byte stores)``."""
        r = np.zeros(2, np.float64)
        check(self._lib.pc_stream_probe(self._h, int(nbytes), int(iters), r.ctypes.data, r.ctypes.data + 8))
        return float(r[0]), float(r[1])

    def reload_knobs(self):
        """Re-read the PC_* environment knobs (they are read once, when the engine is created)."""
        check(self._lib.pc_reload_knobs(self._h))

    def last_algorithmic_bytes(self):
        return int(self._lib.pc_last_algorithmic_bytes(self._h))

    def center_replay_steps(self, plan):
        """``(steps, waves)`` of one center-rule count of `plan` (a diagnostic launch; see ``pc_center_replay_steps``)."""
        steps, waves = ctypes.c_int64(0), ctypes.c_int64(0)
        check(self._lib.pc_center_replay_steps(self._h, plan._h, ctypes.byref(steps), ctypes.byref(waves)))
        return steps.value, waves.value

    def center_row_fill(self, plan):
        """``(row_entries, row_slots)`` of `plan`'s center dispatch list (``pc_center_row_fill``): the fraction of the
        lock-step rows' capacity that holds an entry."""
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        check(self._lib.pc_center_row_fill(self._h, plan._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    @property
    def stream(self):
        return self._lib.pc_stream(self._h)


class Plan(object):
    """A batch of GenomicSegments and the layout of their count vectors (``pc_plan``)."""

    def __init__(self, engine, tid, start, end, strand, out_off, out_step, row_stride, out_elems, rows):
        self.engine = engine
        self._lib = engine._lib
        self.tid = _c(tid, np.int32)
        self.start = _c(start, np.int64)
        self.end = _c(end, np.int64)
        self.strand = _c(strand, np.uint8)
        self.out_off = _c(out_off, np.int64)
        self.out_step = _c(out_step, np.int8)
        self.row_stride = _c(row_stride, np.int64)
        self.nseg = len(self.tid)
        for name in ("start", "end", "strand", "out_off", "out_step", "row_stride"):
            if len(getattr(self, name)) != self.nseg:
                raise ValueError("plan array '%s' has %d entries, expected %d" % (name, len(getattr(self, name)), self.nseg))
        self.out_elems = int(out_elems)
        self.rows = int(rows)
        h = ctypes.c_void_p()
        check(self._lib.pc_plan_create(
            engine._h, self.nseg, _ptr(self.tid), _ptr(self.start), _ptr(self.end), _ptr(self.strand),
            _ptr(self.out_off), _ptr(self.out_step), _ptr(self.row_stride), self.out_elems, self.rows,
            ctypes.byref(h)))
        self._h = h
        # the plan must die before its engine
        self._finalizer = weakref.finalize(self, Plan._destroy, self._lib, h, engine)

    @staticmethod
    def _destroy(lib, h, engine):
        lib.pc_plan_destroy(h)

    def close(self):
        self._finalizer()

    @property
    def positions(self):
        return int(self._lib.pc_plan_positions(self._h))

    @property
    def tiles(self):
        return int(self._lib.pc_plan_tiles(self._h))

    def tables(self):
        """The plan's tables as ``pc_plan_create`` built them (tests compare the GPU builder with the host builder):
        dict of raw ``uint8`` arrays ``tiles`` (32 bytes per record), ``pieces`` (24), ``opieces`` (40), ``gsegs`` (64)
        and ``scalars`` (int64[12]: window size, strand-mode mask, most modes in a window, histogram positions, covered
        output elements, has summed slices, output needs zeroing, tiles, pieces, output pieces, built on the GPU,
        segments)."""
        import ctypes
        out = {}
        for which, name in enumerate(("tiles", "pieces", "opieces", "gsegs", "scalars")):
            n = ctypes.c_int64(0)
            check(self._lib.pc_plan_table(self._h, which, None, 0, ctypes.byref(n)))
            buf = np.zeros(max(int(n.value), 1), np.uint8)
            check(self._lib.pc_plan_table(self._h, which, _ptr(buf), int(n.value), ctypes.byref(n)))
            out[name] = buf[:int(n.value)].view(np.int64) if name == "scalars" else buf[:int(n.value)]
        return out

    def coordinates(self):
        """Genomic coordinate of every output element of the plan's layout (``-1``: not covered)."""
        out = np.empty(self.out_elems, np.int64)
        check(self._lib.pc_plan_coordinates(self.engine._h, self._h, _ptr(out), self.out_elems))
        return out

    def launch(self, dtype):
        """Asynchronously run the counting kernels; results stay in HBM."""
        code = OUT_FLOAT64 if np.dtype(dtype) == np.float64 else OUT_INT64
        check(self._lib.pc_count(self.engine._h, self._h, code))
        self._dtype = np.dtype(np.float64 if code == OUT_FLOAT64 else np.int64)

    def read(self, out=None):
        if out is None:
            out = np.empty(self.out_elems, self._dtype)
        assert out.dtype == self._dtype and out.size == self.out_elems and out.flags.c_contiguous
        check(self._lib.pc_read_counts(self.engine._h, self._h, _ptr(out), self.out_elems))
        return out

    def count(self, dtype, out=None):
        self.launch(dtype)
        return self.read(out)

    def warn_flags(self):
        flags = np.zeros(self.nseg, np.uint8)
        check(self._lib.pc_warn_flags(self.engine._h, self._h, _ptr(flags)))
        return flags

    def warn_details(self):
        """``(flags, last_len)``: per segment whether the reference would warn, and the aligned
        length of the last offending read in fetch order (-1: none)."""
        flags = np.zeros(self.nseg, np.uint8)
        lens = np.full(self.nseg, -1, np.int32)
        check(self._lib.pc_warn_details(self.engine._h, self._h, _ptr(flags), _ptr(lens)))
        return flags, lens

    def mapped_reads(self):
        """``reads_out`` of the current mapping rule for every segment of the plan, in ONE pass: ``(offsets, rec)`` --
        a CSR over (segment, file) pairs (pair ``s * n_files + f``); ``rec[offsets[k]:offsets[k + 1]]`` are the record
        indices, within file ``f``, of the reads the reference's map function would append for segment ``s``."""
        nfiles = int(self._lib.pc_num_files(self.engine._h))
        offsets = np.zeros(self.nseg * max(nfiles, 0) + 1, np.int64)
        total = ctypes.c_int64(0)
        check(self._lib.pc_mapped_reads_batch(self.engine._h, self._h, _ptr(offsets), ctypes.byref(total)))
        rec = np.zeros(total.value, np.uint32)
        check(self._lib.pc_read_mapped_reads(self.engine._h, self._h, _ptr(rec), total.value))
        return offsets, rec

    def rle(self, period=0):
        """Run-length encode the last count on the GPU: ``(starts, values)``; run k covers
        ``[starts[k], starts[k+1])`` (the last one ends at ``out_elems``).  Runs are also cut at
        every multiple of `period` (0: never)."""
        import ctypes
        n = ctypes.c_int64(0)
        check(self._lib.pc_rle(self.engine._h, self._h, int(period), ctypes.byref(n)))
        starts = np.zeros(n.value, np.int64)
        values = np.zeros(n.value, self._dtype)
        check(self._lib.pc_read_rle(self.engine._h, self._h, _ptr(starts), _ptr(values), n.value))
        return starts, values

    def total(self):
        buf = np.zeros(1, self._dtype)
        check(self._lib.pc_total(self.engine._h, self._h, _ptr(buf)))
        return buf[0]

    @property
    def device_ptr(self):
        return self._lib.pc_counts_device_ptr(self._h)

    @property
    def total_device_ptr(self):
        return self._lib.pc_total_device_ptr(self._h)


_default_engines = {}


def default_engine(device=0):
    """Process-wide engine used when map factories are called directly on read lists."""
    if device not in _default_engines:
        _default_engines[device] = Engine(device)
    return _default_engines[device]


def chain_layout(chains_segments, chain_strands, rows=1, stranded=True):
    """Output layout of a batch of chains, mirroring ``SegmentChain.get_counts``
    (roitools.pyx:3259-3271): each chain is a ``[rows, chain_length]`` block; its
    segments sit at their spliced offsets; '-' chains are stored 5'->3'.

    `chains_segments`: list (per chain) of ``[(start, end), ...]`` sorted, merged.
    Returns ``(seg_chain, out_off, out_step, row_stride, chain_base, chain_len, total)``.
    """
    seg_chain, out_off, out_step, row_stride = [], [], [], []
    chain_base, chain_len = [], []
    base = 0
    for ci, (segs, strand) in enumerate(zip(chains_segments, chain_strands)):
        length = sum(e - s for s, e in segs)
        chain_base.append(base)
        chain_len.append(length)
        rev = stranded and strand == "-"
        off = 0
        for s, e in segs:
            seg_chain.append(ci)
            if rev:
                out_off.append(base + length - 1 - off)
                out_step.append(-1)
            else:
                out_off.append(base + off)
                out_step.append(1)
            row_stride.append(length)
            off += e - s
        base += rows * length
    return (np.array(seg_chain, np.int64), np.array(out_off, np.int64), np.array(out_step, np.int8),
            np.array(row_stride, np.int64), np.array(chain_base, np.int64), np.array(chain_len, np.int64), base)
