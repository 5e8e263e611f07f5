"""Parity oracle, Python side -- TEST INFRASTRUCTURE ONLY.

ctypes loader for ``libplastid_oracle.so`` (the C restatement in
``plastid_oracle.c``) plus numpy restatements of the thin Python layers that sit
on top of the mapping functions in the reference:

* ``BAMGenomeArray.get_reads_and_counts`` tail -- normalisation and strand flip
  (plastid/genomics/genome_array.py:795-798, 826-830)
* ``SegmentChain.get_counts`` / ``get_masked_counts``
  (plastid/genomics/roitools.pyx:3221-3273, 3275-3315)
* ``VariableFivePrimeMapFactory.__cinit__`` offset tables
  (plastid/genomics/map_factories.pyx:494-543)

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing in ``plastid_amd`` does.

Pinned against the golden vectors in ``tests/golden`` (generated from the
reference itself by ``tests/golden/make_golden.py``).
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libplastid_oracle.so")

FIVE, THREE, CENTER, VAR5, STRAT5 = 0, 1, 2, 3, 4
KIND_NAMES = {"fiveprime": FIVE, "threeprime": THREE, "center": CENTER,
              "variable": VAR5, "stratified": STRAT5}
STRAND_CODE = {"\x00": 0, "+": 1, "-": 2, ".": 3}  # plastid/genomics/c_common.pxd:1-6
TABLE_LEN = 10000

_lib = None


def build(force=False):
    """Compile the C oracle with gcc (test infrastructure)."""
    src = os.path.join(HERE, "plastid_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s", "-B", "libplastid_oracle.so"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB_PATH)
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        L.po_count_segments.restype = ctypes.c_int
        L.po_count_segments.argtypes = [
            i64, vp, vp, vp, vp, vp, vp, vp, vp,          # alignments
            i32, i32, vp, vp, i32, i32, i32, i32, i32,    # mapping + filter
            i64, vp, vp, vp, vp, vp, vp, vp, vp]          # segments + outputs
        L.po_count_segments_mt.restype = ctypes.c_int
        L.po_count_segments_mt.argtypes = L.po_count_segments.argtypes + [i32]
        L.po_count_segments_wide_mt.restype = ctypes.c_int
        L.po_count_segments_wide_mt.argtypes = L.po_count_segments_mt.argtypes + [i64, vp, vp, vp]
        L.po_open.restype = vp
        L.po_open.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp]
        L.po_close.restype = None
        L.po_close.argtypes = [vp]
        L.po_prepared_count.restype = ctypes.c_int
        L.po_prepared_count.argtypes = [vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, i32]
        L.po_cigar_to_runs.restype = ctypes.c_int
        L.po_cigar_to_runs.argtypes = [ctypes.c_int32, i32, vp, vp, i32, vp, vp, vp]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------- tables
def variable_offset_tables(offset_dict):
    """``VariableFivePrimeMapFactory.__cinit__`` (map_factories.pyx:494-543),
    restated: returns ``(forward_offsets, reverse_offsets)`` int32[10000]."""
    fw = np.full(TABLE_LEN, -1, np.int32)          # :511-512
    rc = np.full(TABLE_LEN, -1, np.int32)
    if offset_dict is None:                         # :517-518
        offset_dict = {"default": 0}
    has_default = "default" in offset_dict
    if has_default:                                 # :520-526
        default = int(offset_dict["default"])
        fw[default + 1:] = default
        i = default + 1
        while i < TABLE_LEN:
            rc[i] = i - default - 1
            i += 1
    for read_length, offset in offset_dict.items():  # :530-543
        if read_length != "default":
            if offset >= read_length:
                if not has_default:
                    # `default` unbound in the reference at :533
                    raise UnboundLocalError("local variable 'default' referenced before assignment")
                continue                            # :540 (both branches)
            fw[read_length] = offset
            rc[read_length] = read_length - offset - 1
    return fw, rc


def mapping_spec(kind, param=0, offset_dict=None, min_len=25, max_len=35, size_filter=None):
    """Bundle map-function parameters for :func:`count_segments`."""
    kind = KIND_NAMES.get(kind, kind)
    spec = {"kind": kind, "param": int(param), "fw": None, "rc": None,
            "min_len": int(min_len), "max_len": int(max_len), "size_filter": size_filter}
    if kind in (VAR5, STRAT5):
        spec["fw"], spec["rc"] = variable_offset_tables(offset_dict)
    return spec


def rows_of(spec):
    return spec["max_len"] - spec["min_len"] + 1 if spec["kind"] == STRAT5 else 1


# ------------------------------------------------------------- segment level
class Prepared(object):
    """An alignment set with its per-record arrays derived once (``po_open``: run offsets, end coordinates, contig
    ranges, longest span -- the oracle's stand-in for opening and indexing a BAM file).  ``count_segments(prepared,
    ...)`` then only pays for the counting; bench.py's CPU baseline times the two apart."""

    def __init__(self, aln):
        L = lib()
        a = {k: np.ascontiguousarray(v) for k, v in aln.items()}
        assert a["tid"].dtype == np.int32 and a["pos"].dtype == np.int32
        assert a["alen"].dtype == np.uint16 and a["flags"].dtype == np.uint8 and a["nblk"].dtype == np.uint8
        assert a["blk_start"].dtype == np.int32 and a["blk_len"].dtype == np.int32
        # wide records (reads beyond the 16-bit / 8-bit fields): true lengths / run counts in the side arrays
        a["wide_idx"] = np.ascontiguousarray(a["wide_idx"], np.int64) if "wide_idx" in a else np.zeros(0, np.int64)
        a["wide_alen"] = np.ascontiguousarray(a["wide_alen"], np.int32) if "wide_alen" in a else np.zeros(0, np.int32)
        a["wide_nblk"] = np.ascontiguousarray(a["wide_nblk"], np.int32) if "wide_nblk" in a else np.zeros(0, np.int32)
        self.arrays = a                      # the handle keeps pointers into these
        self.n = len(a["tid"])
        rcode = ctypes.c_int(0)
        self.handle = L.po_open(self.n, _ptr(a["tid"]), _ptr(a["pos"]), _ptr(a["alen"]), _ptr(a["flags"]), _ptr(a["nblk"]),
                                _ptr(a.get("file_id")), _ptr(a["blk_start"]), _ptr(a["blk_len"]), len(a["wide_idx"]),
                                _ptr(a["wide_idx"]), _ptr(a["wide_alen"]), _ptr(a["wide_nblk"]), ctypes.byref(rcode))
        if not self.handle:
            raise RuntimeError("oracle: po_open failed with code %d" % rcode.value)

    def close(self):
        if getattr(self, "handle", None):
            lib().po_close(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def count_segments(aln, spec, seg_tid, seg_start, seg_end, seg_strand, want_mapped=False, threads=1):
    """Run the C oracle: one independent reference ``map_fn`` call per segment.

    `aln` is a dict of packed arrays (``tid,pos,alen,flags,nblk,blk_start,blk_len``
    and optionally ``file_id``; file-major order) or a :class:`Prepared` made from one.  Returns
    ``(arrays, warn_flags[, mapped])`` where ``arrays[s]`` has shape ``(len,)`` or
    ``(rows, len)`` and dtype int64 (point maps) / float64 (center) -- exactly what
    the reference's map function returns for segment ``s``.  `threads` > 1 deals the segments
    to that many POSIX threads (same results; the all-cores CPU baseline of bench.py)."""
    L = lib()
    seg_tid = np.ascontiguousarray(seg_tid, np.int32)
    seg_start = np.ascontiguousarray(seg_start, np.int64)
    seg_end = np.ascontiguousarray(seg_end, np.int64)
    seg_strand = np.ascontiguousarray(seg_strand, np.uint8)
    nseg = len(seg_tid)
    rows = rows_of(spec)
    lens = seg_end - seg_start
    out_off = np.zeros(nseg + 1, np.int64)
    np.cumsum(lens * rows, out=out_off[1:])
    dtype = np.float64 if spec["kind"] == CENTER else np.int64
    out = np.zeros(int(out_off[-1]), dtype)
    warn = np.zeros(nseg, np.uint8)
    prep = aln if isinstance(aln, Prepared) else Prepared(aln)
    mapped = np.zeros((nseg, prep.n), np.uint8) if want_mapped else None
    sf = spec.get("size_filter")
    try:
        rcode = L.po_prepared_count(
            prep.handle, spec["kind"], spec["param"], _ptr(spec["fw"]), _ptr(spec["rc"]),
            spec["min_len"], spec["max_len"],
            0 if sf is None else 1, 0 if sf is None else int(sf[0]), 0 if sf is None else int(sf[1]),
            nseg, _ptr(seg_tid), _ptr(seg_start), _ptr(seg_end), _ptr(seg_strand),
            _ptr(out_off), _ptr(out), _ptr(warn), _ptr(mapped), int(threads))
    finally:
        if prep is not aln:
            prep.close()
    if rcode != 0:
        raise RuntimeError("oracle: po_count_segments failed with code %d" % rcode)
    arrays = []
    for s in range(nseg):
        block = out[out_off[s]:out_off[s + 1]]
        arrays.append(block.reshape(rows, int(lens[s])) if spec["kind"] == STRAT5 else block)
    if want_mapped:
        return arrays, warn, mapped
    return arrays, warn


def get_segment(aln, spec, tid, start, end, strand, roi_order=True, normalize_sum=None,
                known_chrom=True):
    """``BAMGenomeArray.get(GenomicSegment, roi_order)`` (genome_array.py:760-832)."""
    if not known_chrom:                              # :795-798
        shape = [1] + ([rows_of(spec)] if spec["kind"] == STRAT5 else [])
        return np.zeros(shape)
    arrays, _ = count_segments(aln, spec, [tid], [start], [end], [STRAND_CODE[strand]])
    count_array = arrays[0]
    if normalize_sum is not None:                    # :826-827
        count_array = count_array / float(normalize_sum) * 1e6
    if roi_order and strand == "-":                  # :829-830
        count_array = count_array[..., ::-1]
    return count_array


# --------------------------------------------------------------- chain level
def chain_get_counts(aln, spec, tid, segments, strand, stranded=True, normalize_sum=None):
    """``SegmentChain.get_counts(ga, stranded)`` (roitools.pyx:3221-3273).
    `segments` = sorted non-overlapping ``[(start, end), ...]`` of the chain."""
    length = sum(e - s for s, e in segments)
    if length == 0:                                  # :3248-3253
        return np.array([], dtype=float)
    count_arrays = [get_segment(aln, spec, tid, s, e, strand, roi_order=False,
                                normalize_sum=normalize_sum) for s, e in segments]  # :3259
    dims = list(count_arrays[0].shape)               # :3260-3262
    dims[-1] = length
    count_array = np.empty(dims, dtype=float)
    i = 0
    for n, (s, e) in enumerate(segments):            # :3264-3268
        j = i + (e - s)
        count_array[..., i:j] = count_arrays[n]
        i = j
    if strand == "-" and stranded is True:           # :3270-3271
        count_array = count_array[..., ::-1]
    return count_array


def chain_position_mask(segments, mask_segments):
    """``SegmentChain.add_masks`` + ``_set_masks`` (roitools.pyx:2213-2301): int
    flag per chain position in genomic order (1 = masked)."""
    positions = []
    for s, e in segments:
        positions.extend(range(s, e))
    masked = set()
    for s, e in mask_segments:
        masked |= set(range(s, e))
    masked &= set(positions)
    return np.array([1 if p in masked else 0 for p in positions], dtype=np.intc)


def chain_get_masked_counts(aln, spec, tid, segments, strand, mask_segments, normalize_sum=None):
    """``SegmentChain.get_masked_counts`` (roitools.pyx:3275-3315)."""
    counts = chain_get_counts(aln, spec, tid, segments, strand, normalize_sum=normalize_sum)  # :3301
    if not mask_segments:
        mask = np.zeros_like(counts)
    else:
        m = chain_position_mask(segments, mask_segments)
        if strand == "-":                            # :3309-3310
            m = m[::-1]
        mask = np.empty_like(counts)
        mask[..., :] = m
    return np.ma.MaskedArray(counts, mask=mask.astype(bool))


def cigar_to_runs(pos, cigartuples):
    """C restatement of CIGAR -> aligned runs (SAM spec)."""
    L = lib()
    ops = np.array([c[0] for c in cigartuples], np.uint8)
    lens = np.array([c[1] for c in cigartuples], np.int32)
    rs = np.zeros(max(len(ops), 1), np.int32)
    rl = np.zeros(max(len(ops), 1), np.int32)
    al = np.zeros(1, np.int32)
    n = L.po_cigar_to_runs(int(pos), len(ops), _ptr(ops), _ptr(lens), len(rs), _ptr(rs), _ptr(rl), _ptr(al))
    if n < 0:
        raise ValueError("bad cigar")
    return [(int(rs[i]), int(rl[i])) for i in range(n)], int(al[0])
