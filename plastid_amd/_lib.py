"""ctypes binding of the C ABI in ``include/plastid_counts.h``.

The product path fails loudly when the HIP library is missing or no GPU is
usable: there is no CPU fallback for counting.
"""
import ctypes
import os

from .exceptions import EngineError

HERE = os.path.dirname(os.path.abspath(__file__))
#: the product library; ``PLASTID_AMD_LIB`` points experiments at a variant built elsewhere
#: (``build_library(out=...)``) so that they never overwrite the product build
LIB_PATH = os.environ.get("PLASTID_AMD_LIB") or os.path.join(HERE, "libplastid_counts.so")

PC_OK = 0
PC_ERR_ARG = -1
PC_ERR_HIP = -2
PC_ERR_NOMEM = -3
PC_ERR_UNSORTED = -4
PC_ERR_STATE = -5

MAP_FIVE, MAP_THREE, MAP_CENTER, MAP_VAR5, MAP_STRAT5 = 0, 1, 2, 3, 4
STRAND_UNDEF, STRAND_FWD, STRAND_REV, STRAND_UNS = 0, 1, 2, 3
STRAND_NOFILTER = 0x10
OUT_INT64, OUT_FLOAT64 = 0, 1
OFFSET_TABLE_LEN = 10000

#: every symbol ``include/plastid_counts.h`` declares: name -> (restype, argtypes)
_vp, _i64, _i32, _int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int
_pp = ctypes.POINTER(ctypes.c_void_p)
SIGNATURES = {
    "pc_last_error": (ctypes.c_char_p, []),
    "pc_abi_version": (_int, []),
    "pc_device_count": (_int, []),
    "pc_create": (_int, [_int, _pp]),
    "pc_destroy": (_int, [_vp]),
    "pc_reload_knobs": (_int, [_vp]),
    "pc_clear_alignments": (_int, [_vp]),
    "pc_add_alignment_file": (_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "pc_add_alignment_file_wide": (_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "pc_update_flags": (_int, [_vp, _int, _i64, _vp]),
    "pc_set_alignment_sam": (_int, [_vp, _int, _i64, _vp, _vp]),
    "pc_set_flag_filter": (_int, [_vp, _int, ctypes.c_uint32, ctypes.c_uint32, _int]),
    "pc_set_alignment_nh": (_int, [_vp, _int, _i64, _vp]),
    "pc_set_nh_filter": (_int, [_vp, _int]),
    "pc_num_files": (_int, [_vp]),
    "pc_num_records": (_i64, [_vp, _int]),
    "pc_read_records": (_int, [_vp, _int, _i64] + [_vp] * 8),
    "pc_read_record_runs": (_int, [_vp, _int, _i64, _vp, _vp, _i64, _vp, _vp]),
    "pc_set_mapping": (_int, [_vp, _int, _int, _vp, _vp, _int, _int, _int]),
    "pc_set_size_filter": (_int, [_vp, _int, _int, _int]),
    "pc_set_normalize": (_int, [_vp, _int, ctypes.c_double]),
    "pc_mapping_rows": (_int, [_vp]),
    "pc_plan_create": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _pp]),
    "pc_plan_destroy": (_int, [_vp]),
    "pc_plan_positions": (_i64, [_vp]),
    "pc_plan_coordinates": (_int, [_vp, _vp, _vp, _i64]),
    "pc_plan_tiles": (_i64, [_vp]),
    "pc_plan_table": (_int, [_vp, _int, _vp, _i64, ctypes.POINTER(_i64)]),
    "pc_count": (_int, [_vp, _vp, _int]),
    "pc_sync": (_int, [_vp]),
    "pc_query_segment": (_int, [_vp, _i32, _i64, _i64, ctypes.c_uint8, _int, _int, _vp]),
    "pc_release_cached_memory": (_int, [_int]),
    "pc_host_alloc": (_int, [_vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]),
    "pc_host_free": (_int, [_vp, _vp]),
    "pc_read_counts": (_int, [_vp, _vp, _vp, _i64]),
    "pc_counts_device_ptr": (_vp, [_vp]),
    "pc_stream": (_vp, [_vp]),
    "pc_rle": (_int, [_vp, _vp, _i64, ctypes.POINTER(_i64)]),
    "pc_read_rle": (_int, [_vp, _vp, _vp, _vp, _i64]),
    "pc_warn_flags": (_int, [_vp, _vp, _vp]),
    "pc_warn_details": (_int, [_vp, _vp, _vp, _vp]),
    "pc_total": (_int, [_vp, _vp, _vp]),
    "pc_total_device_ptr": (_vp, [_vp]),
    "pc_mapped_reads": (_int, [_vp, _int, _i64, _i64, _i32, _i64, _i64, ctypes.c_uint8, _vp]),
    "pc_mapped_reads_batch": (_int, [_vp, _vp, _vp, ctypes.POINTER(_i64)]),
    "pc_read_mapped_reads": (_int, [_vp, _vp, _vp, _i64]),
    "pc_set_profiling": (_int, [_vp, _int]),
    "pc_last_timing": (_int, [_vp, _vp, _int]),
    "pc_last_algorithmic_bytes": (_i64, [_vp]),
    "pc_center_replay_steps": (_int, [_vp, _vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "pc_center_row_fill": (_int, [_vp, _vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "pc_stream_probe": (_int, [_vp, _i64, _int, _vp, _vp]),
    "pc_bam_open": (_int, [_vp, _vp, _i64, ctypes.c_char_p, _pp]),
    "pc_bam_counts": (_int, [_vp, _vp]),
    "pc_bam_timing": (_int, [_vp, _vp]),
    "pc_bam_nref": (_int, [_vp]),
    "pc_bam_ref_name": (ctypes.c_char_p, [_vp, _int]),
    "pc_bam_ref_length": (_i32, [_vp, _int]),
    "pc_bam_read": (_int, [_vp] * 11),
    "pc_bam_read_sam": (_int, [_vp, _vp, _vp, _vp]),
    "pc_bam_read_nh": (_int, [_vp, _vp]),
    "pc_bam_close": (_int, [_vp]),
    "pc_add_alignment_bam": (_int, [_vp, _vp, _i64, ctypes.c_char_p, ctypes.POINTER(_i64)]),
    "pc_bam_open_path": (_int, [_vp, ctypes.c_char_p, _pp]),
    "pc_add_alignment_bam_path": (_int, [_vp, ctypes.c_char_p, ctypes.POINTER(_i64)]),
    "pc_bam_open_span": (_int, [_vp, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, _int, _vp, _vp, _vp, _pp]),
    "pc_add_alignment_bam_span": (_int, [_vp, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, _int, _vp, _vp, _vp, ctypes.POINTER(_i64)]),
}

_lib = None


#: PC_ABI_VERSION of include/plastid_counts.h this binding was written against
ABI_VERSION = 6


def load():
    """Load ``libplastid_counts.so`` (built in-tree by :mod:`plastid_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            "HIP counting library %s is missing. Build it with "
            "`python -m plastid_amd.build` (needs hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise EngineError("cannot load %s: %s" % (LIB_PATH, e))
    # the revision first: a stale library (an older header's build) must fail HERE, not at a missing symbol below
    try:
        lib.pc_abi_version.restype = _int
        have = lib.pc_abi_version()
    except AttributeError:
        have = None
    if have != ABI_VERSION:
        raise EngineError("%s implements ABI revision %s, this package needs %d: rebuild it with `python -m plastid_amd.build --force`"
                          % (LIB_PATH, have, ABI_VERSION))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    msg = load().pc_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, what=""):
    """Map C error codes to Python exceptions (error conventions of SURVEY 8b)."""
    if rc == PC_OK:
        return
    msg = last_error() or what
    if rc in (PC_ERR_ARG, PC_ERR_UNSORTED):
        raise ValueError(msg)
    if rc == PC_ERR_NOMEM:
        raise MemoryError(msg)
    raise EngineError(msg)
