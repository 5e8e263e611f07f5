"""Loading of the golden fixtures (tests/golden/*.npz, made by make_golden.py
from the reference itself)."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STRAND_CODE = {"\x00": 0, "+": 1, "-": 2, ".": 3}
_cache = {}


class Group(object):
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.arrays = {k: z[k] for k in z.files if k != "manifest"}
        self.cases = json.loads(str(z["manifest"]))

    def aln(self, case):
        pfx = case["aln"]["prefix"]
        keys = ("tid", "pos", "alen", "flags", "nblk", "file_id", "blk_start", "blk_len")
        out = {k: self.arrays["%s_%s" % (pfx, k)] for k in keys}
        for k in ("wide_idx", "wide_alen", "wide_nblk", "flag16", "mapq", "nh"):    # reads beyond the 16-bit / 8-bit fields (wide_reads.npz); FLAG / MAPQ (flag_filters.npz); NH (nh_filters.npz)
            if "%s_%s" % (pfx, k) in self.arrays:
                out[k] = self.arrays["%s_%s" % (pfx, k)]
        return out

    def __getitem__(self, key):
        return self.arrays[key]


def load(name):
    if name not in _cache:
        _cache[name] = Group(name)
    return _cache[name]


def offset_dict_of(spec):
    od = spec.get("offset_dict")
    if od is None:
        return None
    return {(k if k == "default" else int(k)): int(v) for k, v in od.items()}


def tid_of(case, chrom):
    refs = case["aln"]["references"]
    return refs.index(chrom) if chrom in refs else -1
