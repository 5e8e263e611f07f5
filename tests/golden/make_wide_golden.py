#!/usr/bin/env python
"""Golden vectors for WIDE reads -- more than 65 535 aligned positions, more than 255 aligned runs -- from the
REFERENCE ITSELF (same set-up as make_golden.py: the scratch build of build_scratch_reference.sh, stub reads):

    bash tests/golden/build_scratch_reference.sh /tmp/oracle
    PYTHONPATH=/tmp/oracle:/tmp/oracle/stubs:. python tests/golden/make_wide_golden.py

The reference has no limit on either (``read.positions`` is a Python list, map_factories.pyx:243, 349); the packed
format keeps such records behind marker values with their true lengths / run counts aside (plastid_amd/packing.py).
Cases: a 70 000-base read on either strand, a 300-run read (600 aligned bases), ordinary reads before, between and
after them, two files; FivePrime / ThreePrime / Center (with and without nibble) over all of them, Variable and
Stratified over the 300-run file (the reference's offset tables end at length 10 000, map_factories.pxd:10-12: a
70 000-base read would index past them).  Writes tests/golden/wide_reads.npz (data only).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from make_golden import Group, ga_case, mapfn_case   # noqa: E402  (drives the reference)
from plastid_amd.packing import PackedAlignments   # noqa: E402


def main():
    g = Group("wide_reads")
    refs, lens = ["chrW", "chrX"], [200000, 5000]
    rng = np.random.default_rng(4242)

    def ordinary(n, lo, hi):
        out = []
        for _ in range(n):
            p = int(rng.integers(lo, hi))
            L = int(rng.integers(24, 36))
            out.append((0, bool(rng.random() < 0.5), [(p, L)]))
        return out
    long_fw = (0, False, [(1500, 70000)])
    long_rv = (0, True, [(1800, 40000), (42000, 30000)])                     # 70 000 bases in two runs
    many = (0, False, [(900 + 5 * k, 2 + (k % 3 == 0)) for k in range(300)])   # 300 runs, 700 bases
    many_rv = (0, True, [(2000 + 4 * k, 2) for k in range(260)])              # 260 runs, 520 bases

    def packed(recs):
        recs = sorted(recs, key=lambda r: (r[0], r[2][0][0]))
        return PackedAlignments.from_runs([r[0] for r in recs], [r[1] for r in recs], [r[2] for r in recs],
                                          references=refs, lengths=lens)
    both = packed(ordinary(40, 800, 3000) + ordinary(20, 60000, 72500) + [long_fw, long_rv, many, many_rv])
    runs_only = packed(ordinary(40, 800, 3500) + [many, many_rv])
    assert both.n_wide == 4 and runs_only.n_wide == 2
    second = packed(ordinary(30, 900, 2500) + [(0, False, [(1000, 66000)])])
    segs = [("chrW", 0, 4000, "+"), ("chrW", 0, 4000, "-"), ("chrW", 1400, 2600, "."), ("chrW", 60000, 73000, "+"),
            ("chrW", 60000, 73000, "-"), ("chrW", 71400, 71900, "+"), ("chrW", 41500, 42500, "-")]
    queries = [{"type": "segment", "chrom": c, "start": s, "end": e, "strand": st, "roi_order": k != 1} for k, (c, s, e, st) in enumerate(segs)]
    queries += [{"type": "chain", "chrom": "chrW", "strand": "+", "segments": [(1000, 1700), (2100, 2600), (71000, 71600)], "masks": [(1500, 1600)]},
                {"type": "chain", "chrom": "chrW", "strand": "-", "segments": [(1900, 2400), (41900, 42100), (71700, 71900)], "masks": None}]
    od = {26: 6, 28: 12, 30: 13, 520: 100, 700: 40, "default": 10}
    for spec in ({"kind": "fiveprime", "param": 0}, {"kind": "fiveprime", "param": 13}, {"kind": "threeprime", "param": 5},
                 {"kind": "center", "param": 0}, {"kind": "center", "param": 11}):
        ga_case(g, [both], spec, queries, note="wide reads: 70 000 aligned bases, 300 runs")
    ga_case(g, [both, second], {"kind": "center", "param": 2}, queries[:5], note="two files, wide reads in both")
    ga_case(g, [both], {"kind": "fiveprime", "param": 3}, queries[:5], size_filter=(25, 100), note="size filter drops the wide reads")
    for spec in ({"kind": "variable", "offset_dict": od}, {"kind": "stratified", "offset_dict": od, "min_len": 518, "max_len": 522},
                 {"kind": "stratified", "offset_dict": od, "min_len": 25, "max_len": 35}, {"kind": "fiveprime", "param": 600},
                 {"kind": "center", "param": 260}):
        ga_case(g, [runs_only], spec, queries[:3] + queries[-2:-1], note="300-run read under every rule")
    mapfn_case(g, both, {"kind": "threeprime", "param": 0}, "chrW", 0, 80000, "-", note="direct call, wide reads")
    g.save()


if __name__ == "__main__":
    main()
