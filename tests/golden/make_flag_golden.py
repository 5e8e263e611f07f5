#!/usr/bin/env python
"""Golden vectors for read filters that look at the SAM FLAG word and MAPQ, from the REFERENCE ITSELF.

    bash tests/golden/build_scratch_reference.sh /tmp/oracle
    PYTHONPATH=/tmp/oracle:/tmp/oracle/stubs:. python tests/golden/make_flag_golden.py

The reference's filter contract is "a function of the read" (``BAMGenomeArray.add_filter``,
plastid/genomics/genome_array.py:697-722; every filter is called on every fetched read, :819-820).  Here its own
``BAMGenomeArray`` runs with plain Python callables such as ``lambda r: not r.is_secondary and
r.mapping_quality >= 10`` over stub reads that carry ``flag`` / ``mapping_quality`` and the ``is_*`` properties
pysam derives from the flag bits, for all five mapping rules, next to a size filter and under normalisation.

What is written to ``tests/golden/flag_filters.npz`` is DATA ONLY: the packed alignments with their FLAG / MAPQ
columns, the filters as (require, exclude, min_mapq) triples, the queries and what the reference returned."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (helpers only: its groups run under __main__)

from plastid.genomics.genome_array import BAMGenomeArray  # noqa: E402
from plastid.genomics.map_factories import SizeFilterFactory  # noqa: E402
from plastid.genomics.roitools import GenomicSegment, SegmentChain  # noqa: E402

BITS = {"is_paired": 0x1, "is_proper_pair": 0x2, "is_unmapped": 0x4, "mate_is_unmapped": 0x8, "is_reverse": 0x10,
        "mate_is_reverse": 0x20, "is_read1": 0x40, "is_read2": 0x80, "is_secondary": 0x100, "is_qcfail": 0x200,
        "is_duplicate": 0x400, "is_supplementary": 0x800}


class FlaggedBAM(MG.FakeBAM):
    """The stub alignment source of make_golden.py whose reads also carry what pysam derives from FLAG / MAPQ."""

    def __init__(self, packed):
        MG.FakeBAM.__init__(self, packed)
        for i, r in enumerate(self.reads):
            r.flag = int(packed.flag16[i])
            r.mapping_quality = int(packed.mapq[i])
            for name, bit in BITS.items():
                if name != "is_reverse":
                    setattr(r, name, bool(r.flag & bit))
            assert r.is_reverse == bool(r.flag & 0x10)


def with_sam_columns(packed, rng):
    """Random FLAG words (strand bit as packed) and MAPQ values for a packed file."""
    n = packed.n
    flag = np.where(packed.flags & 1, 0x10, 0).astype(np.uint16)
    paired = rng.random(n) < 0.5
    flag[paired] |= 0x1
    flag[paired & (rng.random(n) < 0.7)] |= 0x2
    flag[paired & (rng.random(n) < 0.5)] |= 0x40
    flag[paired & ((flag & 0x40) == 0)] |= 0x80
    flag[paired & (rng.random(n) < 0.5)] |= 0x20
    for bit, frac in ((0x100, 0.15), (0x200, 0.08), (0x400, 0.12), (0x800, 0.05)):
        flag[rng.random(n) < frac] |= bit
    mapq = rng.choice(np.array([0, 1, 3, 9, 10, 11, 20, 30, 42, 60, 255], np.uint8), n)
    packed.flag16, packed.mapq = flag, mapq.astype(np.uint8)
    return packed


def the_callable(req, exc, mq):
    """A filter as a reference user writes it: a plain function of the read's pysam-style properties."""
    names_req = [k for k, b in BITS.items() if req & b]
    names_exc = [k for k, b in BITS.items() if exc & b]

    def keep(read):
        for nm in names_req:
            if not getattr(read, nm):
                return False
        for nm in names_exc:
            if getattr(read, nm):
                return False
        return read.mapping_quality >= mq
    return keep


FILTERS = [
    ("primary, MAPQ >= 10", 0, 0x100, 10),                       # lambda r: not r.is_secondary and r.mapping_quality >= 10
    ("no duplicates, no QC failures", 0, 0x400 | 0x200, 0),
    ("proper pairs, first mate", 0x1 | 0x2 | 0x40, 0, 0),
    ("primary + supplementary out, MAPQ >= 30", 0, 0x100 | 0x800, 30),
    ("everything out (MAPQ 255 wanted, duplicates required and excluded elsewhere)", 0x400, 0x200, 255),
]


def main():
    g = MG.Group("flag_filters")
    rng = np.random.default_rng(20261003)
    refs, lens = ["chrA", "chrB"], [2400, 1200]
    f1 = with_sam_columns(MG.random_packed(rng, 400, refs, lens, 24, 34, gapped_frac=0.25, max_intron=40), rng)
    f2 = with_sam_columns(MG.random_packed(rng, 250, refs, lens, 24, 34, gapped_frac=0.25, max_intron=40), rng)
    segs = [("chrA", 0, 2400), ("chrA", 500, 700), ("chrB", 100, 1150)]
    specs = [{"kind": "fiveprime", "param": 12}, {"kind": "threeprime", "param": 0}, {"kind": "center", "param": 2},
             {"kind": "variable", "offset_dict": {25: 3, 28: 12, "default": 13}},
             {"kind": "stratified", "offset_dict": {"default": 5}, "min_len": 25, "max_len": 31}]
    for files in ([f1], [f1, f2]):
        aln = g.add_alignments(files)
        offs = np.cumsum([0] + [f.n for f in files])
        for fi, (name, req, exc, mq) in enumerate(FILTERS):
            for si, spec in enumerate(specs):
                for extra in ("none", "size", "norm"):
                    if extra != "none" and (si + fi) % 3:
                        continue
                    fakes = [FlaggedBAM(f) for f in files]
                    ga = BAMGenomeArray(fakes, mapping=MG.make_factory(spec))
                    ga.add_filter("flags", the_callable(req, exc, mq))
                    if extra == "size":
                        ga.add_filter("size", SizeFilterFactory(min=26, max=31))
                    if extra == "norm":
                        ga.set_normalize(True)
                    case = {"kind": "ga_flag", "aln": aln, "spec": MG.jsonable_spec(spec), "filter": [req, exc, mq], "filter_name": name,
                            "size_filter": [26, 31] if extra == "size" else None, "normalize": extra == "norm", "sum": ga.sum(), "queries": []}
                    cid = len(g.cases)
                    for qi, (chrom, s, e) in enumerate(segs):
                        for strand in "+-.":
                            seg = GenomicSegment(chrom, s, e, strand)
                            (reads, arr), warns = MG.call_with_warnings(ga.get_reads_and_counts, seg)
                            key = "c%d_q%d%s" % (cid, qi, {"+": "p", "-": "m", ".": "u"}[strand])
                            case["queries"].append({
                                "type": "segment", "chrom": chrom, "start": s, "end": e, "strand": strand,
                                "expected": g.put(key + "_exp", arr),
                                "reads_out": g.put(key + "_reads", np.array([offs[fakes.index(r.file)] + r.index for r in reads], np.int64)),
                                "warned": len(warns) > 0})
                    # one spliced chain per strand through SegmentChain.get_counts
                    for strand in "+-":
                        chain = SegmentChain(GenomicSegment("chrA", 300, 420, strand), GenomicSegment("chrA", 900, 1010, strand),
                                             GenomicSegment("chrA", 1500, 1600, strand))
                        arr, warns = MG.call_with_warnings(chain.get_counts, ga)
                        key = "c%d_chain%s" % (cid, {"+": "p", "-": "m"}[strand])
                        case["queries"].append({"type": "chain", "chrom": "chrA", "strand": strand, "segments": [(300, 420), (900, 1010), (1500, 1600)],
                                                "expected": g.put(key + "_exp", arr), "warned": len(warns) > 0})
                    g.cases.append(case)
    # the columns themselves (concat_file_major carries flag16 / mapq when every file has them)
    g.save()


if __name__ == "__main__":
    main()
