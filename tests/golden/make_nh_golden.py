#!/usr/bin/env python
"""Golden vectors for read filters that look at the ``NH:i`` tag, from the REFERENCE ITSELF.

    bash tests/golden/build_scratch_reference.sh /tmp/oracle
    PYTHONPATH=/tmp/oracle:/tmp/oracle/stubs:. python tests/golden/make_nh_golden.py

The reference's filter contract is "a function of the read" (``BAMGenomeArray.add_filter``,
plastid/genomics/genome_array.py:697-722; every filter is called on every fetched read, :819-820), and the filter its
users write most is the unique-mapper test ``lambda read: read.get_tag("NH") == 1``.  Here the reference's own
``BAMGenomeArray`` runs with such callables -- ``read.has_tag("NH") and read.get_tag("NH") <= k``, alone and next to a
FLAG / MAPQ test -- over stub reads that answer ``has_tag`` / ``get_tag`` as pysam's ``AlignedSegment`` does (KeyError
without the tag), for all five mapping rules, next to a size filter and under normalisation.

What is written to ``tests/golden/nh_filters.npz`` is DATA ONLY: the packed alignments with their FLAG / MAPQ / NH
columns, the filters as (require, exclude, min_mapq, max_nh), the queries and what the reference returned."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (helpers only: its groups run under __main__)
import make_flag_golden as MF  # noqa: E402

from plastid.genomics.genome_array import BAMGenomeArray  # noqa: E402
from plastid.genomics.map_factories import SizeFilterFactory  # noqa: E402
from plastid.genomics.roitools import GenomicSegment, SegmentChain  # noqa: E402


class TaggedBAM(MF.FlaggedBAM):
    """The stub alignment source of make_flag_golden.py whose reads also answer for their NH tag."""

    def __init__(self, packed):
        MF.FlaggedBAM.__init__(self, packed)
        for i, r in enumerate(self.reads):
            nh = int(packed.nh[i])
            r.has_tag = (lambda tag, nh=nh: tag == "NH" and nh > 0)

            def get_tag(tag, nh=nh):
                if tag != "NH" or nh == 0:
                    raise KeyError("tag '%s' not present" % tag)
                return nh
            r.get_tag = get_tag


def with_nh(packed, rng):
    """A random NH column: no tag (0) on a quarter of the reads, unique mappers on half, multi-mappers on the rest."""
    n = packed.n
    u = rng.random(n)
    nh = np.where(u < 0.25, 0, np.where(u < 0.75, 1, rng.integers(2, 12, n))).astype(np.uint16)
    nh[rng.random(n) < 0.01] = 65535
    packed.nh = nh
    return packed


def the_callable(req, exc, mq, max_nh):
    flags = MF.the_callable(req, exc, mq)

    def keep(read):
        return flags(read) and read.has_tag("NH") and read.get_tag("NH") <= max_nh
    return keep


FILTERS = [
    ("unique mappers", 0, 0, 0, 1),                                  # lambda r: r.get_tag("NH") == 1 (on reads that have the tag)
    ("at most two alignments, primary only", 0, 0x100, 0, 2),
    ("NH <= 3, MAPQ >= 10, no duplicates", 0, 0x400, 10, 3),
    ("anything with an NH tag", 0, 0, 0, 65535),
]


def main():
    g = MG.Group("nh_filters")
    rng = np.random.default_rng(20261004)
    refs, lens = ["chrA", "chrB"], [2400, 1200]
    f1 = with_nh(MF.with_sam_columns(MG.random_packed(rng, 400, refs, lens, 24, 34, gapped_frac=0.25, max_intron=40), rng), rng)
    f2 = with_nh(MF.with_sam_columns(MG.random_packed(rng, 250, refs, lens, 24, 34, gapped_frac=0.25, max_intron=40), rng), rng)
    segs = [("chrA", 0, 2400), ("chrA", 500, 700), ("chrB", 100, 1150)]
    specs = [{"kind": "fiveprime", "param": 12}, {"kind": "threeprime", "param": 0}, {"kind": "center", "param": 2},
             {"kind": "variable", "offset_dict": {25: 3, 28: 12, "default": 13}},
             {"kind": "stratified", "offset_dict": {"default": 5}, "min_len": 25, "max_len": 31}]
    for files in ([f1], [f1, f2]):
        aln = g.add_alignments(files)
        offs = np.cumsum([0] + [f.n for f in files])
        for fi, (name, req, exc, mq, max_nh) in enumerate(FILTERS):
            for si, spec in enumerate(specs):
                for extra in ("none", "size", "norm"):
                    if extra != "none" and (si + fi) % 3:
                        continue
                    fakes = [TaggedBAM(f) for f in files]
                    ga = BAMGenomeArray(fakes, mapping=MG.make_factory(spec))
                    ga.add_filter("nh", the_callable(req, exc, mq, max_nh))
                    if extra == "size":
                        ga.add_filter("size", SizeFilterFactory(min=26, max=31))
                    if extra == "norm":
                        ga.set_normalize(True)
                    case = {"kind": "ga_nh", "aln": aln, "spec": MG.jsonable_spec(spec), "filter": [req, exc, mq, max_nh], "filter_name": name,
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
                    for strand in "+-":
                        chain = SegmentChain(GenomicSegment("chrA", 300, 420, strand), GenomicSegment("chrA", 900, 1010, strand),
                                             GenomicSegment("chrA", 1500, 1600, strand))
                        arr, warns = MG.call_with_warnings(chain.get_counts, ga)
                        key = "c%d_chain%s" % (cid, {"+": "p", "-": "m"}[strand])
                        case["queries"].append({"type": "chain", "chrom": "chrA", "strand": strand, "segments": [(300, 420), (900, 1010), (1500, 1600)],
                                                "expected": g.put(key + "_exp", arr), "warned": len(warns) > 0})
                    g.cases.append(case)
    g.save()


if __name__ == "__main__":
    main()
