"""Golden vectors for the length-stratified consumer of the counting path: the reference's
``plastid.bin.psite.do_count`` (bin/psite.py:88-243) run on stub reads.

Run in THIS container against the scratch build of the reference:

    bash tests/golden/build_scratch_reference.sh /tmp/oracle
    PYTHONPATH=/tmp/oracle:/tmp/oracle/stubs:. python tests/golden/make_psite_golden.py

Writes tests/golden/psite_do_count.npz: packed reads, the ROI table (strings + numbers), the
parameters of every case, and what ``do_count`` returned (raw / normalised count matrices with
their masks, profile columns).  Data only.
"""
import json
import os
import sys
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from make_golden import FakeBAM  # noqa: E402  (stub-read alignment source)
from plastid.bin.psite import do_count  # noqa: E402
from plastid.genomics.genome_array import BAMGenomeArray  # noqa: E402
from plastid.genomics.map_factories import FivePrimeMapFactory, ThreePrimeMapFactory  # noqa: E402
from plastid.genomics.roitools import GenomicSegment, SegmentChain  # noqa: E402
from plastid_amd.packing import PackedAlignments  # noqa: E402


def build_inputs(seed=31):
    rng = np.random.default_rng(seed)
    names, lens = ["chrA", "chrB"], [9000, 6000]
    # ROIs: windows of up to 120 nt around a landmark; some truncated at the 5' end (alignment
    # offset > 0), two spliced, both strands, one with a masked stretch
    window, flank = 120, 40
    rois, masks, offsets = [], [], []
    for i in range(26):
        chrom = names[i % 2]
        strand = "+-"[(i // 2) % 2]
        start = int(rng.integers(100, lens[i % 2] - 400))
        trunc = int(rng.choice([0, 0, 0, 7, 25]))
        length = window - trunc
        if i % 5 == 0:   # spliced ROI
            cut = int(rng.integers(20, length - 20))
            gap = int(rng.integers(30, 200))
            segs = [GenomicSegment(chrom, start, start + cut, strand),
                    GenomicSegment(chrom, start + cut + gap, start + gap + length, strand)]
        else:
            segs = [GenomicSegment(chrom, start, start + length, strand)]
        roi = SegmentChain(*segs)
        rois.append(str(roi))
        if i % 6 == 1:
            masks.append(str(SegmentChain(GenomicSegment(chrom, start + 30, start + 42, strand))))
        else:
            masks.append("na")
        offsets.append(trunc)
    # reads: dense around the ROIs, lengths 14..36, a few gapped
    tids, rev, runs = [], [], []
    for roi_s in rois:
        roi = SegmentChain.from_str(roi_s)
        t = names.index(roi.chrom)
        n = int(rng.integers(30, 400))
        for _ in range(n):
            L = int(rng.integers(14, 37))
            p = int(rng.integers(max(0, roi.spanning_segment.start - 45), roi.spanning_segment.end + 5))
            if rng.random() < 0.04:
                a = int(rng.integers(3, L - 3))
                runs.append([(p, a), (p + a + int(rng.integers(1, 90)), L - a)])
            else:
                runs.append([(p, L)])
            tids.append(t)
            rev.append(bool(rng.random() < (0.85 if roi.strand == "-" else 0.15)))
    packed = PackedAlignments.from_runs(tids, rev, runs, references=names, lengths=lens, sort=True)
    table = pd.DataFrame({"region": rois, "masked": masks, "alignment_offset": offsets,
                          "window_size": window, "zero_point": flank,
                          "region_id": ["roi%d" % i for i in range(len(rois))]})
    return packed, table


def main():
    packed, table = build_inputs()
    out = {"tid": packed.tid, "pos": packed.pos, "alen": packed.alen, "flags": packed.flags, "nblk": packed.nblk,
           "blk_start": packed.blk_start, "blk_len": packed.blk_len}
    meta = {"references": list(packed.references), "lengths": [int(x) for x in packed.lengths],
            "table": {c: [x if isinstance(x, str) else int(x) for x in table[c]] for c in table.columns},
            "cases": []}
    cases = [("fiveprime", 0, 20, 50, 10, 25, 30, False), ("fiveprime", 0, 10, 60, 3, 14, 20, True),
             ("fiveprime", 16, 0, 120, 1, 14, 19, False), ("threeprime", 2, 20, 80, 5, 26, 29, False)]
    for ci, (kind, off, ns, ne, min_counts, lo, hi, agg) in enumerate(cases):
        fac = FivePrimeMapFactory(off) if kind == "fiveprime" else ThreePrimeMapFactory(off)
        ga = BAMGenomeArray(FakeBAM(packed), mapping=fac)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            raw, norm, prof = do_count(table, ga, ns, ne, min_counts, lo, hi, aggregate=agg)
        for k in raw:
            out["c%d_raw_%d" % (ci, k)] = np.asarray(raw[k].data, float)
            out["c%d_rawmask_%d" % (ci, k)] = np.asarray(np.ma.getmaskarray(raw[k]))
            out["c%d_norm_%d" % (ci, k)] = np.asarray(np.ma.getdata(norm[k]), float)
        for c in prof.columns:
            col = prof[c].values
            out["c%d_prof_%s" % (ci, c)] = np.asarray(np.ma.filled(np.ma.masked_invalid(np.ma.asarray(col, float)), np.nan))
            if hasattr(col, "mask"):
                pass
        meta["cases"].append(dict(kind=kind, offset=off, norm_start=ns, norm_end=ne, min_counts=min_counts,
                                  min_len=lo, max_len=hi, aggregate=agg, columns=list(prof.columns)))
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(HERE, "psite_do_count.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", packed.n, "reads,", len(table), "ROIs")


if __name__ == "__main__":
    main()
