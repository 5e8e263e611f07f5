#!/usr/bin/env python
"""Golden text and numbers for the export and region-statistics consumers, from the REFERENCE ITSELF.

Run only in the build container (needs /root/reference):

    bash tests/golden/build_scratch_reference.sh /tmp/oracle
    PYTHONPATH=/tmp/oracle:/tmp/oracle/stubs:. python tests/golden/make_export_golden.py

Drives the reference's own ``BAMGenomeArray.to_bedgraph`` / ``to_variable_step``
(genome_array.py:990-1111) and the per-chain formulas of ``bin/counts_in_region.py:103-125``
(``numpy.nansum(chain.get_masked_counts(ga))``, ``masked_length``, reads per nucleotide, RPKM and
their ``%.8e`` renderings) over stub reads served by the duck-typed alignment source of
``make_golden.py``.  Writes DATA ONLY to ``tests/golden/export_regions.npz``: the packed input
arrays, the query chains / masks and the text / numbers the reference produced.
"""
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import make_golden as mg  # FakeBAM, random_packed, make_factory (imports the scratch reference)
from plastid.genomics.genome_array import BAMGenomeArray
from plastid.genomics.map_factories import SizeFilterFactory
from plastid.genomics.roitools import GenomicSegment, SegmentChain
from plastid_amd.packing import concat_file_major


def main():
    rng = np.random.default_rng(4242)
    refs, lens = ["chrA", "chrB", "chrC"], [5000, 3000, 900]
    packed = mg.random_packed(rng, 1800, refs, lens, 24, 36, gapped_frac=0.2, max_intron=90)
    arrays = {"aln_" + k: v for k, v in concat_file_major([packed]).items()}
    vdict = {26: 12, 27: 12, 28: 13, 29: 13, 30: 14, 31: 13, "default": 13}
    exports = []
    for spec, norm in (({"kind": "fiveprime", "param": 12}, False), ({"kind": "threeprime", "param": 0}, True),
                       ({"kind": "center", "param": 3}, False), ({"kind": "variable", "offset_dict": vdict}, False)):
        ga = BAMGenomeArray([mg.FakeBAM(packed)], mapping=mg.make_factory(spec))
        ga.set_normalize(norm)
        for strand in ("+", "-", "."):
            for window in (100000, 777):
                fh = io.StringIO()
                ga.to_bedgraph(fh, "trk", strand, window_size=window, color="0,0,255")
                exports.append({"what": "bedgraph", "spec": mg.jsonable_spec(spec), "normalize": norm, "strand": strand,
                                "window": window, "kwargs": {"color": "0,0,255"}, "text": fh.getvalue()})
            fh = io.StringIO()
            ga.to_variable_step(fh, "trk", strand, window_size=777)
            exports.append({"what": "variable_step", "spec": mg.jsonable_spec(spec), "normalize": norm, "strand": strand,
                            "window": 777, "kwargs": {}, "text": fh.getvalue()})
    # ---- counts_in_region: the per-chain lines of bin/counts_in_region.py:113-124
    chains = [
        {"name": "c0", "chrom": "chrA", "strand": "+", "segments": [(100, 400)], "masks": []},
        {"name": "c1", "chrom": "chrA", "strand": "-", "segments": [(50, 200), (260, 300), (1000, 1400), (2000, 2600)], "masks": [(60, 80), (290, 1010)]},
        {"name": "c2", "chrom": "chrA", "strand": "+", "segments": [(50, 200), (260, 300), (1000, 1400), (2000, 2600)], "masks": [(0, 5000)]},   # fully masked: length 0 -> nan
        {"name": "c3", "chrom": "chrB", "strand": "-", "segments": [(0, 90), (300, 340)], "masks": [(10, 12), (100, 310)]},
        {"name": "c4", "chrom": "chrB", "strand": "+", "segments": [(700, 1000), (1200, 1210)], "masks": [(1205, 1300)]},
        {"name": "c5", "chrom": "chrC", "strand": ".", "segments": [(0, 900)], "masks": []},
        {"name": "c6", "chrom": "chrZ", "strand": "+", "segments": [(5, 50)], "masks": []},                                      # chromosome not in the data
    ]
    regions = []
    for spec, sf in (({"kind": "fiveprime", "param": 12}, None), ({"kind": "threeprime", "param": 2}, (25, 100)),
                     ({"kind": "variable", "offset_dict": vdict}, (25, 100))):
        ga = BAMGenomeArray([mg.FakeBAM(packed)], mapping=mg.make_factory(spec))
        if sf is not None:
            ga.add_filter("size", SizeFilterFactory(min=sf[0], max=sf[1]))
        ga_sum = ga.sum()
        normconst = 1000.0 * 1e6 / ga_sum
        rows = []
        for c in chains:
            ivc = SegmentChain(*[GenomicSegment(c["chrom"], s, e, c["strand"]) for s, e in c["segments"]], ID=c["name"])
            ivc.add_masks(*[GenomicSegment(c["chrom"], s, e, c["strand"]) for s, e in c["masks"]])
            counts = np.nansum(ivc.get_masked_counts(ga))
            length = ivc.masked_length
            rpnt = np.nan if length == 0 else float(counts) / length
            rpkm = np.nan if length == 0 else rpnt * normconst
            rows.append({"name": c["name"], "region": str(ivc), "counts": float(counts), "length": int(length),
                         "rpnt": None if length == 0 else float(rpnt), "rpkm": None if length == 0 else float(rpkm),
                         "line": "\t".join([c["name"], str(ivc), "%.8e" % counts, "%.8e" % rpnt, "%.8e" % rpkm, "%d" % length])})
        regions.append({"spec": mg.jsonable_spec(spec), "size_filter": sf, "sum": float(ga_sum), "rows": rows})
    manifest = {"references": refs, "lengths": lens, "mapped": int(packed.mapped), "exports": exports, "chains": chains,
                "regions": regions}
    out = os.path.join(HERE, "export_regions.npz")
    np.savez_compressed(out, manifest=np.array(json.dumps(manifest)), **arrays)
    print("wrote %s: %d export texts, %d region tables; %.0f kB" % (out, len(exports), len(regions), os.path.getsize(out) / 1e3))


if __name__ == "__main__":
    main()
