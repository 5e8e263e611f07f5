"""BED -> interval table and SegmentChain.from_bed (CPU only)."""
import io
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plastid_amd as pa  # noqa: E402
from plastid_amd.annotation import IntervalTable, read_bed  # noqa: E402

BED = """track name=test
# a comment
chrA\t100\t1100\ttx1\t0\t+\t200\t900\t0,0,0\t3\t100,200,300\t0,400,700
chrB\t50\t80\ttx2\t5\t-
chrA\t10\t20
browser position chrA:1-100
chrZ\t5\t50\ttx4\t0\t-\t5\t5\t0\t2\t10,10\t0,35
"""


def test_from_bed_line_and_reader():
    chains = read_bed(io.StringIO(BED))
    assert len(chains) == 4
    c = chains[0]
    assert [(s.start, s.end) for s in c] == [(100, 200), (500, 700), (800, 1100)]
    assert c.strand == "+" and c.chrom == "chrA" and c.length == 600 and c.attr["ID"] == "tx1"
    assert c.attr["thickstart"] == 200 and c.attr["thickend"] == 900
    assert chains[1].strand == "-" and chains[1].attr["score"] == 5.0
    assert chains[2].strand == "." and chains[2].get_name() == "chrA:10-20(.)"
    assert pa.SegmentChain.from_bed("chrA\t10\t20\tx\t0\t-") == pa.SegmentChain(pa.GenomicSegment("chrA", 10, 20, "-"))
    with pytest.raises(ValueError):
        pa.SegmentChain.from_bed("chrA\t10")


def test_interval_table_layout_matches_chains():
    refs = ["chrA", "chrB"]
    tab = IntervalTable.from_bed(io.StringIO(BED), refs)
    assert tab.n == 4 and tab.n_segments == 7 and tab.tid.tolist() == [0, 1, 0, -1]
    assert tab.length.tolist() == [600, 30, 10, 20] and tab.ids[3] == "tx4"
    tab2 = IntervalTable.from_chains(read_bed(io.StringIO(BED)), refs)
    for k in ("tid", "strand", "ex_off", "ex_start", "ex_end"):
        assert np.array_equal(getattr(tab, k), getattr(tab2, k)), k
    p = tab.plan_arrays(rows=2)
    assert p["out_elems"] == 2 * 660
    # '-' chains are laid out 5'->3': the last genomic position of chain 1 is output element 0 of its block
    base = p["chain_base"][1]
    seg = 3  # the single exon of tx2
    assert p["out_step"][seg] == -1 and p["out_off"][seg] == base + 29 and p["row_stride"][seg] == 30
    flat = np.arange(p["out_elems"])
    views = tab.split_counts(flat, rows=2)
    assert views[0].shape == (2, 600) and views[1].shape == (2, 30) and views[1][0, 0] == base
    chains = tab.chains()
    assert str(chains[0]) == "chrA:100-200^500-700^800-1100(+)"


# ---------------------------------------------------------------------------- GTF2
def _gtf2_golden():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gtf2_transcripts.json")) as fh:
        return json.load(fh)


def test_gtf2_assembly_matches_reference_golden():
    """Transcripts, exon structure, CDS bounds, rejected ids and ORDER as the reference's
    GTF2_TranscriptAssembler gave them (fixture: tests/golden/make_gtf2_golden.py)."""
    import warnings
    from plastid_amd.annotation import read_gtf2
    gold = _gtf2_golden()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        chains = read_gtf2(io.StringIO(gold["gtf2"]))
    rejected = sorted(str(x.message).split("'")[1] for x in w if "Rejecting" in str(x.message))
    assert rejected == gold["rejected"]
    assert all(issubclass(x.category, pa.DataWarning) for x in w)
    got = [[c.get_name(), str(c), c.attr.get("cds_genome_start"), c.attr.get("cds_genome_end"),
            c.attr.get("gene_id")] for c in chains]
    assert got == gold["transcripts"]
    assert all(c.attr["tag"] == "a;b" for c in chains if c.attr.get("gene_id", "").startswith("g"))


def test_interval_table_from_gtf2():
    gold = _gtf2_golden()
    import warnings
    refs = ["chrI", "chrII", "chrM"]  # '2-micron' is not in the array -> tid -1
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tab = IntervalTable.from_gtf2(io.StringIO(gold["gtf2"]), refs)
    assert tab.n == len(gold["transcripts"]) and tab.ids == [r[0] for r in gold["transcripts"]]
    for c, row in zip(tab.chains(), gold["transcripts"]):
        if not row[1].startswith("2-micron"):
            assert str(c) == row[1]
    assert (tab.tid == -1).sum() == sum(r[1].startswith("2-micron") for r in gold["transcripts"])
    assert (np.diff(tab.ex_start) > 0)[tab.ex_tx[1:] == tab.ex_tx[:-1]].all()


# ---------------------------------------------------------------------------- GFF3
def _gff3_golden():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gff3_transcripts.json")) as fh:
        return json.load(fh)


def test_gff3_assembly_matches_reference_golden():
    """Transcripts (Parent-linked, shared-ID, implied by a non-transcript parent, exons with several
    parents), CDS bounds, gene ids, types, rejected ids and ORDER (``###`` batches, each sorted) as the
    reference's GFF3_TranscriptAssembler gave them (fixture: tests/golden/make_gff3_golden.py)."""
    import warnings
    from plastid_amd.annotation import read_gff3
    gold = _gff3_golden()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        chains = read_gff3(io.StringIO(gold["gff3"]))
    rejected = sorted(str(x.message).split("'")[1] for x in w if "Rejecting" in str(x.message))
    assert rejected == gold["rejected"]
    got = [[c.attr["transcript_id"], str(c), c.attr.get("cds_genome_start"), c.attr.get("cds_genome_end"),
            c.attr.get("gene_id"), c.attr.get("type")] for c in chains]
    assert got == gold["transcripts"]


def test_interval_table_from_gff3():
    import warnings
    gold = _gff3_golden()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tab = IntervalTable.from_gff3(io.StringIO(gold["gff3"]), ["chrI", "chrII", "chrM"])
    assert tab.ids == [r[0] for r in gold["transcripts"]]
    assert [str(c) for c in tab.chains()] == [r[1] for r in gold["transcripts"]]


def test_gtf2_attribute_fast_path_equals_the_tokenizer():
    """The split-based fast path of the ninth GTF2 column must agree with the general tokenizer
    (quoted semicolons, escapes, bare numbers, empty values, repeated keys) on random strings."""
    import random
    from plastid_amd import annotation as A

    def tokenizer_only(text):
        out = {}
        for m in A._GTF2_TOKEN.finditer(text.strip()):
            key = m.group(1)
            val = m.group(2) if m.group(2) is not None else m.group(3)
            out[key] = "%s,%s" % (out[key], val) if key in out else val
        return out

    rng = random.Random(11)
    keys = ["gene_id", "transcript_id", "exon_number", "gene_name", "tag", "note", "level"]

    def rand_val():
        r = rng.random()
        if r < 0.5:
            return '"%s"' % "".join(rng.choice("abcXYZ012._-") for _ in range(rng.randint(0, 8)))
        if r < 0.6:
            return '"a;b c"'
        if r < 0.7:
            return '"with \\\\"esc\\\\" quote"'
        if r < 0.8:
            return str(rng.randint(0, 99))
        if r < 0.9:
            return '"two words"'
        return '""'

    for _ in range(5000):
        parts = ["%s%s%s" % (rng.choice(keys), rng.choice([" ", "  ", "\\t"]), rand_val()) for _ in range(rng.randint(0, 6))]
        text = rng.choice(["; ", ";", " ; ", ";  "]).join(parts) + rng.choice(["", ";", "; "])
        assert A._gtf2_attributes(text) == tokenizer_only(text), text
