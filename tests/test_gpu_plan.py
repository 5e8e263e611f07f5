"""The plan of a large annotation is built ON THE GPU (csrc/plan_kernels.hip.h: sorts and scans over the segment table)
and must equal the host builder's (pc_plan_create in csrc/plastid_counts.hip) table by table: tiles, island pieces,
output pieces, the per-segment gather records, and every scalar -- on the golden chains of the reference
(tests/golden/chains.npz: plastid's own SegmentChain test set), on the 479 k-exon human-scale annotation, on
adversarial layouts, and through the counts that follow."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plastid_amd as pa  # noqa: E402
from plastid_amd import synth  # noqa: E402
from plastid_amd.engine import Engine  # noqa: E402

pytestmark = pytest.mark.gpu


def build(monkeypatch, where, reads, args, rows, tile_g=None):
    monkeypatch.setenv("PC_PLAN_BUILD", where)
    if tile_g:
        monkeypatch.setenv("PC_TILE_G", str(tile_g))
    eng = Engine(0)          # (the knobs are read when the engine is made)
    eng.set_alignments([reads])
    plan = eng.plan(*args, rows)
    return eng, plan


def same_tables(a, b):
    ta, tb = a.tables(), b.tables()
    sa, sb = ta["scalars"].copy(), tb["scalars"].copy()
    assert sa[10] != sb[10], "one plan from each builder"
    sa[10] = sb[10] = 0
    assert sa.tolist() == sb.tolist()
    for k in ("tiles", "pieces", "opieces", "gsegs"):
        assert ta[k].shape == tb[k].shape, k
        assert np.array_equal(ta[k], tb[k]), k


def plan_args(p):
    return (p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"])


def both(monkeypatch, reads, args, rows, tile_g=None):
    eh, ph = build(monkeypatch, "host", reads, args, rows, tile_g)
    eg, pg = build(monkeypatch, "gpu", reads, args, rows, tile_g)
    assert int(ph.tables()["scalars"][10]) == 0 and int(pg.tables()["scalars"][10]) == 1
    same_tables(ph, pg)
    return eh, ph, eg, pg


def close(*xs):
    for x in xs:
        x.close()


def test_golden_chains_of_the_reference(monkeypatch):
    """plastid's own SegmentChain test set (tests/golden/chains.npz): every chain's segments, both builders."""
    from tests.test_annotation import _golden_chain_queries, _tables_from_queries
    g, queries = _golden_chain_queries()
    table, _ = _tables_from_queries(pa, queries, ["chrA", "chrB"])
    genome = (["chrA", "chrB"], [1000000, 1000000])
    reads = synth.make_reads(genome, None, 20000, seed=3)
    for rows in (1, 3):
        p = table.plan_arrays(rows=rows)
        eh, ph, eg, pg = both(monkeypatch, reads, plan_args(p), rows)
        close(ph, pg, eh, eg)      # (a plan goes before its engine)


@pytest.mark.parametrize("rows,tile_g", [(1, None), (11, None), (1, 256), (2, 1024)])
def test_human_scale_annotation(monkeypatch, rows, tile_g):
    """60 k transcripts, 479 k exons (the annotation of C4 / C5): identical tables, and identical counts from them."""
    genome, tx, reads, mapping = synth.make_config("C4", scale=0.002, tx_scale=1.0)
    p = tx.plan_arrays(rows=rows)
    assert len(p["tid"]) > 400000
    eh, ph, eg, pg = both(monkeypatch, reads, plan_args(p), rows, tile_g)
    if rows == 1:
        for eng in (eh, eg):
            synth.mapping_factory(("fiveprime", 12))._configure(eng)
        a, b = ph.count(np.int64).copy(), pg.count(np.int64).copy()
        assert a.sum() > 0 and np.array_equal(a, b)
        for eng in (eh, eg):
            synth.mapping_factory(("center", 0))._configure(eng)        # (the center tables of a GPU-built plan come from its fetched tables)
        a, b = ph.count(np.float64).copy(), pg.count(np.float64).copy()
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
        assert np.array_equal(ph.coordinates(), pg.coordinates())
    close(ph, pg, eh, eg)


def test_adversarial_layouts(monkeypatch):
    """Touching, nested, duplicated and overlapping segments on both strands; unknown contigs; empty and clipped
    segments; summed slices; reversed output; segments that span many windows -- and defects, reported as by the host."""
    rng = np.random.default_rng(5)
    genome = (["c%d" % i for i in range(40)], [300000] * 40)
    tx = synth.make_transcripts(genome, 300, 7, "yeast")
    reads = synth.make_reads(genome, tx, 5000, seed=1)
    n = 70000
    tid = rng.integers(-1, 42, n).astype(np.int32)                      # -1 and 40, 41: unknown contigs
    start = rng.integers(-50, 299000, n).astype(np.int64)
    ln = rng.choice([0, 1, 2, 30, 255, 256, 257, 1000, 9000], n, p=[.05, .1, .1, .3, .1, .1, .1, .1, .05]).astype(np.int64)
    start[::7] = (start[::7] // 256) * 256                              # window-aligned starts
    start[1::11] = start[0:-1:11] + ln[0:-1:11]                         # touching the previous segment
    start[2::13] = start[1:-1:13]                                       # duplicates of the previous start
    end = start + ln
    start[5::97] = 2 ** 31 - 300 + rng.integers(0, 200, len(start[5::97]))  # clipped at the top
    end[5::97] = 2 ** 31 + 5
    strand = rng.choice(np.array([0, 1, 2, 3, 0x11, 0x12], np.uint8), n)
    step = rng.choice(np.array([1, -1, 0], np.int8), n, p=[.6, .3, .1])
    L = end - start
    size = np.where(step == 0, (L > 0).astype(np.int64), L)
    for rows in (1, 2):
        off = np.zeros(n, np.int64)
        np.cumsum(size[:-1] * rows, out=off[1:])
        row_stride = size.copy()
        out_off = np.where(step == -1, off + L - 1, off)
        out_elems = int(off[-1] + size[-1] * rows)
        args = (tid, start, end, strand, out_off, step, row_stride, out_elems)
        eh, ph, eg, pg = both(monkeypatch, reads, args, rows)
        if rows == 1:
            for eng in (eh, eg):
                synth.mapping_factory(("threeprime", 0))._configure(eng)
            assert np.array_equal(ph.count(np.int64), pg.count(np.int64))
        close(ph, pg)
        # defects: the lowest bad segment decides the message
        for bad_at, what in ((n - 3, "end"), (17, "step"), (40000, "slice")):
            a = [x.copy() if isinstance(x, np.ndarray) else x for x in args]
            if what == "end":
                a[2][bad_at] = a[1][bad_at] - 1
            elif what == "step":
                a[5][bad_at] = 2
            else:
                a[4][bad_at] = out_elems + 10
                a[2][bad_at] = a[1][bad_at] + 5
            msgs = []
            for eng in (eh, eg):
                with pytest.raises(ValueError) as ei:
                    eng.plan(*a, rows)
                msgs.append(str(ei.value))
            assert msgs[0] == msgs[1] and ("segment %d" % bad_at) in msgs[0], msgs
        close(eh, eg)


def test_large_plans_take_the_gpu_builder_by_default(monkeypatch):
    monkeypatch.delenv("PC_PLAN_BUILD", raising=False)
    genome, tx, reads, mapping = synth.make_config("C4", scale=0.0005, tx_scale=1.0)
    eng = Engine(0)
    eng.set_alignments([reads])
    p = tx.plan_arrays(rows=1)
    plan = eng.plan(*plan_args(p), 1)
    assert int(plan.tables()["scalars"][10]) == 1
    small = eng.plan(*[x[:100] if isinstance(x, np.ndarray) else x for x in plan_args(p)], 1)
    assert int(small.tables()["scalars"][10]) == 0
    # reads_out of a GPU-built plan (the segment arrays come back from HBM when a host pass needs them)
    synth.mapping_factory(("fiveprime", 0))._configure(eng)
    off, idx = plan.mapped_reads()
    assert len(off) == len(p["tid"]) + 1 and off[-1] == len(idx) and len(idx) > 0
    monkeypatch.setenv("PC_PLAN_BUILD", "host")
    eh = Engine(0)
    eh.set_alignments([reads])
    synth.mapping_factory(("fiveprime", 0))._configure(eh)
    ph = eh.plan(*plan_args(p), 1)
    off_h, idx_h = ph.mapped_reads()
    assert np.array_equal(off, off_h) and np.array_equal(idx, idx_h)
    close(plan, small, ph, eng, eh)
