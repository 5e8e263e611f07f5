"""BASELINE-size checks through size-independent properties (tests/fullsize_util.py).

At 100 M reads the C oracle is too slow to replay every query, so the full-size run is pinned
differently: (1) the whole-contig count vectors must equal a numpy ``bincount`` of the mapped
positions -- an expectation that is independent of the HIP path AND of the oracle, and that the
CPU test below first checks against the oracle at a size the oracle handles; (2) every chain
vector of the 20 k-transcript plan must be the spliced slice of those contig vectors; (3) totals
are read counts; (4) two files are counted as the sum of each (linearity); (5) a second launch of
the same plan returns the same bits (idempotence); (6) the center rule conserves one unit per
read, and equals the oracle on a seeded sample of chains.

``PC_FULLSIZE_SCALE`` (default 1.0 = BASELINE C2/C3, 100 M reads) scales the GPU run down.
"""
import os

import numpy as np
import pytest

from plastid_amd import synth
from plastid_amd.packing import PackedAlignments, concat_file_major
from tests import fullsize_util as fu


def test_numpy_expectation_matches_oracle():
    """The bincount expectation is itself pinned against the oracle (and through it against the
    reference's golden vectors) where the oracle is fast."""
    from oracle import oracle
    genome, tx, reads, _ = synth.make_config("C2", scale=0.002, tx_scale=0.02)
    aln = concat_file_major([reads])
    assert np.any(reads.nblk >= 2)                       # gapped reads are part of the case
    for kind, offset in (("fiveprime", 12), ("threeprime", 0), ("fiveprime", 30)):
        vec = fu.contig_vectors(reads, kind, offset)
        p = tx.plan_arrays(rows=1)
        arrays, _ = oracle.count_segments(aln, oracle.mapping_spec(kind, offset), p["tid"], p["start"], p["end"], p["strand"])
        want = np.zeros(p["out_elems"], np.int64)
        for s, a in enumerate(arrays):
            o, st = int(p["out_off"][s]), int(p["out_step"][s])
            want[o + st * np.arange(len(a))] = a
        assert np.array_equal(fu.chain_vectors(tx, vec), want), (kind, offset)
        # whole contigs, both strands
        for code in (1, 2):
            wp = fu.whole_contig_plan(reads.lengths, code)
            arrays, _ = oracle.count_segments(aln, oracle.mapping_spec(kind, offset), wp["tid"], wp["start"], wp["end"], wp["strand"])
            for t, a in enumerate(arrays):
                assert np.array_equal(a, vec[(t, code)]), (kind, offset, t, code)


def _plan(eng, p, rows=1):
    return eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"],
                    p["out_elems"] * rows if rows > 1 else p["out_elems"], rows)


@pytest.mark.gpu
def test_full_size_point_rule_properties():
    from plastid_amd.engine import Engine
    scale = float(os.environ.get("PC_FULLSIZE_SCALE", "1.0"))
    genome, tx, reads, mapping = synth.make_config("C2", scale=scale)
    assert mapping == ("fiveprime", 12)
    vec = fu.contig_vectors(reads, "fiveprime", 12)
    _, ok = fu.mapped_positions(reads, "fiveprime", 12)
    rev = (reads.flags & fu.FLAG_REVERSE) != 0

    eng = Engine(0)
    eng.set_alignments([reads])
    synth.mapping_factory(mapping)._configure(eng)
    # (1) whole contigs, each strand, and (3) totals = mapped reads of that strand
    for code, mask in ((1, ~rev), (2, rev)):
        wp = fu.whole_contig_plan(reads.lengths, code)
        plan = _plan(eng, wp)
        got = plan.count(np.int64)
        want = np.concatenate([vec[(t, code)] for t in range(len(reads.lengths))])
        assert np.array_equal(got, want), "whole-contig vectors differ on strand code %d" % code
        assert int(plan.total()) == int((ok & mask).sum()) == int(want.sum())
        plan.close()
    # (2) the BASELINE plan: 20 k transcripts, spliced, '-' chains reversed
    p = tx.plan_arrays(rows=1)
    plan = _plan(eng, p)
    got = plan.count(np.int64)
    assert np.array_equal(got, fu.chain_vectors(tx, vec)), "chain vectors are not slices of the contig vectors"
    # (5) idempotence of a plan (the kernels leave their scratch as they found it)
    again = plan.count(np.int64)
    assert np.array_equal(got, again)
    # (4) linearity: the records dealt alternately into two files count as the sum of both
    halves = []
    multi = np.nonzero(reads.nblk >= 2)[0]
    rec_of_run = np.repeat(multi, reads.nblk[multi])          # record of every aligned run, in run order
    for k in (0, 1):
        sel = np.arange(k, reads.n, 2)
        runs = np.nonzero((rec_of_run & 1) == k)[0]
        halves.append(PackedAlignments(reads.tid[sel], reads.pos[sel], reads.alen[sel], reads.flags[sel], reads.nblk[sel],
                                       reads.blk_start[runs], reads.blk_len[runs], references=reads.references,
                                       lengths=reads.lengths, validate=scale < 0.05))
    plan.close()
    eng.set_alignments(halves)
    plan = _plan(eng, p)
    both = plan.count(np.int64)
    assert np.array_equal(both, got), "two files do not count as their sum"
    plan.close()
    parts = []
    for h in halves:
        eng.set_alignments([h])
        plan = _plan(eng, p)
        parts.append(plan.count(np.int64))
        plan.close()
    assert np.array_equal(parts[0] + parts[1], got)
    eng.close()


@pytest.mark.gpu
def test_full_size_center_rule_properties():
    from oracle import oracle
    from plastid_amd.engine import Engine
    scale = float(os.environ.get("PC_FULLSIZE_SCALE", "1.0"))
    genome, tx, reads, mapping = synth.make_config("C3", scale=scale)
    assert mapping == ("center", 0)
    eng = Engine(0)
    eng.set_alignments([reads])
    synth.mapping_factory(mapping)._configure(eng)
    rev = (reads.flags & fu.FLAG_REVERSE) != 0
    live = (reads.flags & fu.FLAG_EXCLUDED) == 0
    # every read spreads exactly one unit over its aligned positions (map_factories.pyx:245-254)
    for code, mask in ((1, ~rev), (2, rev)):
        plan = _plan(eng, fu.whole_contig_plan(reads.lengths, code))
        got = plan.count(np.float64)
        n = int((mask & live).sum())
        assert abs(float(got.sum()) - n) <= 1e-9 * n
        assert abs(float(plan.total()) - n) <= 1e-9 * n
        assert got.min() >= 0.0
        plan.close()
    # order-exact float64 sums against the oracle on a seeded sample of chains
    rng = np.random.default_rng(77)
    sel = np.sort(rng.choice(tx.n, size=min(tx.n, 300), replace=False))
    sub = tx.subset(sel)
    p = sub.plan_arrays(rows=1)
    plan = _plan(eng, p)
    got = plan.count(np.float64)
    aln = concat_file_major([reads])
    arrays, _ = oracle.count_segments(aln, oracle.mapping_spec("center", 0), p["tid"], p["start"], p["end"], p["strand"])
    want = np.zeros(p["out_elems"], np.float64)
    for s, a in enumerate(arrays):
        o, st = int(p["out_off"][s]), int(p["out_step"][s])
        want[o + st * np.arange(len(a))] = a
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), "center sums differ from the oracle in some bit"
    plan.close()
    eng.close()


def _scatter(p, arrays, rows=1):
    want = np.zeros(p["out_elems"] * rows, np.int64)
    for s, a in enumerate(arrays):
        a = a.reshape(rows, -1)
        o, st, rs = int(p["out_off"][s]), int(p["out_step"][s]), int(p["row_stride"][s])
        for r in range(rows):
            want[o + r * rs + st * np.arange(a.shape[1])] = a[r]
    return want


def test_sparse_table_expectation_matches_oracle():
    """Variable and Stratified rules: the sparse numpy expectation vs the oracle at oracle size."""
    from oracle import oracle
    genome, tx, reads, mapping = synth.make_config("C5", scale=0.0002, tx_scale=0.005)
    aln = concat_file_major([reads])
    off = fu.offsets_by_length(mapping[1])
    pos, ok = fu.mapped_positions_by_table(reads, off)
    p = tx.plan_arrays(rows=1)
    arrays, _ = oracle.count_segments(aln, oracle.mapping_spec("variable", 0, mapping[1]), p["tid"], p["start"], p["end"], p["strand"])
    assert np.array_equal(fu.sparse_chain_vectors(tx, reads, pos, ok), _scatter(p, arrays))
    # stratified: row L - min_len holds the reads of aligned length L, mapped with the same table
    lo, hi = mapping[2], mapping[3]
    rows = hi - lo + 1
    pr = tx.plan_arrays(rows=rows)
    arrays, _ = oracle.count_segments(aln, oracle.mapping_spec("stratified", 0, mapping[1], lo, hi), pr["tid"], pr["start"], pr["end"], pr["strand"])
    want = _scatter(pr, arrays, rows)
    L = reads.alen.astype(np.int64)
    base = pr["chain_base"]
    for r in range(rows):
        flat = fu.sparse_chain_vectors(tx, reads, pos, ok & (L == lo + r))
        got_r = np.concatenate([want[base[c] + r * tx.length[c]: base[c] + (r + 1) * tx.length[c]] for c in range(tx.n)])
        assert np.array_equal(flat, got_r), r


def test_sparse_row_expectation_matches_oracle():
    """fu.sparse_chain_rows (one pass for all rows) vs the oracle at oracle size, on the blocked generator's records."""
    from oracle import oracle
    genome, tx, lay, mapping = synth.job_layout("C5", scale=0.0002, tx_scale=0.005)
    reads = synth.make_reads_blocked(lay)
    lo, hi = mapping[2], mapping[3]
    rows = hi - lo + 1
    pos, ok = fu.mapped_positions_by_table(reads, fu.offsets_by_length(mapping[1]))
    pr = tx.plan_arrays(rows=rows)
    arrays, _ = oracle.count_segments(concat_file_major([reads]), oracle.mapping_spec("stratified", 0, mapping[1], lo, hi), pr["tid"],
                                      pr["start"], pr["end"], pr["strand"])
    want = _scatter(pr, arrays, rows)[:pr["out_elems"]]     # (_scatter sizes its buffer for rows = 1 plans)
    assert np.array_equal(fu.sparse_chain_rows(tx, reads, pos, ok, reads.alen.astype(np.int64) - lo, rows, threads=4), want)


@pytest.mark.gpu
def test_c5_at_its_full_size():
    """C5 at BASELINE size on ONE GPU: 10^9 mate records (blocked generator, drawn range by range on a thread
    pool), StratifiedVariableFivePrimeMapFactory(25..35), 60 k human-scale transcripts x 11 rows = 1.0e9 outputs --
    every output element against the sparse numpy expectation, the total against the mapped records, a second launch
    bit-identical.  PC_FULLSIZE_SCALE shrinks it."""
    from concurrent.futures import ThreadPoolExecutor
    from plastid_amd.engine import Engine
    from plastid_amd.packing import PackedAlignments
    scale = float(os.environ.get("PC_FULLSIZE_SCALE", "1.0"))
    genome, tx, lay, mapping = synth.job_layout("C5", scale=scale, tx_scale=1.0 if scale >= 0.5 else 0.05)
    nparts = 16
    cuts = lay.cuts(nparts)
    with ThreadPoolExecutor(nparts) as pool:
        parts = list(pool.map(lambda r: synth.make_reads_blocked(lay, *lay.rank_range(cuts, r)), range(nparts)))
    reads = PackedAlignments(*[np.concatenate([getattr(q, k) for q in parts]) for k in ("tid", "pos", "alen", "flags", "nblk", "blk_start", "blk_len")],
                             references=genome[0], lengths=[int(x) for x in genome[1]], validate=False)
    del parts
    assert reads.n == lay.n
    lo, hi = mapping[2], mapping[3]
    rows = hi - lo + 1
    eng = Engine(0)
    eng.set_alignments([reads])
    synth.mapping_factory(mapping)._configure(eng)
    pr = tx.plan_arrays(rows=rows)
    plan = eng.plan(pr["tid"], pr["start"], pr["end"], pr["strand"], pr["out_off"], pr["out_step"], pr["row_stride"],
                    pr["out_elems"], rows)
    got = plan.count(np.int64)
    pos, ok = fu.mapped_positions_by_table(reads, fu.offsets_by_length(mapping[1]))
    want = fu.sparse_chain_rows(tx, reads, pos, ok, reads.alen.astype(np.int64) - lo, rows)
    assert np.array_equal(got, want), "stratified chain vectors differ at full size"
    assert int(plan.total()) == int(want.sum())
    del want
    again = plan.count(np.int64)
    assert np.array_equal(again, got)
    plan.close()
    eng.close()


@pytest.mark.gpu
def test_large_table_rule_properties():
    """C4 at its BASELINE size on ONE GPU (500 M reads, Variable, 60 k human-scale transcripts) and C5 at
    a per-GPU share of its reads over the WHOLE annotation (125 M mate records, Stratified, 60 k
    transcripts x 11 rows = 1.7e9 outputs): engine vs the sparse numpy expectation, every chain, every
    row."""
    from plastid_amd.engine import Engine
    scale = float(os.environ.get("PC_FULLSIZE_SCALE", "1.0"))
    eng = Engine(0)
    # ---- C4: VariableFivePrimeMapFactory
    genome, tx, reads, mapping = synth.make_config("C4", scale=scale, tx_scale=1.0 if scale >= 0.5 else 0.25)
    off = fu.offsets_by_length(mapping[1])
    pos, ok = fu.mapped_positions_by_table(reads, off)
    eng.set_alignments([reads])
    synth.mapping_factory(mapping)._configure(eng)
    p = tx.plan_arrays(rows=1)
    plan = _plan(eng, p)
    got = plan.count(np.int64)
    assert np.array_equal(got, fu.sparse_chain_vectors(tx, reads, pos, ok)), "variable-offset chain vectors differ"
    assert np.array_equal(plan.count(np.int64), got)
    plan.close()
    # ---- C5: StratifiedVariableFivePrimeMapFactory, rows = read lengths
    del reads, pos, ok, got
    genome, tx, reads, mapping = synth.make_config("C5", scale=0.125 * scale, tx_scale=1.0 if scale >= 0.5 else 0.05)
    lo, hi = mapping[2], mapping[3]
    rows = hi - lo + 1
    pos, ok = fu.mapped_positions_by_table(reads, fu.offsets_by_length(mapping[1]))
    eng.set_alignments([reads])
    synth.mapping_factory(mapping)._configure(eng)
    pr = tx.plan_arrays(rows=rows)
    plan = eng.plan(pr["tid"], pr["start"], pr["end"], pr["strand"], pr["out_off"], pr["out_step"], pr["row_stride"],
                    pr["out_elems"], rows)
    got = plan.count(np.int64)
    per_chain = tx.split_counts(got, rows)
    L = reads.alen.astype(np.int64)
    total = 0
    for r in range(rows):
        want = fu.sparse_chain_vectors(tx, reads, pos, ok & (L == lo + r))
        have = np.concatenate([m[r] for m in per_chain])
        assert np.array_equal(have, want), "stratified row %d (length %d) differs" % (r, lo + r)
        total += int(want.sum())
    assert int(plan.total()) == total
    plan.close()
    eng.close()


@pytest.mark.gpu
def test_c1_workload_fully_vs_oracle():
    """BASELINE configs[0] (C1: 1 M reads, FivePrimeMapFactory(offset=0), 200 SegmentChains -- the
    reference's own CPU-runnable shape; the real test BAM is not in the container) through
    the HIP path, EVERY output position against the oracle; plus the CLI size filter and the batch
    API of the Python mirror."""
    from oracle import oracle
    import plastid_amd as pa
    from plastid_amd.engine import Engine
    genome, tx, reads, mapping = synth.make_config("C1")
    assert mapping == ("fiveprime", 0) and reads.n == 1_000_000 and tx.n == 200
    aln = concat_file_major([reads])
    p = tx.plan_arrays(rows=1)
    eng = Engine(0)
    eng.set_alignments([reads])
    synth.mapping_factory(mapping)._configure(eng)
    plan = _plan(eng, p)
    for sf in (None, (25, 100)):
        eng.set_size_filter(*(sf if sf else (None,)))
        got = plan.count(np.int64)
        arrays, _ = oracle.count_segments(aln, oracle.mapping_spec("fiveprime", 0, size_filter=sf), p["tid"], p["start"], p["end"], p["strand"])
        assert np.array_equal(got, _scatter(p, arrays)), sf
    eng.set_size_filter(None)
    plan.close()
    eng.close()
    # the same through the drop-in classes: ga.get_counts_batch / chain.get_counts on a few chains
    ga = pa.BAMGenomeArray(reads, mapping=pa.FivePrimeMapFactory(0))
    chains = tx.chains(limit=20)
    arrays, _ = oracle.count_segments(aln, oracle.mapping_spec("fiveprime", 0), p["tid"], p["start"], p["end"], p["strand"])
    want = tx.split_counts(_scatter(p, arrays).astype(np.float64))
    for c, chain in enumerate(chains):
        assert np.array_equal(chain.get_counts(ga), want[c])
