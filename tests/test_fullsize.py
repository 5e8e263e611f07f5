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
