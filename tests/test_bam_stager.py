"""Native BAM -> packed-array stager (csrc/bam_stager.cpp) against BAM files written by the
test-side writer (SAM/BAM spec) -- CPU only."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plastid_amd import synth  # noqa: E402
from plastid_amd.bam import read_bam  # noqa: E402
from plastid_amd.packing import PackedAlignments, parse_cigar_string  # noqa: E402
from tests import bam_writer  # noqa: E402
from tests import golden_util as gu  # noqa: E402


def test_hand_checked_cigars_roundtrip(tmp_path):
    """The hand-checked gapped reads of the golden fixtures (N, D, I, S, H, =, X)."""
    g = gu.load("quirks")
    case = [c for c in g.cases if c["kind"] == "hand_cigars"][0]
    pos = g["hand_pos"]
    recs = [(0, int(pos[i]), parse_cigar_string(cg), 16 if rev else 0)
            for i, (cg, rev) in enumerate(zip(case["cigars"], case["reverse"]))]
    path = str(tmp_path / "hand.bam")
    bam_writer.write_bam(path, ["chrQ"], [500], recs)
    got = read_bam(path, threads=2)
    exp = PackedAlignments.from_cigars([0] * len(recs), [r[1] for r in recs], case["cigars"], case["reverse"],
                                       references=["chrQ"], lengths=[500])
    for k in ("tid", "pos", "alen", "flags", "nblk", "blk_start", "blk_len"):
        assert np.array_equal(getattr(got, k), getattr(exp, k)), k
    assert got.references == ("chrQ",) and got.lengths == (500,) and got.mapped == len(recs)
    for i in g["hand_indices"]:
        assert got.read(int(i)).positions == list(g["hand_positions_%d" % i])


def test_synthetic_bam_many_blocks(tmp_path):
    genome, tx, reads, _ = synth.make_config("C2", scale=0.0005, tx_scale=0.01)
    path = str(tmp_path / "synth.bam")
    recs = bam_writer.packed_to_records(reads)
    # a few unmapped-but-placed and unplaced reads, like real BAMs have
    recs.insert(10, (recs[10][0], recs[10][1], [], 4))
    tail = [(-1, -1, [], 4)] * 3
    bam_writer.write_bam(path, list(reads.references), list(reads.lengths), recs + tail, block_bytes=20000)
    got = read_bam(path)
    assert got.n == reads.n + 1 and got.mapped == reads.n
    keep = np.ones(got.n, bool)
    keep[10] = False
    assert got.alen[10] == 0 and got.nblk[10] == 0
    for k in ("tid", "pos", "alen", "flags", "nblk"):
        assert np.array_equal(getattr(got, k)[keep], getattr(reads, k)), k
    assert np.array_equal(got.blk_start, reads.blk_start) and np.array_equal(got.blk_len, reads.blk_len)
    assert list(got.references) == list(reads.references) and list(got.lengths) == list(reads.lengths)


def test_errors(tmp_path):
    path = str(tmp_path / "unsorted.bam")
    bam_writer.write_bam(path, ["c"], [1000], [(0, 50, [(0, 30)], 0), (0, 10, [(0, 30)], 0)])
    with pytest.raises(ValueError) as e:
        read_bam(path)
    assert "sorted" in str(e.value)
    with pytest.raises(IOError):
        read_bam(str(tmp_path / "missing.bam"))
    bad = str(tmp_path / "garbage.bam")
    open(bad, "wb").write(b"this is not a bam file at all, not even gzip")
    with pytest.raises(ValueError):
        read_bam(bad)
    trunc = str(tmp_path / "trunc.bam")
    data = open(path, "rb").read()
    open(trunc, "wb").write(data[:len(data) // 2])
    with pytest.raises(ValueError):
        read_bam(trunc)
