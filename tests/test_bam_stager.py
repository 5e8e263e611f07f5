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


@pytest.mark.parametrize("seed", range(6))
def test_random_cigars_roundtrip(tmp_path, seed):
    """Random CIGAR strings over all nine operations (runs split by I/S/H/P must merge, N/D must
    split, leading/trailing clips, reads with no aligned base): the native stager's packed arrays
    equal the Python restatement of `AlignedSegment.positions` (SAM spec), across BGZF block sizes
    and thread counts."""
    rng = np.random.default_rng(seed)
    ops_body = [0, 0, 0, 7, 8, 1, 2, 3, 6]          # M = X I D N P
    nref = int(rng.integers(1, 4))
    names = ["r%d" % i for i in range(nref)]
    lens = [int(rng.integers(5000, 60000)) for _ in range(nref)]
    recs = []
    for t in range(nref):
        n = int(rng.integers(0, 400))
        pos = np.sort(rng.integers(0, lens[t] - 3000, n))
        for p in pos:
            cig = []
            if rng.random() < 0.2:
                cig.append((5, int(rng.integers(1, 9))))         # H
            if rng.random() < 0.3:
                cig.append((4, int(rng.integers(1, 9))))         # S
            if rng.random() < 0.03:
                body = [(1, int(rng.integers(1, 30)))]           # insertion only: no aligned base
            else:
                body = []
                for _ in range(int(rng.integers(1, 8))):
                    op = int(rng.choice(ops_body))
                    ln = int(rng.integers(1, 400)) if op == 3 else int(rng.integers(1, 40))
                    if body and body[-1][0] == op:
                        body[-1] = (op, body[-1][1] + ln)
                    else:
                        body.append((op, ln))
                # an alignment starts and ends on an aligned base
                while body and body[0][0] not in (0, 7, 8):
                    body.pop(0)
                while body and body[-1][0] not in (0, 7, 8):
                    body.pop()
                if not body:
                    body = [(0, int(rng.integers(1, 40)))]
            cig += body
            if rng.random() < 0.3:
                cig.append((4, int(rng.integers(1, 9))))
            recs.append((t, int(p), cig, 16 if rng.random() < 0.5 else 0))
    path = str(tmp_path / "rand.bam")
    bam_writer.write_bam(path, names, lens, recs, block_bytes=int(rng.choice([700, 5000, 60000])))
    got = read_bam(path, threads=int(rng.choice([1, 3])))
    exp = PackedAlignments.from_cigars([r[0] for r in recs], [r[1] for r in recs], [r[2] for r in recs],
                                       [bool(r[3] & 16) for r in recs], references=names, lengths=lens)
    assert got.n == len(recs)
    for k in ("tid", "pos", "alen", "flags", "nblk", "blk_start", "blk_len"):
        assert np.array_equal(getattr(got, k), getattr(exp, k)), k
