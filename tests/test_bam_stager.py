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


def test_corrupt_bam_is_rejected_not_crashed(tmp_path):
    """Damaged files -- flipped bytes, truncation, absurd length fields, both in the BGZF container
    and inside well-formed BGZF blocks -- either load or raise; they never take the process down
    (the same loop ran clean under AddressSanitizer / UBSan during development)."""
    import struct
    from plastid_amd.exceptions import MalformedFileError
    genome, tx, reads, _ = synth.make_config("C4", scale=0.00001, tx_scale=0.001)
    recs = bam_writer.packed_to_records(reads)
    refs, lens = list(reads.references), list(reads.lengths)
    text = b"@HD\\tVN:1.6\\tSO:coordinate\\n"
    head = b"BAM\\x01" + struct.pack("<I", len(text)) + text + struct.pack("<I", len(refs))
    for nm, ln in zip(refs, lens):
        nmb = nm.encode() + b"\\x00"
        head += struct.pack("<I", len(nmb)) + nmb + struct.pack("<I", ln)
    good = head + b"".join(bam_writer.encode_record(t, p, c, f) for t, p, c, f in recs)
    rng = np.random.default_rng(5)
    path = str(tmp_path / "damaged.bam")
    outcomes = {"ok": 0, "rejected": 0}
    for it in range(150):
        b = bytearray(good)
        mode = int(rng.integers(0, 4))
        lo = 0 if rng.random() < 0.2 else len(head)
        if mode == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(lo, len(b)))] = int(rng.integers(0, 256))
        elif mode == 1:
            b = b[:int(rng.integers(lo, len(b)))]
        else:
            i = int(rng.integers(lo, len(b) - 4))
            b[i:i + 4] = struct.pack("<I", int(rng.choice([0, 1, 0x7fffffff, 0xffffffff, 0x80000000, 65536])))
        blob = b"".join(bam_writer.bgzf_block(bytes(b[o:o + 3000])) for o in range(0, len(b), 3000)) + bam_writer.BGZF_EOF
        if mode == 3:   # damage the container itself
            blob = bytearray(blob)
            blob[int(rng.integers(0, len(blob)))] ^= 0x5a
            blob = bytes(blob[:int(rng.integers(len(blob) // 2, len(blob) + 1))])
        with open(path, "wb") as fh:
            fh.write(blob)
        try:
            read_bam(path, threads=int(rng.integers(1, 3)))
            outcomes["ok"] += 1
        except (ValueError, MalformedFileError, OSError):
            outcomes["rejected"] += 1
    assert outcomes["rejected"] > 30 and outcomes["ok"] + outcomes["rejected"] == 150


@pytest.mark.parametrize("piece", [1, 2, 7])
def test_piecewise_decoding_is_the_serial_walk(tmp_path, monkeypatch, piece):
    """The reader decodes pieces of the record stream in parallel and stitches them; with tiny
    pieces (PB_PIECE: the region loader's record pieces; PB_CHUNK: the whole-file loader's chunks of
    BGZF members, here one member each, down to members shorter than a record) every cross-piece
    check is exercised: same arrays as one piece, and the same verdict on files whose defect sits on
    a piece boundary."""
    genome, tx, reads, _ = synth.make_config("C4", scale=0.00002, tx_scale=0.001)
    recs = bam_writer.packed_to_records(reads)
    refs, lens = list(reads.references), list(reads.lengths)
    path = str(tmp_path / "p.bam")
    bam_writer.write_bam(path, refs, lens, recs, block_bytes=900)
    whole = read_bam(path, threads=1)
    exp = PackedAlignments.from_cigars([r[0] for r in recs], [r[1] for r in recs], [r[2] for r in recs],
                                       [bool(r[3] & 16) for r in recs], references=refs, lengths=lens)
    for name in ("tid", "pos", "alen", "flags", "nblk", "blk_start", "blk_len"):
        assert np.array_equal(getattr(whole, name), getattr(exp, name)), name
    monkeypatch.setenv("PB_PIECE", str(piece))
    monkeypatch.setenv("PB_CHUNK", "1")
    for block_bytes in (900, 150, 23 * piece):
        bam_writer.write_bam(path, refs, lens, recs, block_bytes=block_bytes)
        monkeypatch.setenv("PB_HEAD", "16" if block_bytes == 900 else "32768")   # heads shorter than a record
        for th in (1, 3):
            got = read_bam(path, threads=th)
            for name in ("tid", "pos", "alen", "flags", "nblk", "blk_start", "blk_len"):
                assert np.array_equal(getattr(got, name), getattr(whole, name)), (name, block_bytes)
            assert got.mapped == whole.mapped
    # defects at every position of a short file: disorder, a placed record after an unplaced one,
    # a leading deletion that breaks the order of the first aligned positions
    base = [(0, 100 + 10 * i, [(0, 30)], 0) for i in range(9)]
    for i in range(1, 9):
        bad = list(base)
        bad[i] = (0, 5, [(0, 30)], 0)
        bam_writer.write_bam(path, ["c"], [5000], bad)
        with pytest.raises(ValueError, match="sorted"):
            read_bam(path, threads=2)
        bad = list(base)
        bad[i - 1] = (-1, -1, [], 4)
        bam_writer.write_bam(path, ["c"], [5000], bad)
        with pytest.raises(ValueError, match="sorted"):
            read_bam(path, threads=2)
        bad = list(base)
        bad[i - 1] = (0, bad[i][1] - 2, [(2, 1), (0, 30)], 0) if i > 1 else bad[i - 1]   # D first: spos = pos + 1
        bad[i - 1] = (0, bad[i][1] - 1, [(2, 5), (0, 30)], 0)                             # starts after record i
        bam_writer.write_bam(path, ["c"], [5000], bad)
        with pytest.raises(ValueError, match="deletion|sorted"):
            read_bam(path, threads=2)
    # unplaced reads at the end are fine
    bam_writer.write_bam(path, ["c"], [5000], base + [(-1, -1, [], 4)] * 3)
    ok = read_bam(path, threads=2)
    assert ok.n == 9 and ok.mapped == 9


@pytest.mark.parametrize("seed", range(4))
def test_region_loading_through_the_index(tmp_path, monkeypatch, seed):
    """``read_bam(path, regions=...)`` walks the BAI index (bins, chunks, linear index) and must return
    exactly the alignments htslib's fetch would: those with ``pos < end and endpos > start`` for
    some region, once each, in file order; ``mapped`` comes from the index."""
    rng = np.random.default_rng(100 + seed)
    genome, tx, reads, _ = synth.make_config("C4", scale=0.00004, tx_scale=0.002, seed_shift=seed)   # spliced, human-scale contigs
    recs = bam_writer.packed_to_records(reads)
    path = str(tmp_path / "r.bam")
    bam_writer.write_bam(path, list(reads.references), list(reads.lengths), recs + [(-1, -1, [], 4)] * 2,
                         block_bytes=int(rng.choice([300, 4000, 60000])), index=True)
    whole = read_bam(path, threads=1)
    assert whole.n == reads.n
    end = whole.pos.astype(np.int64) + np.maximum(whole.alen, 1)
    multi = np.nonzero(whole.nblk >= 2)[0]
    off = whole.block_offsets()
    last = off[multi] + whole.nblk[multi] - 1
    end[multi] = whole.blk_start[last].astype(np.int64) + whole.blk_len[last]
    if seed == 1:
        monkeypatch.setenv("PB_PIECE", "3")
    for trial in range(12):
        regs = []
        for _ in range(int(rng.integers(1, 6))):
            if rng.random() < 0.7 and reads.n:                      # around a read
                i = int(rng.integers(0, reads.n))
                t, c = int(whole.tid[i]), int(whole.pos[i]) + int(rng.integers(-2000, 2000))
            else:
                t = int(rng.integers(0, len(reads.references)))
                c = int(rng.integers(0, reads.lengths[t]))
            s = max(0, c)
            regs.append((reads.references[t], s, s + int(rng.choice([1, 50, 5000, 300000]))))
        if trial == 0:
            regs.append(("not_a_contig", 0, 1000))
        keep = np.zeros(whole.n, bool)
        for chrom, s, e in regs:
            if chrom in reads.references:
                t = reads.references.index(chrom)
                keep |= (whole.tid == t) & (whole.pos < e) & (end > s)
        got = read_bam(path, threads=int(rng.integers(1, 4)), regions=regs)
        idx = np.nonzero(keep)[0]
        assert got.n == len(idx), (regs, got.n, len(idx))
        for name in ("tid", "pos", "alen", "flags", "nblk"):
            assert np.array_equal(getattr(got, name), getattr(whole, name)[idx]), name
        runs = np.concatenate([np.arange(off[i], off[i + 1]) for i in idx]) if len(idx) else np.zeros(0, np.int64)
        assert np.array_equal(got.blk_start, whole.blk_start[runs]) and np.array_equal(got.blk_len, whole.blk_len[runs])
        assert got.mapped == whole.mapped
    # no index: a clear error
    os.remove(path + ".bai")
    with pytest.raises(ValueError, match="index"):
        read_bam(path, regions=[(reads.references[0], 0, 10)])


def test_damaged_index_is_rejected_not_crashed(tmp_path):
    """A BAI file with flipped bytes, absurd counts or a cut end either still resolves or raises;
    virtual offsets that point nowhere are caught when the members are read."""
    import struct
    rng = np.random.default_rng(9)
    genome, tx, reads, _ = synth.make_config("C4", scale=0.00002, tx_scale=0.001)
    path = str(tmp_path / "d.bam")
    bam_writer.write_bam(path, list(reads.references), list(reads.lengths), bam_writer.packed_to_records(reads),
                         block_bytes=2000, index=True)
    good = open(path + ".bai", "rb").read()
    regs = [(reads.references[int(reads.tid[i])], max(0, int(reads.pos[i]) - 100), int(reads.pos[i]) + 100)
            for i in rng.integers(0, reads.n, 8)]
    outcomes = {"ok": 0, "rejected": 0}
    for it in range(120):
        b = bytearray(good)
        mode = it % 3
        if mode == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif mode == 1:
            b = b[:int(rng.integers(0, len(b)))]
        else:
            i = int(rng.integers(4, len(b) - 8))
            b[i:i + 8] = struct.pack("<Q", [0, 1, 2 ** 63, 2 ** 64 - 1, 2 ** 40, 70000 << 16][int(rng.integers(0, 6))])
        with open(path + ".bai", "wb") as fh:
            fh.write(bytes(b))
        try:
            read_bam(path, threads=2, regions=regs)
            outcomes["ok"] += 1
        except (ValueError, OSError):
            outcomes["rejected"] += 1
    assert outcomes["rejected"] > 20 and outcomes["ok"] + outcomes["rejected"] == 120


@pytest.mark.parametrize("seed", range(6))
def test_chunk_chaining_under_random_geometry(tmp_path, monkeypatch, seed):
    """Whole-file loads guess a record boundary per chunk of BGZF members and chain the chunks: random member
    sizes, chunk sizes and kept-head sizes over records with sequences, every CIGAR operation, long names
    (records from 40 bytes to several KB, so that records straddle several chunks and heads) must always give
    the arrays of the packed expectation."""
    rng = np.random.default_rng(900 + seed)
    refs, lens = ["chrA", "chrB", "chrC"], [400000, 300000, 200000]
    recs = []
    for tid in range(3):
        pos = 0
        for _ in range(int(rng.integers(150, 400))):
            pos += int(rng.integers(0, 400))
            cig = []
            if rng.random() < 0.2:
                cig.append((4, int(rng.integers(1, 20))))                   # soft clip
            nrun = 1 if rng.random() < 0.7 else int(rng.integers(2, 6))
            for k in range(nrun):
                cig.append((int(rng.choice([0, 7, 8])), int(rng.integers(1, 120 if rng.random() < 0.9 else 3000))))
                if k + 1 < nrun:
                    cig.append((int(rng.choice([2, 3, 1])), int(rng.integers(1, 500))))
            if rng.random() < 0.1:
                cig.append((5, 3))                                          # hard clip
            recs.append((tid, pos, cig, int(rng.choice([0, 16]))))
    recs += [(-1, -1, [], 4)] * int(rng.integers(0, 4))
    path = str(tmp_path / "g.bam")
    placed = [r for r in recs if r[0] >= 0]
    exp = PackedAlignments.from_cigars([r[0] for r in placed], [r[1] for r in placed], [r[2] for r in placed],
                                       [bool(r[3] & 16) for r in placed], references=refs, lengths=lens)
    for trial in range(6):
        bam_writer.write_bam(path, refs, lens, recs, block_bytes=int(rng.choice([40, 333, 2000, 60000])))
        monkeypatch.setenv("PB_CHUNK", str(int(rng.choice([1, 100, 5000, 1 << 20]))))
        monkeypatch.setenv("PB_HEAD", str(int(rng.choice([0, 16, 300, 32768]))))
        got = read_bam(path, threads=int(rng.integers(1, 5)))
        for name in ("tid", "pos", "alen", "flags", "nblk", "blk_start", "blk_len"):
            assert np.array_equal(getattr(got, name), getattr(exp, name)), (name, seed, trial)
        assert got.mapped == len(placed)


def test_wide_reads_are_staged_not_refused(tmp_path):
    """Reads with more than 65 535 aligned positions or more than 255 aligned runs (the reference has no such limit:
    read.positions is a Python list, map_factories.pyx:243, 349) come out of the native reader as WIDE records --
    markers in the packed columns, true lengths / run counts aside -- for a whole file and for a region load; a read of
    exactly 65 535 bases in one run stays an ordinary record."""
    from plastid_amd.packing import PackedAlignments
    runs = [[(100, 30)], [(150, 70000)], [(200 + 3 * k, 2) for k in range(300)], [(5000, 20), (5100, 10)], [(6000, 65535)],
            [(90000, 25)]]
    pa = PackedAlignments.from_runs([0] * 6, [False, True, False, True, False, True], runs, references=["c"], lengths=[200000])
    assert list(pa.wide_idx) == [1, 2] and list(pa.wide_alen) == [70000, 600] and list(pa.wide_nblk) == [1, 300]
    assert list(pa.alen[[1, 2, 4]]) == [65535, 65535, 65535] and list(pa.nblk[[1, 2, 4]]) == [255, 255, 1]
    path = str(tmp_path / "w.bam")
    bam_writer.write_bam(path, pa.references, pa.lengths, bam_writer.packed_to_records(pa), index=True)
    for small in (False, True):
        if small:
            os.environ.update(PB_CHUNK="4096", PB_PIECE="2")
        try:
            back = read_bam(path)
        finally:
            os.environ.pop("PB_CHUNK", None); os.environ.pop("PB_PIECE", None)
        for k in ("tid", "pos", "alen", "flags", "nblk", "blk_start", "blk_len", "wide_idx", "wide_alen", "wide_nblk"):
            assert np.array_equal(getattr(back, k), getattr(pa, k)), (k, small)
        back.validate()
    assert np.array_equal(back.ref_end(), [130, 70150, 1099, 5110, 71535, 90025])
    # a region only the long read (and the 65 535-base one) reaches
    reg = read_bam(path, regions=[("c", 60000, 60010)])
    assert reg.n == 2 and list(reg.wide_idx) == [0] and list(reg.wide_alen) == [70000] and list(reg.alen) == [65535, 65535]
    sub = pa.subset([0, 2, 5])
    assert list(sub.wide_idx) == [1] and list(sub.wide_nblk) == [300] and len(sub.blk_start) == 300
    assert list(pa.slice(1, 4).wide_idx) == [0, 1]
