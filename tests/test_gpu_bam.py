"""Compressed BAM on the GPU (``pc_bam_open``: BGZF inflate + BAM record decode as HIP kernels, csrc/bam_kernels.hip.h)
against the host decoder (csrc/bam_stager.cpp, itself pinned to htslib by tests/test_hts_golden.py) and against the
bytes htslib itself wrote (tests/golden/hts_fixture.npz): the packed columns must be bit-identical, the errors the
same exceptions with the same messages."""
import os
import struct
import sys
import zlib

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plastid_amd as pa  # noqa: E402
from plastid_amd import synth  # noqa: E402
from plastid_amd.bam import read_bam, read_bam_gpu  # noqa: E402
from plastid_amd.engine import Engine  # noqa: E402
from tests import bam_writer  # noqa: E402

pytestmark = pytest.mark.gpu
FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hts_fixture.npz")
COLS = ("tid", "pos", "alen", "flags", "nblk", "blk_start", "blk_len", "wide_idx", "wide_alen", "wide_nblk")


@pytest.fixture(scope="module")
def eng():
    e = Engine(0)
    yield e
    e.close()


def same(a, b):
    for k in COLS:
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    assert a.references == b.references and a.lengths == b.lengths and a.mapped == b.mapped and a.n == b.n


def member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    comp = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    cdata = comp.compress(data) + comp.flush()
    header = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, ord("B"), ord("C"), 2, len(cdata) + 25)
    return header + cdata + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


def bam_stream(refs, lens, recs, names=True):
    text = b"@HD\tVN:1.6\tSO:coordinate\n"
    out = b"BAM\x01" + struct.pack("<I", len(text)) + text + struct.pack("<I", len(refs))
    for nm, ln in zip(refs, lens):
        nmb = nm.encode() + b"\x00"
        out += struct.pack("<I", len(nmb)) + nmb + struct.pack("<I", ln)
    return out + b"".join(bam_writer.encode_record(t, p, c, f, name=(("read%07d" % i).encode() if names else b"r"))
                          for i, (t, p, c, f) in enumerate(recs))


def write_members(path, data, block_bytes, **kw):
    with open(path, "wb") as fh:
        for off in range(0, len(data), block_bytes):
            fh.write(member(data[off:off + block_bytes], **kw))
        fh.write(bam_writer.BGZF_EOF)


def test_the_htslib_written_bam(eng, tmp_path):
    hts = np.load(FIX)
    path = str(tmp_path / "htslib.bam")
    open(path, "wb").write(hts["bam"].tobytes())
    timing = {}
    got = read_bam_gpu(path, eng, timing=timing)
    same(got, read_bam(path))
    keep = np.nonzero(hts["tid"] >= 0)[0]
    assert got.n == len(keep) and np.array_equal(got.pos, hts["pos"][keep])
    assert got.mapped == int(hts["index_stat"][:, 1].sum())
    assert np.array_equal(got.ref_end(), hts["endpos"][keep])          # htslib's bam_endpos
    assert timing["members"] >= 1 and timing["records"] == len(hts["tid"])


@pytest.mark.parametrize("kw,block", [({"level": 6}, 20000), ({"level": 1}, 65280), ({"level": 9}, 3000), ({"level": 0}, 40000),
                                      ({"level": 6, "strategy": zlib.Z_FIXED}, 20000), ({"level": 6, "strategy": zlib.Z_HUFFMAN_ONLY}, 9000),
                                      ({"level": 6}, 300), ({"level": 0}, 65000)])
def test_deflate_block_types_and_member_geometry(eng, tmp_path, kw, block):
    """Dynamic, fixed and stored DEFLATE blocks; members of 64 KiB down to 300 bytes (records then span several
    members and most members hold no record start); unmapped-but-placed and unplaced records."""
    genome, tx, reads, _ = synth.make_config("C4", scale=0.00004, tx_scale=0.002)
    recs = bam_writer.packed_to_records(reads)
    recs.insert(10, (recs[10][0], recs[10][1], [], 4))
    recs += [(-1, -1, [], 4)] * 3
    data = bam_stream(list(reads.references), list(reads.lengths), recs)
    path = str(tmp_path / "x.bam")
    write_members(path, data, block, **kw)
    timing = {}
    got = read_bam_gpu(path, eng, timing=timing)
    same(got, read_bam(path))
    assert got.n == reads.n + 1 and got.mapped == reads.n and timing["records"] == len(recs)


def test_a_record_longer_than_many_members(eng, tmp_path):
    """One read with thousands of CIGAR operations and a long sequence (a record of > 200 kB): whole members lie
    inside it, and it is a wide record (> 255 aligned runs)."""
    cig = []
    for k in range(3000):
        cig += [(0, 20), (3, 50)]
    cig += [(0, 30)]
    recs = [(0, 10, [(0, 30)], 0), (0, 40, cig, 16), (0, 50, [(0, 28)], 0), (0, 300000, [(0, 25), (2, 3), (0, 5)], 0)]
    data = bam_stream(["c"], [1000000], recs)
    path = str(tmp_path / "long.bam")
    write_members(path, data, 20000, level=6)
    got = read_bam_gpu(path, eng)
    same(got, read_bam(path))
    assert got.n == 4 and len(got.wide_idx) == 1 and got.true_nblk()[1] == 3001


def test_realistic_records_and_the_counts_that_follow(eng, tmp_path):
    """Records as an aligner writes them (names, sequences, qualities, tags); the staged GPU-decoded file counts
    like the host-decoded one under a point rule and the center rule."""
    genome, tx, reads, _ = synth.make_config("C2", scale=0.0005, tx_scale=0.01)
    path = str(tmp_path / "real.bam")
    bam_writer.write_bam_realistic(path, reads, threads=4)
    got = read_bam_gpu(path, eng)
    ref = read_bam(path)
    same(got, ref)
    assert got.n == reads.n
    outs = []
    for aln in (got, ref):
        eng.set_alignments([aln])
        res = []
        for mapping in (("fiveprime", 12), ("center", 0)):
            synth.mapping_factory(mapping)._configure(eng)
            p = tx.plan_arrays(rows=1)
            plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
            res.append(plan.count(np.float64).copy())
            plan.close()
        outs.append(res)
    for a, b in zip(*outs):
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


@pytest.mark.parametrize("level", [1, 6, 9])
def test_symbol_decoders_and_upload_pieces_agree(eng, tmp_path, monkeypatch, level):
    """The batch decoder of the block symbols (every bit offset looked up by the lanes, then a walk), the wave-uniform
    one (PC_BGZF_SERIAL=1) and an upload cut into pieces of a few members each (PC_BAM_PIECE) give the same columns
    on records as an aligner writes them."""
    genome, tx, reads, _ = synth.make_config("C4", scale=0.0002, tx_scale=0.01)      # (spliced reads: multi-run records)
    path = str(tmp_path / "real.bam")
    bam_writer.write_bam_realistic(path, reads, threads=4, level=level)
    ref = read_bam(path)
    same(read_bam_gpu(path, eng), ref)
    monkeypatch.setenv("PC_BGZF_SERIAL", "1")
    same(read_bam_gpu(path, eng), ref)
    monkeypatch.setenv("PC_BAM_PIECE", "70000")
    same(read_bam_gpu(path, eng), ref)
    monkeypatch.delenv("PC_BGZF_SERIAL")
    same(read_bam_gpu(path, eng), ref)


def test_long_matches_and_long_distances(eng, tmp_path):
    """Streams the record tests do not make: runs of one byte (matches of 258 that overlap themselves), a period just
    short of the 32 KiB window (sources far behind the part of the window kept in LDS), incompressible bytes, and
    blocks that end in the middle of a batch of bit offsets -- as the payload of unplaced reads' names."""
    rng = np.random.default_rng(11)
    period = rng.integers(0, 256, 32000, dtype=np.uint8).tobytes()
    blobs = [b"\x00" * 70000, b"ab" * 40000, period * 3, rng.integers(0, 256, 50000, dtype=np.uint8).tobytes(),
             (b"ACGT" * 50 + rng.integers(65, 70, 37, dtype=np.uint8).tobytes()) * 600]
    refs, lens = ["chrA"], [100000]
    head = bam_stream(refs, lens, [])
    for level in (1, 6, 9):
        for block in (65000, 9000):
            data = head
            for i, blob in enumerate(blobs):
                # a placed record whose (long) read name carries the blob: l_read_name is one byte, so the blob rides in the sequence / quality fields
                l_seq = len(blob) // 2
                seq = blob[:(l_seq + 1) // 2]
                qual = blob[(l_seq + 1) // 2:(l_seq + 1) // 2 + l_seq]
                body = struct.pack("<iiBBHHHIiii", 0, 100 + i, 2, 30, 4680, 1, 0, l_seq, -1, -1, 0) + b"r\x00" + struct.pack("<I", (l_seq << 4) | 0) + seq + qual
                data += struct.pack("<I", len(body)) + body
            path = str(tmp_path / ("m%d_%d.bam" % (level, block)))
            write_members(path, data, block, level=level)
            same(read_bam_gpu(path, eng), read_bam(path))


def test_errors_are_the_host_decoders(eng, tmp_path):
    """Unsorted input, a damaged payload (CRC), a damaged DEFLATE stream, truncation, a foreign file: the same exception
    class and message as the host decoder raises."""
    def both(path):
        out = []
        for fn in (lambda: read_bam(path), lambda: read_bam_gpu(path, eng)):
            try:
                fn()
                out.append(None)
            except Exception as e:   # noqa: BLE001 -- compared below
                out.append((type(e).__name__ if not isinstance(e, (ValueError, OSError)) else "ValueError/IOError", str(e)))
        return out
    path = str(tmp_path / "e.bam")
    bam_writer.write_bam(path, ["c"], [1000], [(0, 50, [(0, 30)], 0), (0, 10, [(0, 30)], 0)])
    a, b = both(path)
    assert a is not None and "sorted" in a[1] and b is not None and b[1] == a[1]
    with pytest.raises(ValueError):
        read_bam_gpu(path, eng)
    # a placed record behind an unplaced one
    bam_writer.write_bam(path, ["c"], [1000], [(0, 5, [(0, 30)], 0), (-1, -1, [], 4), (0, 50, [(0, 30)], 0)])
    a, b = both(path)
    assert a is not None and b is not None and a[1] == b[1] and "sorted" in a[1]
    # reference id out of range, unknown CIGAR operation
    bam_writer.write_bam(path, ["c"], [1000], [(0, 5, [(0, 30)], 0), (3, 50, [(0, 30)], 0)])
    a, b = both(path)
    assert a is not None and b is not None and a[1] == b[1]
    bam_writer.write_bam(path, ["c"], [1000], [(0, 5, [(0, 30)], 0), (0, 50, [(0, 10), (11, 4), (0, 5)], 0)])
    a, b = both(path)
    assert a is not None and b is not None and a[1] == b[1] and "CIGAR" in a[1]
    # payload damaged, CRC and ISIZE kept: the inflated bytes differ from what the trailer promises
    genome, tx, reads, _ = synth.make_config("C2", scale=0.0002, tx_scale=0.01)
    data = bam_stream(list(reads.references), list(reads.lengths), bam_writer.packed_to_records(reads))
    good = b"".join(member(data[o:o + 30000]) for o in range(0, len(data), 30000)) + bam_writer.BGZF_EOF
    m0 = member(data[:30000], level=0)      # stored: flipping a payload byte leaves a valid DEFLATE stream with a wrong CRC
    bad = bytearray(m0 + good[len(member(data[:30000])):])
    bad[18 + 5 + 1000] ^= 0x40
    open(path, "wb").write(bytes(bad))
    a, b = both(path)
    assert a is not None and b is not None and "CRC" in a[1] and a[1] == b[1]
    # a damaged DEFLATE stream
    bad = bytearray(good)
    for k in range(40, 80):
        bad[k] ^= 0xa5
    open(path, "wb").write(bytes(bad))
    a, b = both(path)
    assert a is not None and b is not None and "BGZF" in a[1] and "BGZF" in b[1]   # (which of the two checks trips first is the inflater's business)
    # truncated inside the last record; not a BAM; not BGZF at all
    cut = data[:len(data) - 17]
    open(path, "wb").write(b"".join(member(cut[o:o + 30000]) for o in range(0, len(cut), 30000)) + bam_writer.BGZF_EOF)
    a, b = both(path)
    assert a is not None and b is not None and a[1] == b[1] and "truncated" in a[1]
    open(path, "wb").write(member(b"SAM\x01" + data[4:2000]) + bam_writer.BGZF_EOF)
    a, b = both(path)
    assert a is not None and b is not None and a[1] == b[1]
    open(path, "wb").write(b"this is not a BGZF file at all, just text " * 10)
    a, b = both(path)
    assert a is not None and b is not None and a[1] == b[1]


@pytest.mark.parametrize("seed", range(6))
def test_damaged_files_are_rejected_not_crashed(eng, tmp_path, seed):
    """Random damage inside well-formed BGZF blocks and to the container: the GPU decoder either loads exactly what
    the host decoder loads or rejects the file; it never hangs or takes the process down."""
    from plastid_amd.exceptions import EngineError, MalformedFileError
    genome, tx, reads, _ = synth.make_config("C4", scale=0.00001, tx_scale=0.001)
    good = bam_stream(list(reads.references), list(reads.lengths), bam_writer.packed_to_records(reads), names=False)
    head_len = good.index(b"r\x00") - 36
    rng = np.random.default_rng(100 + seed)
    path = str(tmp_path / "damaged.bam")
    for it in range(25):
        b = bytearray(good)
        mode = int(rng.integers(0, 4))
        lo = 0 if rng.random() < 0.2 else head_len
        if mode == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(lo, len(b)))] = int(rng.integers(0, 256))
        elif mode == 1:
            b = b[:int(rng.integers(lo, len(b)))]
        else:
            i = int(rng.integers(lo, len(b) - 4))
            b[i:i + 4] = struct.pack("<I", int(rng.choice([0, 1, 0x7fffffff, 0xffffffff, 0x80000000, 65536])))
        blob = b"".join(member(bytes(b[o:o + 3000])) for o in range(0, len(b), 3000)) + bam_writer.BGZF_EOF
        if mode == 3:
            blob = bytearray(blob)
            blob[int(rng.integers(0, len(blob)))] ^= 0x5a
            blob = bytes(blob[:int(rng.integers(len(blob) // 2, len(blob) + 1))])
        open(path, "wb").write(blob)
        res = []
        for fn in (lambda: read_bam(path, threads=2), lambda: read_bam_gpu(path, eng)):
            try:
                res.append(fn())
            except (ValueError, MalformedFileError, OSError, EngineError) as e:
                res.append(str(e))
        if isinstance(res[0], str) or isinstance(res[1], str):
            assert isinstance(res[0], str) and isinstance(res[1], str), (it, mode, res)
        else:
            same(res[1], res[0])


def test_bam_genome_array_decodes_on_the_gpu(tmp_path):
    """``BAMGenomeArray("x.bam", decode="gpu")`` -- what ``decode="auto"`` does for large files -- equals the host-decoded
    array: references, sum, counts of a chain under two rules."""
    genome, tx, reads, _ = synth.make_config("C2", scale=0.0005, tx_scale=0.01)
    path = str(tmp_path / "ga.bam")
    bam_writer.write_bam_realistic(path, reads, threads=4, level=6)
    a = pa.BAMGenomeArray(path, decode="host", mapping=pa.FivePrimeMapFactory(12))
    b = pa.BAMGenomeArray(path, decode="gpu", mapping=pa.FivePrimeMapFactory(12))
    assert a.chroms() == b.chroms() and a.sum() == b.sum() == reads.n
    chains = tx.chains(limit=40)
    for ga in (a, b):
        ga.set_mapping(pa.FivePrimeMapFactory(12))
    for x, y in zip(a.get_counts_batch(chains), b.get_counts_batch(chains)):
        assert np.array_equal(x, y)
    for ga in (a, b):
        ga.set_mapping(pa.CenterMapFactory(3))
    for x, y in zip(a.get_counts_batch(chains), b.get_counts_batch(chains)):
        assert np.array_equal(x.view(np.uint64), y.view(np.uint64))
    with pytest.raises(ValueError):
        pa.BAMGenomeArray(path, decode="fpga")


def _count_all_rules(eng, tx):
    out = []
    for mapping, dtype in ((("fiveprime", 12), np.int64), (("threeprime", 0), np.int64), (("center", 2), np.float64),
                           (("variable", synth.VARIABLE_OFFSETS), np.int64), (("stratified", synth.VARIABLE_OFFSETS, 25, 35), np.int64)):
        f = synth.mapping_factory(mapping)
        f._configure(eng)
        rows = getattr(f, "_numlengths", 1)
        p = tx.plan_arrays(rows=rows)
        plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], rows)
        out.append(plan.count(dtype).copy().view(np.uint64))
        plan.close()
    return out


@pytest.mark.parametrize("config,scale", [("C2", 0.0005), ("C4", 0.0003), ("C5", 0.0002)])
def test_a_bam_file_staged_without_leaving_the_device(tmp_path, monkeypatch, config, scale):
    """``Engine.add_bam`` (pc_add_alignment_bam): decode on the GPU, staging by kernels on the columns in HBM -- the counts
    under all five rules equal those of the host-decoded, host-staged file, for single-end reads, spliced reads
    (multi-run records, run stream) and mate pairs; the read-back-and-host-pass form of the same call
    (PC_BAM_STAGE_HOST=1) agrees too; a second file on top (one staged each way) counts like two host-staged files."""
    genome, tx, reads, _ = synth.make_config(config, scale=scale, tx_scale=0.01)
    path = str(tmp_path / "dev.bam")
    bam_writer.write_bam_realistic(path, reads, threads=4, level=6)
    host = read_bam(path)
    a = Engine(0)
    a.set_alignments([host])
    want = _count_all_rules(a, tx)
    b = Engine(0)
    assert b.add_bam(path) == host.mapped
    for x, y in zip(want, _count_all_rules(b, tx)):
        assert np.array_equal(x, y)
    monkeypatch.setenv("PC_BAM_STAGE_HOST", "1")
    c = Engine(0)
    assert c.add_bam(path) == host.mapped
    for x, y in zip(want, _count_all_rules(c, tx)):
        assert np.array_equal(x, y)
    monkeypatch.delenv("PC_BAM_STAGE_HOST")
    a.add_alignment_file(host, None)
    b.add_alignment_file(host, None)
    for x, y in zip(_count_all_rules(a, tx), _count_all_rules(b, tx)):
        assert np.array_equal(x, y)
    for e in (a, b, c):
        e.close()


def test_wide_and_long_reads_staged_on_the_device(eng, tmp_path):
    """A read beyond the 16-bit / 8-bit columns (3 001 aligned runs), long-span reads and unplaced reads, through
    ``add_bam``: counts equal the host path's."""
    refs, lens = ["chrA", "chrB"], [5000000, 200000]
    cig_wide = []
    for k in range(3000):
        cig_wide += [(0, 20), (3, 30)]
    cig_wide += [(0, 20)]
    recs = [(0, 100, [(0, 30)], 0), (0, 120, cig_wide, 16), (0, 150, [(0, 10), (3, 5000), (0, 25)], 0), (0, 90000, [(0, 300)], 0),
            (1, 5, [(0, 40)], 16), (1, 50, [(0, 10), (2, 1), (0, 10)], 0), (-1, -1, [], 4)]
    data = bam_stream(refs, lens, recs)
    path = str(tmp_path / "wide.bam")
    write_members(path, data, 20000, level=6)
    host = read_bam(path)
    tid = np.array([0, 0, 0, 1], np.int32)
    start = np.array([0, 80000, 150000, 0], np.int64)
    end = np.array([70000, 95000, 160000, 300], np.int64)
    strand = np.array([1, 2, 3, 3], np.uint8)
    L = end - start
    off = np.zeros(4, np.int64)
    np.cumsum(L[:-1], out=off[1:])
    outs = []
    for how in ("host", "dev"):
        e = Engine(0)
        if how == "host":
            e.set_alignments([host])
        else:
            assert e.add_bam(path) == host.mapped
        res = []
        for mapping, dtype in ((("fiveprime", 0), np.int64), (("threeprime", 3), np.int64), (("center", 0), np.float64)):
            synth.mapping_factory(mapping)._configure(e)
            plan = e.plan(tid, start, end, strand, off, np.ones(4, np.int8), L, int(L.sum()), 1)
            res.append(plan.count(dtype).copy().view(np.uint64))
            plan.close()
        outs.append(res)
        e.close()
    for x, y in zip(*outs):
        assert x.sum() > 0 and np.array_equal(x, y)


@pytest.mark.parametrize("seed", range(3))
def test_member_walk_by_all_host_threads(eng, tmp_path, monkeypatch, seed):
    """Large files have their BGZF member boundaries walked by every host thread from a guessed start (chained by
    where each stretch lands): forced here for small files -- same columns, and damaged files are rejected with the
    same messages as by the serial walk."""
    genome, tx, reads, _ = synth.make_config("C4", scale=0.0002, tx_scale=0.01, seed_shift=seed)
    path = str(tmp_path / "walk.bam")
    bam_writer.write_bam_realistic(path, reads, threads=4, level=6, block_bytes=3000 + 700 * seed)
    ref = read_bam(path)
    monkeypatch.setenv("PC_BAM_WALK_MIN", "1")
    same(read_bam_gpu(path, eng), ref)
    raw = bytearray(open(path, "rb").read())
    rng = np.random.default_rng(seed)
    for trial in range(12):
        bad = bytearray(raw)
        kind = trial % 3
        at = int(rng.integers(100, len(bad) - 100))
        if kind == 0:
            bad[at] ^= 0xff                                   # a flipped byte (header or payload)
        elif kind == 1:
            del bad[at:]                                      # truncated
        else:
            bad[at:at] = b"\x1f\x8b\x08\x04" + bytes(rng.integers(0, 256, 40, dtype=np.uint8))   # a false member start inside
        p2 = str(tmp_path / ("bad%d.bam" % trial))
        open(p2, "wb").write(bytes(bad))
        outcomes = []
        for walk_min in ("1", "1000000000"):
            monkeypatch.setenv("PC_BAM_WALK_MIN", walk_min)
            try:
                got = read_bam_gpu(p2, eng)
                outcomes.append(("ok", got.n, int(got.pos.sum()) if got.n else 0))
            except (ValueError, IOError) as e:
                outcomes.append(("error", str(e)))
        assert outcomes[0] == outcomes[1], outcomes


def test_files_without_placed_reads_staged_on_the_device(tmp_path):
    """A BAM file with a header only, and one with unplaced reads only, through ``add_bam``: staged as files of zero
    records, counted as zeros; a file with records staged after it counts as if alone."""
    refs, lens = ["chrA", "chrB"], [100000, 50000]
    tid = np.array([0, 1], np.int32)
    start = np.array([0, 0], np.int64)
    end = np.array([2000, 1000], np.int64)
    strand = np.array([3, 3], np.uint8)
    L = end - start
    off = np.array([0, 2000], np.int64)
    for name, recs, mapped in (("empty", [], 0), ("unplaced", [(-1, -1, [], 4), (-1, -1, [], 4)], 0)):
        path = str(tmp_path / (name + ".bam"))
        write_members(path, bam_stream(refs, lens, recs), 20000, level=6)
        e = Engine(0)
        assert e.add_bam(path) == mapped
        synth.mapping_factory(("fiveprime", 0))._configure(e)
        plan = e.plan(tid, start, end, strand, off, np.ones(2, np.int8), L, int(L.sum()), 1)
        assert plan.count(np.int64).sum() == 0
        plan.close()
        more = str(tmp_path / (name + "_more.bam"))
        write_members(more, bam_stream(refs, lens, [(0, 100, [(0, 30)], 0), (1, 7, [(0, 25)], 16)]), 20000, level=6)
        assert e.add_bam(more) == 2
        plan = e.plan(tid, start, end, strand, off, np.ones(2, np.int8), L, int(L.sum()), 1)
        got = plan.count(np.int64).copy()
        plan.close()
        h = Engine(0)
        h.set_alignments([read_bam(path), read_bam(more)])
        synth.mapping_factory(("fiveprime", 0))._configure(h)
        plan = h.plan(tid, start, end, strand, off, np.ones(2, np.int8), L, int(L.sum()), 1)
        assert got.sum() == 2 and np.array_equal(got, plan.count(np.int64))
        plan.close()
        e.close()
        h.close()


def test_bam_genome_array_without_host_reads(tmp_path):
    """``BAMGenomeArray(path, keep_reads=False)``: the files go from bytes to staged alignments on the GPU; count
    vectors, chains, normalisation, size filters and read indices equal those of the ordinary array; what needs read
    objects says so."""
    genome, tx, reads, _ = synth.make_config("C4", scale=0.0003, tx_scale=0.01)
    path = str(tmp_path / "dev.bam")
    bam_writer.write_bam_realistic(path, reads, threads=4, level=6)
    a = pa.BAMGenomeArray(path, mapping=pa.FivePrimeMapFactory(12))
    b = pa.BAMGenomeArray(path, keep_reads=False, mapping=pa.FivePrimeMapFactory(12))
    assert a.chroms() == b.chroms() and a.sum() == b.sum() == reads.n
    chains = tx.chains(limit=30)
    segs = [c[0] for c in chains[:10]]
    for factory in (pa.FivePrimeMapFactory(12), pa.CenterMapFactory(0), pa.VariableFivePrimeMapFactory(synth.VARIABLE_OFFSETS)):
        for ga in (a, b):
            ga.set_mapping(factory)
        for x, y in zip(a.get_counts_batch(chains), b.get_counts_batch(chains)):
            assert np.array_equal(x.view(np.uint64), y.view(np.uint64))
        for s_ in segs:
            assert np.array_equal(np.asarray(a[s_]).view(np.uint64), np.asarray(b[s_]).view(np.uint64))
    for ga in (a, b):
        ga.set_mapping(pa.FivePrimeMapFactory(0))
        ga.add_filter("size", pa.SizeFilterFactory(27, 31))
        ga.set_normalize(True)
    for x, y in zip(a.get_counts_batch(chains), b.get_counts_batch(chains)):
        assert np.array_equal(x.view(np.uint64), y.view(np.uint64))
    ia, ib = a.get_reads_batch(segs, as_indices=True), b.get_reads_batch(segs, as_indices=True)
    assert len(ia) == len(ib) and all(len(p) == len(q) and all(np.array_equal(u[1], v[1]) for u, v in zip(p, q)) for p, q in zip(ia, ib))
    with pytest.raises(NotImplementedError):
        b.get_reads(segs[0])
    with pytest.raises(NotImplementedError):
        b.add_filter("mine", lambda read: True)
        b[segs[0]]
    with pytest.raises(ValueError):
        pa.BAMGenomeArray(reads, keep_reads=False)
