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
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hts_fixture.npz")
COLS = ("tid", "pos", "alen", "flags", "nblk", "blk_start", "blk_len", "wide_idx", "wide_alen", "wide_nblk", "flag16", "mapq", "qlen", "nh")


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
    assert np.array_equal(got.flag16, hts["flag"][keep]) and np.array_equal(got.mapq, hts["mapq"][keep])   # FLAG / MAPQ as htslib reads them back
    assert np.array_equal(got.qlen, hts["l_qseq"][keep])
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


def multi_block_member(data, level, rng):
    """One BGZF member of SEVERAL dynamic DEFLATE blocks that start at arbitrary bit positions: the payload goes
    through one compressor in random pieces with ``flush(Z_BLOCK)`` between them (the block is completed, nothing is
    padded and no empty stored block is written -- as when zlib's symbol buffer fills up in the middle of a member)."""
    comp = zlib.compressobj(level, zlib.DEFLATED, -15)
    cdata, o = [], 0
    while o < len(data):
        k = int(rng.integers(400, 6000))
        cdata.append(comp.compress(data[o:o + k]))
        o += k
        if o < len(data):
            cdata.append(comp.flush(zlib.Z_BLOCK))
    cdata = b"".join(cdata) + comp.flush()
    assert len(cdata) + 25 < 65536
    header = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, ord("B"), ord("C"), 2, len(cdata) + 25)
    return header + cdata + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


@pytest.mark.parametrize("level", [6, 9])
def test_thousands_of_multi_block_members(eng, tmp_path, level):
    """Members of 64 KiB with a dozen or two DEFLATE blocks each, by the thousand: a block header that ends just behind
    a ring refill of the wave-uniform reader (whose bit buffer then still holds bits of bytes the refill has
    overwritten in LDS) hands the batch decoder a bit position inside those bytes -- about one block header in a
    hundred.  The reader keeps eight bytes of history in the ring for exactly that (round 4's reader did not, and
    failed this test on dozens of members)."""
    genome, tx, reads, _ = synth.make_config("C2", scale=0.011, tx_scale=0.01)
    whole = str(tmp_path / "whole.bam")
    bam_writer.write_bam_realistic(whole, reads, threads=8, level=1, block_bytes=65280)
    data = b"".join(zlib.decompressobj(-15).decompress(m) for m in _member_payloads(whole))
    rng = np.random.default_rng(level)
    path = str(tmp_path / "many.bam")
    with open(path, "wb") as fh:
        for off in range(0, len(data), 65280):
            fh.write(multi_block_member(data[off:off + 65280], level, rng))
        fh.write(bam_writer.BGZF_EOF)
    timing = {}
    got = read_bam_gpu(path, eng, timing=timing)
    assert timing["members"] >= 2000
    same(got, read_bam(path))


def _member_payloads(path):
    """The raw DEFLATE streams of the BGZF members of a file."""
    raw = open(path, "rb").read()
    off = 0
    while off < len(raw):
        bsize = struct.unpack("<H", raw[off + 16:off + 18])[0] + 1
        if bsize > 28:
            yield raw[off + 18:off + bsize - 8]
        off += bsize


class _Bits(object):
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value, nbits):            # LSB first (header fields, extra bits)
        self.acc |= value << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def code(self, c, nbits):               # Huffman codes go MSB first
        self.put(int(format(c, "0%db" % nbits)[::-1], 2), nbits)

    def done(self):
        if self.n:
            self.out.append(self.acc & 255)
        return bytes(self.out)


def _code_lengths(freq, limit):
    """Huffman code lengths (<= limit) for the symbols with freq > 0."""
    import heapq
    f = {s: c for s, c in enumerate(freq) if c}
    while True:
        heap = [(c, s, (s,)) for s, c in f.items()]
        heapq.heapify(heap)
        depth = dict.fromkeys(f, 0)
        if len(heap) == 1:
            depth[heap[0][1]] = 1
        while len(heap) > 1:
            a, b = heapq.heappop(heap), heapq.heappop(heap)
            for s in a[2] + b[2]:
                depth[s] += 1
            heapq.heappush(heap, (a[0] + b[0], min(a[1], b[1]), a[2] + b[2]))
        if max(depth.values()) <= limit:
            return depth
        f = {s: (c + 1) // 2 for s, c in f.items()}


def _canonical(lengths):
    codes, code = {}, 0
    for ln in range(1, 16):
        for s in sorted(s for s, l in lengths.items() if l == ln):
            codes[s] = (code, ln)
            code += 1
        code <<= 1
    return codes


def literal_only_dynamic_member(data):
    """One BGZF member whose payload is ONE dynamic DEFLATE block of literals only, declaring HDIST = 1 distance code
    of length ZERO (what libdeflate before 1.15 wrote for such blocks; zlib's inflate_table accepts `max == 0`)."""
    freq = [0] * 257
    for b in data:
        freq[b] += 1
    freq[256] = 1
    ll = _code_lengths(freq, 15)
    lens = [ll.get(s, 0) for s in range(257)] + [0]          # 257 literal/length lengths + the one distance length
    cl_freq = [0] * 19
    for v in lens:
        cl_freq[v] += 1
    cl = _code_lengths(cl_freq, 7)
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    hclen = max(i for i, s in enumerate(order) if cl.get(s, 0)) + 1
    w = _Bits()
    w.put(1, 1); w.put(2, 2)                                   # BFINAL, dynamic
    w.put(0, 5); w.put(0, 5); w.put(max(hclen, 4) - 4, 4)      # HLIT = 257, HDIST = 1
    for s in order[:max(hclen, 4)]:
        w.put(cl.get(s, 0), 3)
    clc = _canonical(cl)
    for v in lens:
        w.code(*clc[v])
    llc = _canonical(ll)
    for b in data:
        w.code(*llc[b])
    w.code(*llc[256])
    cdata = w.done()
    assert zlib.decompressobj(-15).decompress(cdata) == data   # zlib itself takes the stream
    header = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, ord("B"), ord("C"), 2, len(cdata) + 25)
    return header + cdata + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


def test_odd_and_damaged_auxiliary_fields_on_the_device(eng, tmp_path):
    """tests/bam_writer.odd_aux_records through the GPU decoder: the same NH column as the host decoder and as expected."""
    recs, want = bam_writer.odd_aux_records()
    path = str(tmp_path / "aux.bam")
    bam_writer.write_bam(path, ["c"], [100000], recs)
    got = read_bam_gpu(path, eng)
    assert got.nh.tolist() == want
    same(got, read_bam(path))


def test_a_literal_only_block_with_an_empty_distance_code(eng, tmp_path):
    """zlib (inftrees.c: `max == 0`) and libdeflate accept a dynamic block whose one distance code has length zero;
    so does the host decoder, and so must the kernel's table builder."""
    recs = [(0, 10 + 3 * i, [(0, 30)], 16 * (i & 1)) for i in range(200)]
    data = bam_stream(["chrA"], [100000], recs)
    path = str(tmp_path / "litonly.bam")
    with open(path, "wb") as fh:
        for off in range(0, len(data), 3000):
            fh.write(literal_only_dynamic_member(data[off:off + 3000]))
        fh.write(bam_writer.BGZF_EOF)
    got = read_bam_gpu(path, eng)
    same(got, read_bam(path))
    assert got.n == 200


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
    # read OBJECTS from the records in HBM (pc_read_records): the reads the rule kept, with positions, strand, FLAG, MAPQ
    for s_ in segs[:6]:
        ra, rb = a.get_reads(s_), b.get_reads(s_)
        assert len(ra) == len(rb) and len(ra) > 0
        for x, y in zip(ra, rb):
            assert (x.index, x.reference_id, x.reference_start, x.is_reverse, x.positions, x.flag, x.mapping_quality) == \
                (y.index, y.reference_id, y.reference_start, y.is_reverse, y.positions, y.flag, y.mapping_quality)
        (r2, c2), (r1, c1) = b.get_reads_and_counts(s_), a.get_reads_and_counts(s_)
        assert np.array_equal(np.asarray(c1).view(np.uint64), np.asarray(c2).view(np.uint64)) and [r.index for r in r1] == [r.index for r in r2]
    ba, bb = a.get_reads_batch(segs), b.get_reads_batch(segs)
    assert [[r.positions for r in x] for x in ba] == [[r.positions for r in x] for x in bb]
    with pytest.raises(NotImplementedError):
        b.add_filter("mine", lambda read: True)
        b[segs[0]]
    with pytest.raises(ValueError):
        pa.BAMGenomeArray(reads, keep_reads=False)


def _with_sam_columns(reads, seed=3):
    rng = np.random.default_rng(seed)
    flag = np.where(reads.flags & 1, 0x10, 0).astype(np.uint16)
    for bit, frac in ((0x100, 0.2), (0x400, 0.15), (0x200, 0.1), (0x1, 0.5), (0x800, 0.05)):
        flag[rng.random(reads.n) < frac] |= bit
    flag[((flag & 1) != 0) & (rng.random(reads.n) < 0.6)] |= 0x2
    reads.flag16 = flag
    reads.mapq = rng.choice(np.array([0, 3, 10, 29, 30, 60, 255], np.uint8), reads.n).astype(np.uint8)
    return reads


@pytest.mark.parametrize("config,scale", [("C2", 0.0005), ("C4", 0.0003)])
def test_flag_and_mapq_filters_on_decoded_files(tmp_path, config, scale):
    """FLAG / MAPQ travel from the BAM records through both decoders to the engine: ``FlagFilterFactory`` on a file
    staged entirely on the GPU (keep_reads=False), on a GPU-decoded and on a host-decoded array give the counts of the
    reads the filter keeps (engine on the subset), under point, center and stratified rules; lifting the filter
    restores the unfiltered counts; the same decision as a plain callable on read objects agrees."""
    genome, tx, reads, _ = synth.make_config(config, scale=scale, tx_scale=0.01)
    reads = _with_sam_columns(reads)
    path = str(tmp_path / "flags.bam")
    bam_writer.write_bam_realistic(path, reads, threads=4, level=6)
    host = read_bam(path)
    assert np.array_equal(host.flag16, reads.flag16) and np.array_equal(host.mapq, reads.mapq)
    chains = tx.chains(limit=40)
    filt = pa.FlagFilterFactory(exclude=["is_secondary", "is_duplicate"], min_mapq=10)
    keep = ((reads.flag16 & 0x500) == 0) & (reads.mapq >= 10)
    assert 0.2 < keep.mean() < 0.8
    arrays = {"host": pa.BAMGenomeArray(path, decode="host"), "gpu": pa.BAMGenomeArray(path, decode="gpu"),
              "device_only": pa.BAMGenomeArray(path, keep_reads=False), "subset": pa.BAMGenomeArray(reads.subset(keep)),
              "callable": pa.BAMGenomeArray(path, decode="host")}
    for name, ga in arrays.items():
        if name == "callable":
            ga.add_filter("mine", lambda r: not r.is_secondary and not r.is_duplicate and r.mapping_quality >= 10)
        elif name != "subset":
            ga.add_filter("flags", filt)
    for factory in (pa.FivePrimeMapFactory(12), pa.CenterMapFactory(1), pa.VariableFivePrimeMapFactory(synth.VARIABLE_OFFSETS),
                    pa.StratifiedVariableFivePrimeMapFactory(synth.VARIABLE_OFFSETS, 25, 35)):
        outs = {}
        for name, ga in arrays.items():
            ga.set_mapping(factory)
            outs[name] = [np.asarray(x).view(np.uint64) for x in ga.get_counts_batch(chains)]
        assert sum(int((x != 0).sum()) for x in outs["subset"]) > 0
        for name in ("host", "gpu", "device_only", "callable"):
            assert all(np.array_equal(x, y) for x, y in zip(outs["subset"], outs[name])), (name, type(factory).__name__)
    # a second FLAG filter combines with the first; removing both restores every read
    arrays["device_only"].add_filter("pairs", pa.FlagFilterFactory(require="is_proper_pair"))
    keep2 = keep & ((reads.flag16 & 0x2) != 0)
    sub2 = pa.BAMGenomeArray(reads.subset(keep2), mapping=pa.FivePrimeMapFactory(12))
    arrays["device_only"].set_mapping(pa.FivePrimeMapFactory(12))
    assert all(np.array_equal(x, y) for x, y in zip(sub2.get_counts_batch(chains), arrays["device_only"].get_counts_batch(chains)))
    arrays["device_only"].remove_filter("pairs")
    arrays["device_only"].remove_filter("flags")
    full = pa.BAMGenomeArray(reads, mapping=pa.FivePrimeMapFactory(12))
    assert all(np.array_equal(x, y) for x, y in zip(full.get_counts_batch(chains), arrays["device_only"].get_counts_batch(chains)))


def test_flag_filter_needs_the_columns(eng):
    """A FLAG / MAPQ filter over a file staged without the two columns is refused (at the setter, or at the count when
    the file arrives later) rather than silently ignored; the columns can follow the file."""
    from plastid_amd.exceptions import EngineError
    genome, tx, reads, _ = synth.make_config("C2", scale=0.0002, tx_scale=0.01)
    e = Engine(0)
    e.set_alignments([reads])                       # synthetic arrays: no FLAG / MAPQ
    with pytest.raises(EngineError):
        e.set_flag_filter(exclude=0x100)
    e.clear_alignments()
    e.set_flag_filter(exclude=0x100, min_mapq=5)    # no file yet: accepted
    e.add_alignment_file(reads)
    synth.mapping_factory(("fiveprime", 0))._configure(e)
    p = tx.plan_arrays(rows=1)
    plan = e.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
    with pytest.raises(EngineError):
        plan.count(np.int64)
    rng = np.random.default_rng(1)
    flag = np.where(rng.random(reads.n) < 0.3, 0x100, 0).astype(np.uint16) | np.where(reads.flags & 1, 0x10, 0).astype(np.uint16)
    mapq = rng.integers(0, 11, reads.n).astype(np.uint8)
    e.set_alignment_sam(0, flag, mapq)
    got = plan.count(np.int64).copy()
    plan.close()
    keep = ((flag & 0x100) == 0) & (mapq >= 5)
    h = Engine(0)
    h.set_alignments([reads.subset(keep)])
    synth.mapping_factory(("fiveprime", 0))._configure(h)
    plan = h.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
    assert got.sum() > 0 and np.array_equal(got, plan.count(np.int64))
    plan.close()
    # the caller's own exclusions (pc_update_flags) and the FLAG filter's combine; lifting the filter leaves the caller's
    mine = rng.random(reads.n) < 0.25
    e.update_flags(0, np.where(mine, reads.flags | 0x80, reads.flags).astype(np.uint8))
    plan = e.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
    both = plan.count(np.int64).copy()
    e.set_flag_filter(enabled=False)
    only_mine = plan.count(np.int64).copy()
    plan.close()
    for want, keepmask in ((both, keep & ~mine), (only_mine, ~mine)):
        h.set_alignments([reads.subset(keepmask)])
        plan = h.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
        assert np.array_equal(want, plan.count(np.int64))
        plan.close()
    e.close()
    h.close()


# ---------------------------------------------------------------- region reads (round 5)
def test_region_reads_match_hts_itr_query_on_the_gpu(eng, tmp_path):
    """``read_bam_gpu(path, regions=...)`` (pc_bam_open_span: only the BGZF members the BAI index points to are inflated,
    the record chain starts where the index says, htslib's overlap rule on the GPU) returns what the host region reader
    returns -- itself pinned to htslib's ``sam_itr_queryi`` result sets (tests/test_hts_golden.py) -- for each of the
    fixture's 400 regions, and, record for record, htslib's own result sets."""
    hts = np.load(FIX)
    path = str(tmp_path / "htslib.bam")
    open(path, "wb").write(hts["bam"].tobytes())
    open(path + ".bai", "wb").write(hts["bai"].tobytes())
    refs = [str(x) for x in hts["references"]]
    placed = np.nonzero(hts["tid"] >= 0)[0]
    key = {(int(hts["tid"][i]), int(hts["pos"][i]), int(hts["flag"][i]), int(hts["mapq"][i]), int(hts["l_qseq"][i])): None for i in placed}
    assert len(key) > 2900           # (tid, pos, flag, mapq, l_seq) names nearly every record of the fixture
    nonempty = 0
    for q in range(len(hts["regions"])):
        t, b, e = (int(x) for x in hts["regions"][q])
        reg = [(refs[t], b, e)]
        got = read_bam_gpu(path, eng, regions=reg)
        same(got, read_bam(path, regions=reg))
        want = hts["region_records"][hts["region_off"][q]:hts["region_off"][q + 1]]
        assert got.n == len(want), (q, reg)
        nonempty += got.n > 0
        # htslib's iterator, record for record (file order)
        assert np.array_equal(got.flag16, hts["flag"][want]) and np.array_equal(got.mapq, hts["mapq"][want]) and np.array_equal(got.qlen, hts["l_qseq"][want])
    assert nonempty > 300
    many = [(refs[int(t)], int(b), int(e)) for t, b, e in hts["regions"][:40]]
    same(read_bam_gpu(path, eng, regions=many), read_bam(path, regions=many))
    same(read_bam_gpu(path, eng, regions=[("chrA", 0, 400000), ("chrB", 0, 60000), ("chrC", 0, 2000)]), read_bam(path))
    none = read_bam_gpu(path, eng, regions=[("nope", 0, 100)])
    assert none.n == 0 and list(none.references) == refs and none.mapped == int(hts["index_stat"][:, 1].sum())


def _indexed_bam(path, reads, block_bytes, rng, header_lines=0):
    recs = bam_writer.packed_to_records(reads)
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@CO\tpadding line %06d %s\n" % (k, "x" * 60) for k in range(header_lines))
    bam_writer.write_bam(path, list(reads.references), [int(x) for x in reads.lengths], recs, block_bytes=block_bytes, header_text=text, index=True)
    return recs


@pytest.mark.parametrize("block_bytes,header_lines", [(3000, 0), (20000, 0), (2500, 9000)])
def test_region_reads_over_many_members(eng, tmp_path, monkeypatch, block_bytes, header_lines):
    """Spans of hundreds of members, chunks that start and end in the middle of members, records that straddle members, a
    header of a dozen members (longer than the first slice of the file's head that is searched for it): random region
    sets through the GPU entry give the host region reader's arrays; staged on the device (``Engine.add_bam(path,
    regions=...)``) they count like the host-staged ones, center rule included."""
    genome, tx, reads, _ = synth.make_config("C4", scale=0.00006, tx_scale=0.002)
    path = str(tmp_path / "idx.bam")
    rng = np.random.default_rng(block_bytes)
    if header_lines:
        monkeypatch.setenv("PC_BAM_HEADER_BYTES", "600")     # the first slice searched for the header ends inside it: the reader comes back for more
    _indexed_bam(path, reads, block_bytes, rng, header_lines)
    refs, lens = list(reads.references), [int(x) for x in reads.lengths]
    same(read_bam_gpu(path, eng), read_bam(path))
    for trial in range(12):
        regs = []
        for _ in range(int(rng.integers(1, 6))):
            t = int(rng.integers(0, len(refs)))
            b = int(rng.integers(0, max(lens[t] - 10, 1)))
            regs.append((refs[t], b, b + int(rng.choice([1, 50, 3000, 200000, lens[t]]))))
        got, want = read_bam_gpu(path, eng, regions=regs), read_bam(path, regions=regs)
        same(got, want)
    # staged without leaving the device: counts inside the regions equal those of the host-staged region read
    chains = tx.chains(limit=60)
    regs = [(c.chrom, c.spanning_segment.start, c.spanning_segment.end) for c in chains[:25]]
    host = read_bam(path, regions=regs)
    a, b = Engine(0), Engine(0)
    a.set_alignments([host])
    kept = b.add_bam(path, regions=regs)
    assert kept == int(((host.flag16 & 4) == 0).sum()) and b.num_records(0) == host.n
    sub = tx.subset(np.arange(25))
    for x, y in zip(_count_all_rules(a, sub), _count_all_rules(b, sub)):
        assert np.array_equal(x, y)
    a.close()
    b.close()


def test_a_foreign_index_is_refused(eng, tmp_path):
    """An index that does not belong to the file (chunks beyond its end, or ending inside records) is an error, not a
    crash; a file without an index says so."""
    genome, tx, reads, _ = synth.make_config("C2", scale=0.00005, tx_scale=0.002)
    path, other = str(tmp_path / "a.bam"), str(tmp_path / "b.bam")
    rng = np.random.default_rng(0)
    _indexed_bam(path, reads, 3000, rng)
    _indexed_bam(other, reads.slice(57, reads.n // 2), 7000, rng)     # another stream: other records, other member boundaries
    with pytest.raises((ValueError, IOError)):
        read_bam_gpu(str(tmp_path / "missing.bam"), eng, regions=[("chr1", 0, 10)])
    os.replace(path + ".bai", other + ".bai")          # b.bam now has a.bam's index: offsets beyond / inside its members and records
    ref0 = reads.references[0]
    outcomes = []
    for t in range(len(reads.references)):
        on = reads.pos[reads.tid == t]
        if not len(on):
            continue
        for b in (int(np.quantile(on, f)) for f in (0.1, 0.5, 0.9)):   # where a.bam has reads: its index names chunks there
            try:
                got = read_bam_gpu(other, eng, regions=[(reads.references[t], b, b + 50000)])
                outcomes.append("ok")
                assert got.n >= 0
            except ValueError as e:
                outcomes.append("error")
                assert any(k in str(e) for k in ("index", "BGZF", "BAM", "sorted")), str(e)
    assert outcomes.count("error") >= len(outcomes) // 2, outcomes
    with pytest.raises((ValueError, IOError)):
        read_bam_gpu(path, eng, regions=[(ref0, 0, 10)])     # a.bam has lost its index


@pytest.mark.gpu
def test_bench_one_job_from_one_bam_file_on_the_gpu(tmp_path):
    """``bench.py --gpus 1 --from-bam``: the job's records are written once as ONE indexed BAM file, the rank stages its
    genome range of it through the BAI index with the decode on the GPU (pc_add_alignment_bam_span) and passes the very
    parity gates of the generated-records run -- point rule (C2), the spliced C4 and the center rule (C3: float64 sums in
    file order, bit for bit); the sum of all counts equals that of the same job counted from generated arrays (--one-job)."""
    import json
    import subprocess
    sums = {}
    for cfg, scale in (("C2", 0.004), ("C3", 0.004), ("C4", 0.0008)):
        for mode in ("--from-bam", "--one-job"):
            detail = str(tmp_path / ("d_%s_%d.json" % (cfg, mode == "--from-bam")))
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", mode, "--config", cfg, "--scale", str(scale), "--tx-scale", "0.02",
                   "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-two-files", "--no-single-query", "--other-configs", "none",
                   "--e2e-records", "0", "--e2e-realistic-records", "0", "--parity-chains", "100", "--detail-out", detail]
            proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            assert proc.returncode == 0, proc.stderr.decode()[-3000:]
            d = json.loads([ln for ln in proc.stdout.decode().splitlines() if ln.startswith("{")][-1])
            full = json.load(open(detail))["headline"]
            assert full["parity_positions"] > 0 and d["value"] > 0
            sums[(cfg, mode)] = full["sum_of_counts_all_ranks"]
            if mode == "--from-bam":
                assert d["config"]["partition"]["from_bam"] is True and "one shared BAM file" in full["partition"]["source"]
        assert sums[(cfg, "--from-bam")] == sums[(cfg, "--one-job")], cfg
