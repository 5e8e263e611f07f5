"""Minimal BAM writer (test infrastructure): BGZF blocks + BAM records per the SAM/BAM
specification v1, so the native reader can be tested without pysam/samtools."""
import struct
import zlib

import numpy as np

CIGAR_OPS = "MIDNSHP=X"


def bgzf_block(data):
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = comp.compress(data) + comp.flush()
    bsize = len(cdata) + 25
    header = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, ord("B"), ord("C"), 2, bsize)
    return header + cdata + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def reg2bin(beg, end):
    end -= 1
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return off + (beg >> shift)
    return 0


def encode_record(tid, pos, cigartuples, flag, name=b"r", seq_len=None, mapq=30):
    qlen = sum(n for op, n in cigartuples if op in (0, 1, 4, 7, 8))
    if seq_len is None:
        seq_len = qlen
    ref_len = sum(n for op, n in cigartuples if op in (0, 2, 3, 7, 8))
    name = name + b"\x00"
    cig = b"".join(struct.pack("<I", (n << 4) | op) for op, n in cigartuples)
    seq = bytes([0x11] * ((seq_len + 1) // 2))
    qual = bytes([0xff] * seq_len)
    body = struct.pack("<iiBBHHHIiii", tid, pos, len(name), mapq, reg2bin(pos, pos + max(ref_len, 1)), len(cigartuples),
                       flag, seq_len, -1, -1, 0) + name + cig + seq + qual
    return struct.pack("<I", len(body)) + body


def write_bam(path, references, lengths, records, block_bytes=60000, header_text="@HD\tVN:1.6\tSO:coordinate\n"):
    """records: iterable of (tid, pos, cigartuples, flag)."""
    text = header_text.encode()
    out = b"BAM\x01" + struct.pack("<I", len(text)) + text + struct.pack("<I", len(references))
    for nm, ln in zip(references, lengths):
        nmb = nm.encode() + b"\x00"
        out += struct.pack("<I", len(nmb)) + nmb + struct.pack("<I", ln)
    chunks = [out]
    for i, (tid, pos, cig, flag) in enumerate(records):
        chunks.append(encode_record(tid, pos, cig, flag, name=("r%d" % i).encode()))
    data = b"".join(chunks)
    with open(path, "wb") as fh:
        for off in range(0, len(data), block_bytes):
            fh.write(bgzf_block(data[off:off + block_bytes]))
        fh.write(BGZF_EOF)


def packed_to_records(packed):
    """PackedAlignments -> (tid, pos, cigartuples, flag) with N gaps between aligned runs."""
    recs = []
    for i in range(packed.n):
        runs = packed.runs_of(i)
        cig = []
        for k, (s, n) in enumerate(runs):
            if k:
                cig.append((3, s - (runs[k - 1][0] + runs[k - 1][1])))
            cig.append((0, n))
        flag = 16 if packed.flags[i] & 1 else 0
        recs.append((int(packed.tid[i]), int(packed.pos[i]), cig, flag))
    return recs
