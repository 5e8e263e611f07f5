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


def encode_record(tid, pos, cigartuples, flag, name=b"r", seq_len=None, mapq=30, aux=b""):
    qlen = sum(n for op, n in cigartuples if op in (0, 1, 4, 7, 8))
    if seq_len is None:
        seq_len = qlen
    ref_len = sum(n for op, n in cigartuples if op in (0, 2, 3, 7, 8))
    name = name + b"\x00"
    cig = b"".join(struct.pack("<I", (n << 4) | op) for op, n in cigartuples)
    seq = bytes([0x11] * ((seq_len + 1) // 2))
    qual = bytes([0xff] * seq_len)
    body = struct.pack("<iiBBHHHIiii", tid, pos, len(name), mapq, reg2bin(pos, pos + max(ref_len, 1)), len(cigartuples),
                       flag, seq_len, -1, -1, 0) + name + cig + seq + qual + aux
    return struct.pack("<I", len(body)) + body


def write_bam(path, references, lengths, records, block_bytes=60000, header_text="@HD\tVN:1.6\tSO:coordinate\n", index=False):
    """records: sequence of (tid, pos, cigartuples, flag) or (tid, pos, cigartuples, flag, aux bytes); `index`: also write
    ``path + ".bai"``."""
    records = list(records)
    text = header_text.encode()
    out = b"BAM\x01" + struct.pack("<I", len(text)) + text + struct.pack("<I", len(references))
    for nm, ln in zip(references, lengths):
        nmb = nm.encode() + b"\x00"
        out += struct.pack("<I", len(nmb)) + nmb + struct.pack("<I", ln)
    chunks = [out]
    for i, rec in enumerate(records):
        tid, pos, cig, flag = rec[:4]
        chunks.append(encode_record(tid, pos, cig, flag, name=("r%d" % i).encode(), aux=rec[4] if len(rec) > 4 else b""))
    data = b"".join(chunks)
    block_coff = []
    coff = 0
    with open(path, "wb") as fh:
        for off in range(0, len(data), block_bytes):
            blk = bgzf_block(data[off:off + block_bytes])
            block_coff.append(coff)
            coff += len(blk)
            fh.write(blk)
        block_coff.append(coff)          # the EOF block: virtual offset of "end of data"
        fh.write(BGZF_EOF)
    if not index:
        return
    # ---- BAI (SAM spec section 5.2): bins -> chunks of virtual offsets, 16 kb linear index, and the
    # samtools metadata pseudo-bin 37450 with the mapped / unmapped record counts
    def voff(u):
        if u >= len(data):
            return block_coff[-1] << 16
        return (block_coff[u // block_bytes] << 16) | (u % block_bytes)
    u = len(chunks[0])
    nref = len(references)
    bins = [dict() for _ in range(nref)]
    linear = [dict() for _ in range(nref)]
    meta = [[None, None, 0, 0] for _ in range(nref)]
    n_no_coor = 0
    last = (None, None)
    for (tid, pos, cig, flag), raw in zip((r[:4] for r in records), chunks[1:]):
        beg, endv = voff(u), voff(u + len(raw))
        u += len(raw)
        if tid < 0:
            n_no_coor += 1
            continue
        ref_len = sum(n for op, n in cig if op in (0, 2, 3, 7, 8))
        end = pos + max(ref_len, 1)
        b = reg2bin(pos, end)
        lst = bins[tid].setdefault(b, [])
        if last == (tid, b) and lst and lst[-1][1] == beg:
            lst[-1][1] = endv
        else:
            lst.append([beg, endv])
        last = (tid, b)
        for w in range(pos >> 14, ((end - 1) >> 14) + 1):
            linear[tid].setdefault(w, beg)
        m = meta[tid]
        m[0] = beg if m[0] is None else m[0]
        m[1] = endv
        m[2 if not (flag & 4) else 3] += 1
    with open(path + ".bai", "wb") as fh:
        fh.write(b"BAI\x01" + struct.pack("<i", nref))
        for t in range(nref):
            nb = len(bins[t]) + (1 if meta[t][0] is not None else 0)
            fh.write(struct.pack("<i", nb))
            for b in sorted(bins[t]):
                fh.write(struct.pack("<Ii", b, len(bins[t][b])))
                for cb, ce in bins[t][b]:
                    fh.write(struct.pack("<QQ", cb, ce))
            if meta[t][0] is not None:
                fh.write(struct.pack("<Ii", 37450, 2) + struct.pack("<QQQQ", meta[t][0], meta[t][1], meta[t][2], meta[t][3]))
            nint = (max(linear[t]) + 1) if linear[t] else 0
            fh.write(struct.pack("<i", nint))
            prev = 0
            for w in range(nint):
                prev = linear[t].get(w, prev)   # empty windows repeat the previous offset, as samtools writes them
                fh.write(struct.pack("<Q", prev))
        fh.write(struct.pack("<Q", n_no_coor))


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


def reg2bin_array(beg, end):
    """Vectorised :func:`reg2bin`."""
    beg = np.asarray(beg, np.int64)
    end = np.asarray(end, np.int64) - 1
    out = np.zeros(len(beg), np.int64)
    done = np.zeros(len(beg), bool)
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        hit = ~done & ((beg >> shift) == (end >> shift))
        out[hit] = off + (beg[hit] >> shift)
        done |= hit
    return out


def write_bai_packed(path, packed, rec_off, data_len, block_bytes, block_coff):
    """``path + ".bai"`` for a file written by :func:`write_bam_packed` -- the index :func:`write_bam` writes (bins ->
    chunks of virtual offsets, 16 kb linear index, the samtools pseudo-bin 37450 with the mapped / unmapped counts; SAM
    spec 5.2), built from arrays.  rec_off[k]: byte offset of record k in the uncompressed stream (n + 1 entries: the
    last one is the end of the data); block_coff: file offset of every BGZF member and of the EOF block."""
    n = packed.n
    coff = np.asarray(block_coff, np.int64)
    u = np.asarray(rec_off, np.int64)
    voff = np.where(u >= data_len, coff[-1] << 16, (coff[np.minimum(u // block_bytes, len(coff) - 1)] << 16) | (u % block_bytes)).astype(np.uint64)
    beg, endv = voff[:-1], voff[1:]
    tid = packed.tid.astype(np.int64)
    pos = packed.pos.astype(np.int64)
    end = np.maximum(packed.ref_end().astype(np.int64), pos + 1)
    rbin = reg2bin_array(pos, end).astype(np.int64)
    nref = len(packed.references)
    key = tid * 65536 + rbin
    first = np.nonzero(np.r_[True, key[1:] != key[:-1]])[0] if n else np.zeros(0, np.int64)   # a chunk: consecutive records of one (reference, bin)
    last = np.r_[first[1:], n] - 1 if n else first
    ckey, cbeg, cend = key[first], beg[first], endv[last]
    order = np.argsort(ckey, kind="stable")
    ckey, cbeg, cend = ckey[order], cbeg[order], cend[order]
    tb = np.searchsorted(tid, np.arange(nref + 1))                  # records of every reference (coordinate sorted)
    w0, w1 = pos >> 14, (end - 1) >> 14
    with open(path + ".bai", "wb") as fh:
        fh.write(b"BAI\x01" + struct.pack("<i", nref))
        for t in range(nref):
            a, b = int(tb[t]), int(tb[t + 1])
            ca, cb = np.searchsorted(ckey, [t * 65536, (t + 1) * 65536])
            bins_t, where, cnt = np.unique(ckey[ca:cb] - t * 65536, return_index=True, return_counts=True)
            fh.write(struct.pack("<i", len(bins_t) + (1 if b > a else 0)))
            for bn, at, k in zip(bins_t.tolist(), where.tolist(), cnt.tolist()):
                fh.write(struct.pack("<Ii", bn, k))
                pair = np.empty((k, 2), "<u8")
                pair[:, 0] = cbeg[ca + at:ca + at + k]
                pair[:, 1] = cend[ca + at:ca + at + k]
                fh.write(pair.tobytes())
            if b > a:
                nm = int(np.count_nonzero((packed.flags[a:b] & 0) == 0))   # (packed records are all mapped)
                fh.write(struct.pack("<Ii", 37450, 2) + struct.pack("<QQQQ", int(beg[a]), int(endv[b - 1]), nm, 0))
            if b > a:
                nint = int(w1[a:b].max()) + 1
                firstrec = np.full(nint, n, np.int64)               # first record (file order) that overlaps each 16 kb window
                ww = w0[a:b]
                at0 = np.nonzero(np.r_[True, ww[1:] != ww[:-1]])[0]
                np.minimum.at(firstrec, ww[at0], a + at0)
                wide = np.nonzero(w1[a:b] > ww)[0]
                if len(wide):
                    reps = (w1[a:b][wide] - ww[wide]).astype(np.int64)
                    rec = np.repeat(a + wide, reps)
                    win = np.repeat(ww[wide], reps) + 1 + (np.arange(int(reps.sum())) - np.repeat(np.cumsum(reps) - reps, reps))
                    np.minimum.at(firstrec, win, rec)
                lin = np.zeros(nint, "<u8")
                have = firstrec < n
                lin[have] = beg[firstrec[have]]
                # empty windows repeat the previous offset, as samtools writes them (leading ones: 0)
                idx = np.where(have, np.arange(nint), -1)
                np.maximum.accumulate(idx, out=idx)
                lin = np.where(idx >= 0, lin[np.maximum(idx, 0)], 0).astype("<u8")
                fh.write(struct.pack("<i", nint) + lin.tobytes())
            else:
                fh.write(struct.pack("<i", 0))
        fh.write(struct.pack("<Q", 0))


def write_bam_packed(path, packed, block_bytes=60000, level=1, threads=8, index=False):
    """Vectorised writer for a whole :class:`PackedAlignments` (millions of records): every record
    becomes ``<run>M`` ops joined by ``N`` gaps, name ``r``, no sequence (``l_seq`` 0).  BGZF members
    are deflated on a thread pool.  Same format as :func:`write_bam`; `index`: also ``path + ".bai"``
    (:func:`write_bai_packed`)."""
    from concurrent.futures import ThreadPoolExecutor
    n = packed.n
    text = b"@HD\tVN:1.6\tSO:coordinate\n"
    head = b"BAM\x01" + struct.pack("<I", len(text)) + text + struct.pack("<I", len(packed.references))
    for nm, ln in zip(packed.references, packed.lengths):
        nmb = nm.encode() + b"\x00"
        head += struct.pack("<I", len(nmb)) + nmb + struct.pack("<I", int(ln))
    nblk = np.maximum(packed.nblk.astype(np.int64), 0)
    ncig = np.where(nblk >= 1, 2 * nblk - 1, 0)
    size = 38 + 4 * ncig                                   # block_size field included
    H = len(head)
    if n and bool(np.all(nblk >= 1)):
        # one 42-byte row per record (fixed part + the first run's <L>M), filled column-wise; the further
        # ops of the multi-run records (N M pairs) are spliced in behind their rows afterwards
        rows = np.zeros((n, 42), np.uint8)

        def col(rel, width, values, dt):
            rows[:, rel:rel + width] = np.ascontiguousarray(values, dtype=dt).view(np.uint8).reshape(n, width)

        end = packed.ref_end().astype(np.int64)
        first_len = packed.alen.astype(np.int64)
        multi = np.nonzero(nblk >= 2)[0]
        extra = np.zeros(0, np.uint8)
        at = np.zeros(0, np.int64)
        if len(multi):
            boff = packed.block_offsets()
            first_len[multi] = packed.blk_len[boff[multi]]
            cnt = nblk[multi] - 1                                  # further runs per record
            owner = np.repeat(np.arange(len(multi)), cnt)
            j = 1 + np.arange(len(owner)) - np.repeat(np.cumsum(cnt) - cnt, cnt)
            run = boff[multi][owner] + j
            st, ln = packed.blk_start[run].astype(np.int64), packed.blk_len[run].astype(np.int64)
            prev_end = packed.blk_start[run - 1].astype(np.int64) + packed.blk_len[run - 1]
            ops = np.empty((len(run), 2), "<u4")
            ops[:, 0] = ((st - prev_end) << 4) | 3
            ops[:, 1] = ln << 4
            extra = ops.view(np.uint8).reshape(-1)
            at = np.repeat((multi + 1) * 42, 8 * cnt)
        col(0, 4, size - 4, "<u4")
        col(4, 4, packed.tid, "<i4")
        col(8, 4, packed.pos, "<i4")
        rows[:, 12] = 2
        rows[:, 13] = 30
        col(14, 2, reg2bin_array(packed.pos, end), "<u2")
        col(16, 2, ncig, "<u2")
        col(18, 2, np.where(packed.flags & 1, 16, 0), "<u2")
        col(24, 4, np.full(n, -1), "<i4")
        col(28, 4, np.full(n, -1), "<i4")
        rows[:, 36] = ord("r")
        col(38, 4, first_len << 4, "<u4")
        body = rows.reshape(-1)
        if len(extra):
            body = np.insert(body.view("<u2"), at[::2] // 2, extra.view("<u2")).view(np.uint8)   # all sizes are even
        whole = np.empty(H + len(body), np.uint8)
        whole[H:] = body
        del rows, body
    else:
        off = np.zeros(n + 1, np.int64)
        np.cumsum(size, out=off[1:])
        whole = np.zeros(H + int(off[-1]), np.uint8)
        buf = whole[H:]
        rec0 = off[:-1]

        def put(rel, width, values):
            v = np.asarray(values, np.int64)
            for k in range(width):
                buf[rec0 + rel + k] = (v >> (8 * k)) & 0xff

        end = packed.ref_end().astype(np.int64)
        put(0, 4, size - 4)
        put(4, 4, packed.tid)
        put(8, 4, packed.pos)
        put(12, 1, np.full(n, 2))                              # l_read_name ("r\0")
        put(13, 1, np.full(n, 30))
        put(14, 2, reg2bin_array(packed.pos, end))
        put(16, 2, ncig)
        put(18, 2, np.where(packed.flags & 1, 16, 0))
        put(20, 4, np.zeros(n, np.int64))                      # l_seq
        put(24, 4, np.full(n, -1) & 0xffffffff)
        put(28, 4, np.full(n, -1) & 0xffffffff)
        put(32, 4, np.zeros(n, np.int64))
        buf[rec0 + 36] = ord("r")
        # CIGAR: single-run records <L>M; multi-run records M N M ...
        single = nblk == 1
        v = (packed.alen[single].astype(np.int64) << 4) | 0
        base = rec0[single] + 38
        for k in range(4):
            buf[base + k] = (v >> (8 * k)) & 0xff
        multi = np.nonzero(nblk >= 2)[0]
        if len(multi):
            boff = packed.block_offsets()
            owner = np.repeat(multi, nblk[multi])
            j = np.arange(len(owner)) - np.repeat(np.cumsum(nblk[multi]) - nblk[multi], nblk[multi])   # run index in its record
            run = boff[owner] + j
            st, ln = packed.blk_start[run].astype(np.int64), packed.blk_len[run].astype(np.int64)
            mbase = rec0[owner] + 38 + 8 * j
            mv = (ln << 4) | 0
            for k in range(4):
                buf[mbase + k] = (mv >> (8 * k)) & 0xff
            gap = j > 0
            prev_end = np.zeros(len(owner), np.int64)
            prev_end[1:] = (st + ln)[:-1]
            gv = ((st - prev_end)[gap] << 4) | 3
            gbase = mbase[gap] - 4
            for k in range(4):
                buf[gbase + k] = (gv >> (8 * k)) & 0xff
    whole[:H] = np.frombuffer(head, np.uint8)
    data = memoryview(whole)

    def member(i):
        raw = data[i:i + block_bytes]
        comp = zlib.compressobj(level, zlib.DEFLATED, -15)
        cdata = comp.compress(raw) + comp.flush()
        return (struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, ord("B"), ord("C"), 2, len(cdata) + 25) + cdata +
                struct.pack("<II", zlib.crc32(raw) & 0xffffffff, len(raw)))

    block_coff = []
    coff = 0
    with ThreadPoolExecutor(max(1, threads)) as pool, open(path, "wb") as fh:
        for blk in pool.map(member, range(0, len(data), block_bytes)):
            block_coff.append(coff)
            coff += len(blk)
            fh.write(blk)
        block_coff.append(coff)          # the EOF block: virtual offset of "end of data"
        fh.write(BGZF_EOF)
    if index:
        rec_off = np.zeros(n + 1, np.int64)
        np.cumsum(size, out=rec_off[1:])
        write_bai_packed(path, packed, rec_off + H, len(data), block_bytes, block_coff)
    return len(data)


def write_bam_realistic(path, packed, block_bytes=60000, level=1, threads=8, seed=5):
    """Like :func:`write_bam_packed`, with records that look like an aligner's: a 23-character read name, the query
    sequence (``l_seq`` = aligned length, 4-bit packed random bases), base qualities (binned, mostly one value), and
    the tags ``NH:C:1`` and ``MD:Z:<n>`` -- about 120 bytes per 30-nt read instead of 42.  Aligned runs are joined by
    ``N`` (gap > 1) or ``D`` (gap of 1) operations.  Returns the number of inflated bytes."""
    from concurrent.futures import ThreadPoolExecutor
    n = packed.n
    rng = np.random.default_rng(seed)
    text = b"@HD\tVN:1.6\tSO:coordinate\n@PG\tID:synth\n"
    head = b"BAM\x01" + struct.pack("<I", len(text)) + text + struct.pack("<I", len(packed.references))
    for nm, ln in zip(packed.references, packed.lengths):
        nmb = nm.encode() + b"\x00"
        head += struct.pack("<I", len(nmb)) + nmb + struct.pack("<I", int(ln))
    H = len(head)
    L = packed.alen.astype(np.int64)
    nblk = np.maximum(packed.nblk.astype(np.int64), 1)
    ncig = 2 * nblk - 1
    name_len = 24                                              # "SRR0000000.%012d" + NUL
    md_len = np.where(L >= 100, 3, np.where(L >= 10, 2, 1)) + 4  # "MDZ" digits NUL
    size = 36 + name_len + 4 * ncig + (L + 1) // 2 + L + 4 + md_len        # block_size field included
    off = np.zeros(n + 1, np.int64)
    np.cumsum(size, out=off[1:])
    whole = np.zeros(H + int(off[-1]), np.uint8)
    whole[:H] = np.frombuffer(head, np.uint8)
    buf = whole[H:]
    end = packed.ref_end().astype(np.int64)
    bins = reg2bin_array(packed.pos, end)
    digits = np.frombuffer(b"0123456789", np.uint8)
    boff = packed.block_offsets() if len(packed.blk_start) else None
    # records of one (aligned length, run count) class share their layout: filled as a 2-D block, scattered by offset
    cls = L * 256 + nblk
    order = np.argsort(cls, kind="stable")
    bounds = np.nonzero(np.diff(cls[order]))[0] + 1
    for grp in np.split(order, bounds):
        if not len(grp):
            continue
        Lg, nb = int(L[grp[0]]), int(nblk[grp[0]])
        sz = int(size[grp[0]])
        k = len(grp)
        rows = np.zeros((k, sz), np.uint8)

        def col(rel, width, values, dt):
            rows[:, rel:rel + width] = np.ascontiguousarray(values, dtype=dt).view(np.uint8).reshape(k, width)
        col(0, 4, np.full(k, sz - 4), "<u4")
        col(4, 4, packed.tid[grp], "<i4")
        col(8, 4, packed.pos[grp], "<i4")
        rows[:, 12] = name_len
        rows[:, 13] = 30 if getattr(packed, "mapq", None) is None else packed.mapq[grp]
        col(14, 2, bins[grp], "<u2")
        col(16, 2, np.full(k, 2 * nb - 1), "<u2")
        # (a packed file that carries SAM FLAG words -- strand bit in step with `flags` -- and MAPQ values has them written)
        col(18, 2, np.where(packed.flags[grp] & 1, 16, 0) if getattr(packed, "flag16", None) is None else packed.flag16[grp], "<u2")
        col(20, 4, np.full(k, Lg), "<i4")
        col(24, 4, np.full(k, -1), "<i4")
        col(28, 4, np.full(k, -1), "<i4")
        # name: SRR0000000.<12 digits of the record index>
        rows[:, 36:47] = np.frombuffer(b"SRR0000000.", np.uint8)
        idx = grp.astype(np.int64)
        for d in range(12):
            rows[:, 47 + 11 - d] = digits[(idx // 10 ** d) % 10]
        at = 36 + name_len
        if nb == 1:
            col(at, 4, np.full(k, Lg << 4), "<u4")
        else:
            runs = boff[grp][:, None] + np.arange(nb)[None, :]
            st, ln = packed.blk_start[runs].astype(np.int64), packed.blk_len[runs].astype(np.int64)
            for j in range(nb):
                if j:
                    gap = st[:, j] - (st[:, j - 1] + ln[:, j - 1])
                    col(at + 8 * j - 4, 4, (gap << 4) | np.where(gap == 1, 2, 3), "<u4")
                col(at + 8 * j, 4, ln[:, j] << 4, "<u4")
        at += 4 * (2 * nb - 1)
        nseq = (Lg + 1) // 2
        base = (1 << rng.integers(0, 4, (k, 2 * nseq))).astype(np.uint8)     # A C G T = 1 2 4 8
        rows[:, at:at + nseq] = (base[:, 0::2] << 4) | base[:, 1::2]
        if Lg & 1:
            rows[:, at + nseq - 1] &= 0xf0
        at += nseq
        q = np.where(rng.random((k, Lg)) < 0.9, 37, np.where(rng.random((k, Lg)) < 0.7, 25, 11)).astype(np.uint8)
        rows[:, at:at + Lg] = q
        at += Lg
        rows[:, at:at + 4] = np.frombuffer(b"NHC\x01", np.uint8)
        at += 4
        md = ("MDZ%d" % Lg).encode() + b"\x00"
        rows[:, at:at + len(md)] = np.frombuffer(md, np.uint8)
        assert at + len(md) == sz
        dest = off[grp][:, None] + np.arange(sz)[None, :]
        buf[dest] = rows
    data = memoryview(whole)

    def member(i):
        raw = data[i:i + block_bytes]
        comp = zlib.compressobj(level, zlib.DEFLATED, -15)
        cdata = comp.compress(raw) + comp.flush()
        return (struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, ord("B"), ord("C"), 2, len(cdata) + 25) + cdata +
                struct.pack("<II", zlib.crc32(raw) & 0xffffffff, len(raw)))

    with ThreadPoolExecutor(max(1, threads)) as pool, open(path, "wb") as fh:
        for blk in pool.map(member, range(0, len(data), block_bytes)):
            fh.write(blk)
        fh.write(BGZF_EOF)
    return len(data)


def odd_aux_records():
    """Records whose auxiliary fields are unusual or damaged, and the NH value (0: none) a decoder must report for each:
    what htslib's bam_aux_get walk does -- a field it cannot size ends the walk (bam_stager.cpp aux_nh, the same on the
    device).  Returns (records for write_bam, expected nh)."""
    nh_i = lambda v: b"NHi" + struct.pack("<i", v)
    cases = [
        (b"XZZno terminator" + nh_i(3), 0),                                    # a string that runs into the end of the record
        (b"XBBs" + struct.pack("<I", 1 << 30) + b"\x01\x02" + nh_i(3), 0),        # an array longer than the record
        (b"XQq\x01" + nh_i(3), 0),                                              # an unknown type
        (b"XBBs" + struct.pack("<Ihhh", 3, -1, 2, 3) + b"XZZab\x00" + b"XFf" + struct.pack("<f", 1.5) + b"NHC\x02", 2),
        (nh_i(-5), 0),                                                          # negative: none
        (b"NHI" + struct.pack("<I", 4000000000), 65535),                        # clamped
        (b"NHZ7\x00", 0),                                                       # not an integer
        (b"XAAx" + b"NH", 0),                                                   # a tag cut off behind its name
        (b"XHH1AFF\x00" + b"XBBC" + struct.pack("<I", 0) + b"NHs" + struct.pack("<h", 300), 300),   # hex string, empty array
        (b"", 0),
        (b"NHc\x7f", 127),
    ]
    recs = [(0, 100 + 10 * i, [(0, 30)], 16 if i % 2 else 0, aux) for i, (aux, _) in enumerate(cases)]
    return recs, [nh for _, nh in cases]
