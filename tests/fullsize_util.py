"""Size-independent expectations for the point mapping rules, in plain numpy.

Every read the five-prime / three-prime rules map lands on exactly ONE reference position, the
base at aligned index ``offset`` counted from the read's 5' (3') end (map_factories.pyx:322-357,
412-447): so the count vector of a whole contig on one strand is a ``bincount`` of those
positions.  That is cheap at any size (no interval logic, no fetch), independent of both the HIP
path and the C oracle, and everything else follows from it: chain vectors are spliced slices of
the contig vectors, totals are read counts.  Test infrastructure only.
"""
import numpy as np

FLAG_REVERSE, FLAG_EXCLUDED = 0x01, 0x80


def mapped_positions(reads, kind, offset):
    """``(pos, ok)``: reference position every record maps to under FivePrime/ThreePrime(`offset`),
    and whether it maps at all (not excluded, offset < aligned length)."""
    L = reads.alen.astype(np.int64)
    rev = (reads.flags & FLAG_REVERSE) != 0
    ok = ((reads.flags & FLAG_EXCLUDED) == 0) & (offset < L)
    # aligned index, in reference order, of the base `offset` from the 5' (fiveprime) / 3' end
    from_left = (~rev) if kind == "fiveprime" else rev
    idx = np.where(from_left, offset, L - 1 - offset)
    pos = reads.pos.astype(np.int64) + idx                     # single-run records
    multi = np.nonzero((reads.nblk >= 2) & ok)[0]
    if len(multi):
        off = reads.block_offsets()[multi]
        left = idx[multi].copy()
        out = np.full(len(multi), -1, np.int64)
        for r in range(int(reads.nblk[multi].max())):
            live = (out < 0) & (reads.nblk[multi] > r)
            j = off[live] + r
            blen = reads.blk_len[j].astype(np.int64)
            here = left[live] < blen
            li = np.nonzero(live)[0]
            out[li[here]] = reads.blk_start[j[here]].astype(np.int64) + left[li[here]]
            left[li[~here]] -= blen[~here]
        assert np.all(out >= 0)
        pos[multi] = out
    return pos, ok


def contig_vectors(reads, kind, offset):
    """``{(tid, strand_code): int64 vector}`` for strand codes 1 ('+') and 2 ('-') of every contig."""
    pos, ok = mapped_positions(reads, kind, offset)
    rev = (reads.flags & FLAG_REVERSE) != 0
    bounds = reads.tid_bounds()
    out = {}
    for t, n in enumerate(reads.lengths):
        lo, hi = int(bounds[t]), int(bounds[t + 1])
        for code, sel in ((1, ~rev[lo:hi]), (2, rev[lo:hi])):
            p = pos[lo:hi][sel & ok[lo:hi]]
            p = p[(p >= 0) & (p < n)]
            out[(t, code)] = np.bincount(p, minlength=n).astype(np.int64)
    return out


def chain_vectors(tx, vectors):
    """Flat ``get_counts`` layout of every chain of `tx` (an IntervalTable), cut out of the contig
    vectors: exons spliced in order, '-' chains reversed (roitools.pyx:3259-3271)."""
    flat = np.zeros(int(tx.length.sum()), np.int64)
    base = 0
    for c in range(tx.n):
        t, code = int(tx.tid[c]), int(tx.strand[c])
        parts = [vectors[(t, code)][int(tx.ex_start[j]):int(tx.ex_end[j])] for j in range(tx.ex_off[c], tx.ex_off[c + 1])]
        v = np.concatenate(parts) if parts else np.zeros(0, np.int64)
        if code == 2:
            v = v[::-1]
        flat[base:base + len(v)] = v
        base += len(v)
    return flat


def whole_contig_plan(lengths, code):
    """Plan arrays of one segment per contig, on strand `code`, laid out back to back, left to right."""
    n = np.asarray(lengths, np.int64)
    off = np.zeros(len(n) + 1, np.int64)
    np.cumsum(n, out=off[1:])
    return dict(tid=np.arange(len(n), dtype=np.int32), start=np.zeros(len(n), np.int64), end=n.copy(),
                strand=np.full(len(n), code, np.uint8), out_off=off[:-1].copy(), out_step=np.ones(len(n), np.int8),
                row_stride=n.copy(), out_elems=int(off[-1]))


# ---------------------------------------------------------------- table rules, large genomes
def offsets_by_length(offset_dict, max_len=65536):
    """``off[L]`` of a Variable / Stratified offset dict (``'default'`` fills the gaps, -1: none)."""
    off = np.full(max_len, int(offset_dict.get("default", -1)), np.int64)
    for k, v in offset_dict.items():
        if k != "default":
            off[int(k)] = int(v)
    return off


def mapped_positions_by_table(reads, off_by_len):
    """Like :func:`mapped_positions` for a per-length 5' offset table (map_factories.pyx:584-600):
    forward reads map at aligned index ``off[L]``, reverse reads at ``L - 1 - off[L]``."""
    L = reads.alen.astype(np.int64)
    o = off_by_len[L]
    rev = (reads.flags & FLAG_REVERSE) != 0
    ok = ((reads.flags & FLAG_EXCLUDED) == 0) & (o >= 0) & (o < L)
    idx = np.where(rev, L - 1 - o, o)
    pos = reads.pos.astype(np.int64) + idx
    multi = np.nonzero((reads.nblk >= 2) & ok)[0]
    if len(multi):
        off = reads.block_offsets()[multi]
        left = idx[multi].copy()
        out = np.full(len(multi), -1, np.int64)
        for r in range(int(reads.nblk[multi].max())):
            live = (out < 0) & (reads.nblk[multi] > r)
            j = off[live] + r
            blen = reads.blk_len[j].astype(np.int64)
            here = left[live] < blen
            li = np.nonzero(live)[0]
            out[li[here]] = reads.blk_start[j[here]].astype(np.int64) + left[li[here]]
            left[li[~here]] -= blen[~here]
        pos[multi] = out
    return pos, ok


def sparse_chain_vectors(tx, reads, pos, sel, threads=16):
    """Flat ``get_counts`` layout (rows = 1) of every chain of `tx`, from the mapped positions of the
    records in boolean `sel`: per contig a sorted (position, strand) key table is sliced per exon, so
    no dense genome-sized vector is ever built (human-scale genomes).  Contigs are independent
    (records are sorted by contig) and are handled on a thread pool -- numpy sorts release the GIL --
    so that the BASELINE sizes (500 M records) stay within a test's time."""
    from concurrent.futures import ThreadPoolExecutor
    clen = np.asarray(reads.lengths, np.int64)
    rev = (reads.flags & FLAG_REVERSE) != 0
    tb = reads.tid_bounds()
    p = tx.plan_arrays(rows=1)
    flat = np.zeros(p["out_elems"], np.int64)
    ex_tx = tx.ex_tx
    ex_tid = tx.tid[ex_tx]
    ex_rev = tx.strand[ex_tx] == 2

    def contig(t):
        a, b = int(tb[t]), int(tb[t + 1])
        span = int(clen[t])
        ps, sl = pos[a:b], sel[a:b]
        inside = sl & (ps >= 0) & (ps < span)
        keys = np.where(rev[a:b][inside], span, 0) + ps[inside]
        ukeys, cnt = np.unique(keys, return_counts=True)
        ex = np.nonzero(ex_tid == t)[0]
        if not len(ex) or not len(ukeys):
            return
        sbase = np.where(ex_rev[ex], span, 0)
        lo = np.searchsorted(ukeys, sbase + tx.ex_start[ex])
        hi = np.searchsorted(ukeys, sbase + tx.ex_end[ex])
        n = hi - lo
        ex_of = np.repeat(np.arange(len(lo)), n)
        k = np.arange(int(n.sum())) - np.repeat(np.cumsum(n) - n, n) + np.repeat(lo, n)
        rel = ukeys[k] - (sbase + tx.ex_start[ex])[ex_of]
        seg = ex[ex_of]
        flat[p["out_off"][seg] + p["out_step"][seg].astype(np.int64) * rel] = cnt[k]   # disjoint output slices per contig

    with ThreadPoolExecutor(max(1, threads)) as pool:
        list(pool.map(contig, range(len(clen))))
    return flat


def sparse_chain_rows(tx, reads, pos, sel, row, rows, threads=16):
    """The stratified rule's output (``tx.plan_arrays(rows=rows)`` layout) from the mapped positions: record i
    counts in row ``row[i]`` (records outside [0, rows) count nowhere).  One sorted (row, strand, position) key
    table per contig serves every exon and every row -- a single pass over the records, whatever the number of
    rows (C5 at its full 10^9 records)."""
    from concurrent.futures import ThreadPoolExecutor
    clen = np.asarray(reads.lengths, np.int64)
    rev = (reads.flags & FLAG_REVERSE) != 0
    tb = reads.tid_bounds()
    p = tx.plan_arrays(rows=rows)
    flat = np.zeros(p["out_elems"], np.int64)
    ex_tx = tx.ex_tx
    ex_tid = tx.tid[ex_tx]
    ex_rev = tx.strand[ex_tx] == 2

    def contig(t):
        a, b = int(tb[t]), int(tb[t + 1])
        span = int(clen[t])
        ps, rw = pos[a:b], row[a:b]
        inside = sel[a:b] & (ps >= 0) & (ps < span) & (rw >= 0) & (rw < rows)
        keys = (rw[inside] * 2 + rev[a:b][inside]) * span + ps[inside]
        ukeys, cnt = np.unique(keys, return_counts=True)
        ex = np.nonzero(ex_tid == t)[0]
        if not len(ex) or not len(ukeys):
            return
        for r in range(rows):
            sbase = (r * 2 + ex_rev[ex].astype(np.int64)) * span
            lo = np.searchsorted(ukeys, sbase + tx.ex_start[ex])
            hi = np.searchsorted(ukeys, sbase + tx.ex_end[ex])
            n = hi - lo
            ex_of = np.repeat(np.arange(len(lo)), n)
            k = np.arange(int(n.sum())) - np.repeat(np.cumsum(n) - n, n) + np.repeat(lo, n)
            rel = ukeys[k] - (sbase + tx.ex_start[ex])[ex_of]
            seg = ex[ex_of]
            flat[p["out_off"][seg] + r * p["row_stride"][seg] + p["out_step"][seg].astype(np.int64) * rel] = cnt[k]

    with ThreadPoolExecutor(max(1, threads)) as pool:
        list(pool.map(contig, range(len(clen))))
    return flat
