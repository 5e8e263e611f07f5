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
