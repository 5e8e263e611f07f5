"""Seeded synthetic workloads of the BASELINE configs.

Everything is generated with ``numpy.random.default_rng(seed)`` (PCG64) and
vectorised so that 10^8 records can be produced on the host in about a minute.
Used by ``bench.py`` and by the parity tests; contains no counting logic.

* genomes: yeast-scale (17 contigs, sacCer3-like lengths, 12.16 Mb) and
  human-scale (25 contigs, hg38-like lengths, 3.1 Gb);
* transcripts: log-normal spliced lengths, uniform placement, overlaps allowed;
* reads: 90 % drawn inside transcripts (5' end uniform in spliced coordinates,
  projected to the genome, so junction-spanning reads get N gaps), 10 % uniform
  background; footprint-like aligned lengths 25..34; 1 % of the records carry a
  1-nt deletion; records sorted by (tid, pos), stable.
"""
import os

import numpy as np

from .packing import FLAG_REVERSE, PackedAlignments

YEAST = (
    ["chrI", "chrII", "chrIII", "chrIV", "chrV", "chrVI", "chrVII", "chrVIII", "chrIX", "chrX", "chrXI",
     "chrXII", "chrXIII", "chrXIV", "chrXV", "chrXVI", "chrM"],
    [230218, 813184, 316620, 1531933, 576874, 270161, 1090940, 562643, 439888, 745751, 666816, 1078177,
     924431, 784333, 1091291, 948066, 85779])

HUMAN = (
    ["chr%s" % x for x in list(range(1, 23)) + ["X", "Y", "M"]],
    [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
     133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285,
     58617616, 64444167, 46709983, 50818468, 156040895, 57227415, 16569])

FOOTPRINT_LENGTHS = np.arange(25, 35)
FOOTPRINT_PMF = np.array([1, 2, 4, 10, 20, 22, 18, 10, 6, 3], float)
FOOTPRINT_PMF /= FOOTPRINT_PMF.sum()

#: the reference's own expected offset table (test_argparsers.py:75-83)
VARIABLE_OFFSETS = {26: 12, 27: 12, 28: 13, 29: 13, 30: 14, 31: 13, "default": 13}


from .annotation import IntervalTable


class Transcripts(IntervalTable):
    """Synthetic annotation: an :class:`~plastid_amd.annotation.IntervalTable`."""


def _place(rng, lengths, span):
    """Uniform placement of features with genomic `span` on contigs chosen in
    proportion to the room they offer."""
    lengths = np.asarray(lengths, np.int64)
    n = len(span)
    tid = np.empty(n, np.int64)
    start = np.empty(n, np.int64)
    todo = np.arange(n)
    p = lengths / lengths.sum()
    while len(todo):
        t = rng.choice(len(lengths), size=len(todo), p=p)
        room = lengths[t] - span[todo]
        ok = room > 0
        tid[todo[ok]] = t[ok]
        start[todo[ok]] = (rng.random(ok.sum()) * room[ok]).astype(np.int64)
        todo = todo[~ok]
    return tid, start


def make_transcripts(genome, n, seed, style="yeast"):
    """Synthetic annotation.  `style` = ``"yeast"`` (log-normal
    spliced length, 5 % two-exon) or ``"human"`` (geometric exon counts)."""
    names, lengths = genome
    rng = np.random.default_rng(seed)
    if style == "yeast":
        tx_len = np.clip(np.exp(rng.normal(np.log(1400.0), 0.6, n)), 200, 15000).astype(np.int64)
        two = rng.random(n) < 0.05
        nex = np.where(two, 2, 1).astype(np.int64)
        ex_off = np.zeros(n + 1, np.int64)
        np.cumsum(nex, out=ex_off[1:])
        cut = (rng.random(n) * (tx_len - 60)).astype(np.int64) + 30
        intron = rng.integers(100, 501, n)
        ex_len = np.empty(ex_off[-1], np.int64)
        gap = np.zeros(ex_off[-1], np.int64)  # intron before this exon
        first = ex_off[:-1]
        ex_len[first] = np.where(two, cut, tx_len)
        second = first[two] + 1
        ex_len[second] = (tx_len - cut)[two]
        gap[second] = intron[two]
    elif style == "human":
        nex = np.minimum(rng.geometric(1.0 / 8.0, n), 60).astype(np.int64)
        ex_off = np.zeros(n + 1, np.int64)
        np.cumsum(nex, out=ex_off[1:])
        m = int(ex_off[-1])
        ex_len = np.clip(np.exp(rng.normal(np.log(150.0), 0.7, m)), 30, 5000).astype(np.int64)
        gap = np.clip(np.exp(rng.normal(np.log(1500.0), 1.2, m)), 70, 200000).astype(np.int64)
        gap[ex_off[:-1]] = 0
    else:
        raise ValueError(style)
    # genomic offsets of exons relative to the transcript start
    rel_end = np.cumsum(ex_len + gap)
    rel_end_tx0 = np.concatenate([[0], rel_end])[ex_off[:-1]]
    ex_tx = np.repeat(np.arange(n), np.diff(ex_off))
    rel_e = rel_end - rel_end_tx0[ex_tx]
    rel_s = rel_e - ex_len
    span = rel_e[ex_off[1:] - 1]
    tid, start = _place(rng, lengths, span)
    strand = np.where(rng.random(n) < 0.5, 1, 2)
    return Transcripts(names, lengths, tid, strand, ex_off, start[ex_tx] + rel_s, start[ex_tx] + rel_e)


def _project(tx, t, x, L):
    """Project spliced intervals ``[x, x+L)`` of transcripts `t` onto the genome.

    Returns ``(pos, multi_idx, nblk, run_start, run_len)``: `pos` = genomic start of
    every read; reads listed in `multi_idx` cross exon junctions and have their
    ``nblk`` (>= 2) aligned runs in ``run_start/run_len`` (CSR, read after read)."""
    t = t.astype(np.int64)
    gx = tx.ex_cum[tx.ex_off[t]] + x                       # global spliced coordinate
    e = tx.ex_off[t].copy()
    m = np.nonzero((tx.ex_off[t + 1] - tx.ex_off[t]) > 1)[0]
    if len(m):
        e[m] = np.searchsorted(tx.ex_cum, gx[m], side="right") - 1  # exon holding the first base
    room = tx.ex_cum[e + 1] - gx
    pos = tx.ex_start[e] + (gx - tx.ex_cum[e])
    multi = np.nonzero(room < L)[0]
    if not len(multi):
        z = np.zeros(0, np.int64)
        return pos, multi, z, z, z
    # general walk, only for the junction-spanning reads
    k = len(multi)
    starts, lens, owner = [], [], []
    active = np.arange(k)
    cur_e = e[multi]
    cur_g = gx[multi]
    remaining = L[multi].astype(np.int64)
    while len(active):
        room_a = tx.ex_cum[cur_e[active] + 1] - cur_g[active]
        take = np.minimum(room_a, remaining[active])
        starts.append(tx.ex_start[cur_e[active]] + (cur_g[active] - tx.ex_cum[cur_e[active]]))
        lens.append(take)
        owner.append(active)
        remaining[active] -= take
        cur_g[active] += take
        cur_e[active] += 1
        active = active[remaining[active] > 0]
    owner = np.concatenate(owner)
    starts = np.concatenate(starts)
    lens = np.concatenate(lens)
    order = np.lexsort((starts, owner))                    # by read, then left to right
    owner, starts, lens = owner[order], starts[order], lens[order]
    # merge runs that are adjacent on the genome (abutting exons)
    same = (owner[1:] == owner[:-1]) & (starts[1:] == starts[:-1] + lens[:-1])
    if same.any():
        grp = np.concatenate([[0], np.cumsum(~same)])
        first = np.concatenate([[True], ~same])
        lens = np.bincount(grp, weights=lens).astype(np.int64)
        starts = starts[first]
        owner = owner[first]
    nblk = np.bincount(owner, minlength=k).astype(np.int64)
    still = nblk >= 2
    if not still.all():                                    # fully merged: a plain single-run read
        keep = still[owner]
        owner, starts, lens = owner[keep], starts[keep], lens[keep]
        multi = multi[still]
        nblk = nblk[still]
    return pos, multi, nblk, starts, lens


#: footprint-like aligned lengths 25..34 peaked at 28-30: 96-entry inverse-cdf lookup table
_FOOTPRINT_TABLE = np.repeat(FOOTPRINT_LENGTHS, [1, 2, 4, 10, 20, 22, 18, 10, 6, 3]).astype(np.int64)


def make_reads(genome, tx, n, seed, paired=False, in_tx_frac=0.9, del_frac=0.01, expr_sigma=1.5):
    """Synthetic coordinate-sorted alignment records as a
    :class:`PackedAlignments`.  `paired`: two mates per fragment (fragment length
    N(180,30) >= 60, mate length U[25,50], mate 2 on the opposite strand); the
    reference treats mates as independent records."""
    names, lengths = genome
    lengths = np.asarray(lengths, np.int64)
    rng = np.random.default_rng(seed)
    nfrag = n // 2 if paired else n
    n_in = int(round(nfrag * in_tx_frac)) if tx is not None and tx.n else 0
    n_bg = nfrag - n_in

    # single-run records accumulate as (tid, pos, L, rev); multi-run records keep CSR runs
    s_tid, s_pos, s_len, s_rev = [], [], [], []
    m_tid, m_rev, m_nblk, m_start, m_len = [], [], [], [], []

    def add(tid, rev, L, pos, multi, nblk, starts, lens):
        single = np.ones(len(tid), bool)
        single[multi] = False
        s_tid.append(tid[single]); s_pos.append(pos[single]); s_len.append(L[single]); s_rev.append(rev[single])
        if len(multi):
            m_tid.append(tid[multi]); m_rev.append(rev[multi]); m_nblk.append(nblk)
            m_start.append(starts); m_len.append(lens)

    def draw_len(k):
        if paired:
            return rng.integers(25, 51, k)
        return _FOOTPRINT_TABLE[rng.integers(0, len(_FOOTPRINT_TABLE), k)]

    if n_in:
        w = np.exp(rng.normal(0.0, expr_sigma, tx.n))
        counts = rng.multinomial(n_in, w / w.sum())
        t = np.repeat(np.arange(tx.n, dtype=np.int64), counts)   # grouped by transcript; sorted later anyway
        tlen = tx.length[t]
        rev_tx = tx.strand[t] == 2
        ttid = tx.tid[t].astype(np.int64)
        if not paired:
            L = np.minimum(draw_len(n_in), tlen)
            x = (rng.random(n_in) * (tlen - L + 1)).astype(np.int64)
            add(ttid, rev_tx, L, *_project(tx, t, x, L))
        else:
            frag = np.minimum(np.maximum(rng.normal(180.0, 30.0, n_in), 60).astype(np.int64), tlen)
            fx = (rng.random(n_in) * (tlen - frag + 1)).astype(np.int64)
            L1 = np.minimum(draw_len(n_in), frag)
            L2 = np.minimum(draw_len(n_in), frag)
            # mate 1 at the fragment's 5' end (transcript orientation), mate 2 at its 3' end
            x1 = np.where(rev_tx, fx + frag - L1, fx)
            x2 = np.where(rev_tx, fx, fx + frag - L2)
            add(ttid, rev_tx, L1, *_project(tx, t, x1, L1))
            add(ttid, ~rev_tx, L2, *_project(tx, t, x2, L2))
    if n_bg:
        k = n_bg * (2 if paired else 1)
        L = draw_len(k)
        tid, start = _place(rng, lengths, L.astype(np.int64))
        z = np.zeros(0, np.int64)
        add(tid, rng.random(k) < 0.5, L, start, z, z, z, z)

    tid = np.concatenate(s_tid); pos = np.concatenate(s_pos); L = np.concatenate(s_len); rev = np.concatenate(s_rev)
    del s_tid, s_pos, s_len, s_rev

    # 1-nt deletions in a random 1 % of the single-run records: they become two-run records
    if del_frac > 0 and len(tid):
        cand = np.nonzero((rng.random(len(tid)) < del_frac) & (L >= 4) & (pos + L + 1 <= lengths[tid]))[0]
        if len(cand):
            cut = 1 + (rng.random(len(cand)) * (L[cand] - 2)).astype(np.int64)
            m_tid.append(tid[cand]); m_rev.append(rev[cand]); m_nblk.append(np.full(len(cand), 2, np.int64))
            st = np.empty(2 * len(cand), np.int64); ln = np.empty(2 * len(cand), np.int64)
            st[0::2] = pos[cand]; st[1::2] = pos[cand] + cut + 1
            ln[0::2] = cut; ln[1::2] = L[cand] - cut
            m_start.append(st); m_len.append(ln)
            keep = np.ones(len(tid), bool)
            keep[cand] = False
            tid, pos, L, rev = tid[keep], pos[keep], L[keep], rev[keep]

    # ---- sort the single-run records: everything fits one 64-bit key, so a value sort suffices
    key = (tid << 40) | (pos << 9) | (L << 1) | rev
    del tid, pos, L, rev
    key.sort()
    ns = len(key)

    # ---- multi-run records: argsort (they are few), then merge the two sorted streams
    if m_tid:
        mt = np.concatenate(m_tid); mr = np.concatenate(m_rev); mn = np.concatenate(m_nblk)
        ms = np.concatenate(m_start); ml = np.concatenate(m_len)
        moff = np.zeros(len(mt) + 1, np.int64)
        np.cumsum(mn, out=moff[1:])
        mpos = ms[moff[:-1]]
        malen = np.add.reduceat(ml, moff[:-1])
        mkey = (mt << 40) | (mpos << 9)
        o = np.argsort(mkey, kind="stable")
        mt, mr, mn, mpos, malen, mkey = mt[o], mr[o], mn[o], mpos[o], malen[o], mkey[o]
        tot = int(mn.sum())
        dst = np.zeros(len(mn) + 1, np.int64)
        np.cumsum(mn, out=dst[1:])
        idx = np.repeat(moff[:-1][o] - dst[:-1], mn) + np.arange(tot)
        blk_start, blk_len = ms[idx], ml[idx]
        ins = np.searchsorted(key >> 9, mkey >> 9, side="left")  # a multi-run read goes before equal-pos singles
        m_dest = ins + np.arange(len(mt))
        nm = len(mt)
    else:
        nm = 0
        blk_start = blk_len = np.zeros(0, np.int64)
    nrec = ns + nm
    out_tid = np.empty(nrec, np.int32); out_pos = np.empty(nrec, np.int32)
    out_len = np.empty(nrec, np.uint16); out_flags = np.empty(nrec, np.uint8); out_nblk = np.ones(nrec, np.uint8)
    if nm:
        is_m = np.zeros(nrec, bool)
        is_m[m_dest] = True
        s_dest = np.nonzero(~is_m)[0]
    else:
        s_dest = slice(None)
    out_tid[s_dest] = key >> 40
    out_pos[s_dest] = (key >> 9) & 0x7fffffff
    out_len[s_dest] = (key >> 1) & 0xff
    out_flags[s_dest] = (key & 1) * FLAG_REVERSE
    if nm:
        out_tid[m_dest] = mt; out_pos[m_dest] = mpos; out_len[m_dest] = malen
        out_flags[m_dest] = np.where(mr, FLAG_REVERSE, 0); out_nblk[m_dest] = mn
    return PackedAlignments(out_tid, out_pos, out_len, out_flags, out_nblk, blk_start.astype(np.int32),
                            blk_len.astype(np.int32), references=names, lengths=[int(x) for x in lengths],
                            mapped=nrec, validate=nrec <= 5_000_000)



# ---------------------------------------------------------------------------------------------------------------------
# Range-addressable generator (bench.py --gpus N: every rank generates ONLY its genome range of the one job).
#
# The job is defined unit by unit -- unit t = the reads of transcript t, unit n_tx + w = the background reads of genome
# window w -- with record counts that follow deterministically from the seed (largest-remainder apportionment of the
# read total over the expression weights / window lengths) and one ``default_rng([seed, 1, unit])`` per unit.  A rank
# draws the units whose reads can fall into its range, in unit order, keeps the records that start inside the range
# and sorts them by the job's canonical order: (tid, pos), multi-run reads before single-run reads of the same start,
# single-run reads by (aligned length, strand), multi-run reads by (unit, draw order).  The union of the ranks' owned
# records IS the job (``make_reads_blocked(layout)`` generates it whole), whatever the number of ranks.
# Same distributions as ``make_reads``; not the same draws: N = 1 keeps ``make_reads``.
def _apportion(total, weights):
    """Largest-remainder apportionment of `total` over `weights` (deterministic, sums to `total`)."""
    weights = np.asarray(weights, np.float64)
    if total <= 0 or len(weights) == 0 or weights.sum() <= 0:
        return np.zeros(len(weights), np.int64)
    q = total * (weights / weights.sum())
    base = np.floor(q).astype(np.int64)
    rem = int(total - base.sum())
    if rem > 0:
        order = np.argsort(-(q - base), kind="stable")
        base[order[:rem]] += 1
    return base


class JobLayout(object):
    """Record counts per unit, linear genome coordinates and rank cuts of one blocked job."""

    def __init__(self, genome, tx, n, seed, paired=False, in_tx_frac=0.9, del_frac=0.01, expr_sigma=1.5, bg_window=1 << 16):
        names, lengths = genome
        self.genome, self.tx, self.n, self.seed, self.paired, self.del_frac = genome, tx, int(n), int(seed), bool(paired), del_frac
        self.lengths = np.asarray(lengths, np.int64)
        nfrag = n // 2 if paired else n
        n_in = int(round(nfrag * in_tx_frac)) if tx is not None and tx.n else 0
        rng = np.random.default_rng([self.seed, 0])
        w = np.exp(rng.normal(0.0, expr_sigma, tx.n)) if n_in else np.zeros(0)
        self.tx_count = _apportion(n_in, w)                                     # fragments per transcript
        wt, ws = [], []
        for t, ln in enumerate(self.lengths):
            s0 = np.arange(0, int(ln), bg_window, dtype=np.int64)
            wt.append(np.full(len(s0), t, np.int64)); ws.append(s0)
        self.win_tid = np.concatenate(wt); self.win_start = np.concatenate(ws)
        self.win_len = np.minimum(bg_window, self.lengths[self.win_tid] - self.win_start)
        self.win_count = _apportion(nfrag - n_in, self.win_len)                 # fragments per background window
        self.tid_off = np.zeros(len(self.lengths) + 1, np.int64)
        np.cumsum(self.lengths + 1, out=self.tid_off[1:])
        self.max_len = 50 if paired else int(FOOTPRINT_LENGTHS.max())
        # genomic extent of every transcript, and the largest reference span a record can have: a read that crosses
        # the longest intron chain of its transcript end to end is bounded by aligned length + deletion + the introns
        # inside any max_len-long stretch of spliced coordinates -- bounded here by the transcript's own extent
        self.tx_lo = tx.ex_start[tx.ex_off[:-1]].astype(np.int64) if tx.n else np.zeros(0, np.int64)
        self.tx_hi = tx.ex_end[tx.ex_off[1:] - 1].astype(np.int64) if tx.n else np.zeros(0, np.int64)
        gap_max = 0
        if tx.n:
            gaps = tx.ex_start[1:] - tx.ex_end[:-1]
            inner = np.ones(len(gaps), bool)
            inner[tx.ex_off[1:-1] - 1] = False                                  # not across transcripts
            # a read of max_len bases crosses at most max_len - 1 junctions; exons are >= 30 nt, so in practice 1 - 2
            k = max(1, min(3, self.max_len // 25))
            g = np.where(inner, gaps, 0).astype(np.int64)
            if len(g):
                run = g.copy()
                for j in range(1, k):
                    run[:-j] += np.where(np.cumsum(~inner)[j:] == np.cumsum(~inner)[:-j], g[j:], 0) if len(g) > j else 0
                gap_max = int(run.max())
        self.halo = int(self.max_len + 1 + gap_max)

    @property
    def n_units(self):
        return self.tx.n + len(self.win_tid)

    def cuts(self, world, bin_size=4096):
        """`world - 1` cut points (linear genome coordinates) at quantiles of the EXPECTED record density:
        a transcript's reads spread evenly over its genomic extent, background reads evenly over their window."""
        total = int(self.tid_off[-1])
        nb = total // bin_size + 2
        dens = np.zeros(nb + 1, np.float64)
        mult = 2.0 if self.paired else 1.0

        def spread(lo, hi, cnt):
            lo = np.asarray(lo, np.int64); hi = np.maximum(np.asarray(hi, np.int64), lo + 1)
            b0, b1 = lo // bin_size, (hi - 1) // bin_size
            per = cnt * mult / (b1 - b0 + 1)
            np.add.at(dens, b0, per)
            np.add.at(dens, b1 + 1, -per)
        if self.tx.n:
            spread(self.tid_off[self.tx.tid] + self.tx_lo, self.tid_off[self.tx.tid] + self.tx_hi, self.tx_count.astype(np.float64))
        spread(self.tid_off[self.win_tid] + self.win_start, self.tid_off[self.win_tid] + self.win_start + self.win_len,
               self.win_count.astype(np.float64))
        cum = np.cumsum(np.cumsum(dens)[:nb])
        if world <= 1 or cum[-1] <= 0:
            return np.zeros(0, np.int64)
        q = cum[-1] * np.arange(1, world) / world
        return (np.searchsorted(cum, q, side="left").astype(np.int64) + 1) * bin_size

    def rank_range(self, cuts, rank):
        """``(lo, hi)``: rank `rank` OWNS the records whose linear start lies in [lo, hi)."""
        world = len(cuts) + 1
        lo = 0 if rank == 0 else int(cuts[rank - 1])
        hi = int(self.tid_off[-1]) if rank == world - 1 else int(cuts[rank])
        return lo, hi


def _assemble(genome, key, mt, mr, mn, ms, ml):
    """Sorted single-run keys + multi-run reads (CSR runs, already in canonical order) -> PackedAlignments."""
    names, lengths = genome
    ns, nm = len(key), len(mt)
    if nm:
        moff = np.zeros(nm + 1, np.int64)
        np.cumsum(mn, out=moff[1:])
        mpos = ms[moff[:-1]]
        malen = np.add.reduceat(ml, moff[:-1])
        mkey = (mt << 40) | (mpos << 9)
        ins = np.searchsorted(key >> 9, mkey >> 9, side="left")    # a multi-run read goes before equal-pos singles
        m_dest = ins + np.arange(nm)
    nrec = ns + nm
    out_tid = np.empty(nrec, np.int32); out_pos = np.empty(nrec, np.int32)
    out_len = np.empty(nrec, np.uint16); out_flags = np.empty(nrec, np.uint8); out_nblk = np.ones(nrec, np.uint8)
    if nm:
        is_m = np.zeros(nrec, bool)
        is_m[m_dest] = True
        s_dest = np.nonzero(~is_m)[0]
    else:
        s_dest = slice(None)
    out_tid[s_dest] = key >> 40
    out_pos[s_dest] = (key >> 9) & 0x7fffffff
    out_len[s_dest] = (key >> 1) & 0xff
    out_flags[s_dest] = (key & 1) * FLAG_REVERSE
    if nm:
        out_tid[m_dest] = mt; out_pos[m_dest] = mpos; out_len[m_dest] = malen
        out_flags[m_dest] = np.where(mr, FLAG_REVERSE, 0); out_nblk[m_dest] = mn
    z = np.zeros(0, np.int32)
    return PackedAlignments(out_tid, out_pos, out_len, out_flags, out_nblk, ms.astype(np.int32) if nm else z,
                            ml.astype(np.int32) if nm else z, references=names, lengths=[int(x) for x in lengths],
                            mapped=nrec, validate=nrec <= 5_000_000)


def make_reads_blocked(layout, lo=None, hi=None, block_reads=4_000_000):
    """The records of the blocked job `layout` whose linear start lies in ``[lo, hi)`` (default: the whole job), in
    the job's canonical order.  Only the units that can reach the range are drawn."""
    tx, genome, lengths = layout.tx, layout.genome, layout.lengths
    total = int(layout.tid_off[-1])
    lo = 0 if lo is None else int(lo)
    hi = total if hi is None else int(hi)
    paired, del_frac = layout.paired, layout.del_frac
    # units that can hold a record starting in [lo, hi)
    t_sel = np.nonzero((layout.tx_count > 0) & (layout.tid_off[tx.tid] + layout.tx_hi > lo) &
                       (layout.tid_off[tx.tid] + layout.tx_lo < hi))[0] if tx.n else np.zeros(0, np.int64)
    w_lin = layout.tid_off[layout.win_tid] + layout.win_start
    w_sel = np.nonzero((layout.win_count > 0) & (w_lin + layout.win_len > lo) & (w_lin < hi))[0]
    keys, m_parts = [], []

    def keep(tid, pos):
        lin = layout.tid_off[tid] + pos
        return (lin >= lo) & (lin < hi)

    def add_block(tid, rev, L, pos, multi, nblk, starts, lens, u_del, u_cut):
        """One block of reads in unit / draw order: junction-spanning reads are multi-run already; a random
        `del_frac` of the others get a 1-nt deletion (two runs).  Filter to the range, queue."""
        n = len(tid)
        single = np.ones(n, bool)
        single[multi] = False
        dele = single & (u_del < del_frac) & (L >= 4) & (pos + L + 1 <= lengths[tid])
        single &= ~dele
        k = keep(tid[single], pos[single])
        keys.append((tid[single][k] << 40) | (pos[single][k] << 9) | (L[single][k] << 1) | rev[single][k])
        # multi-run reads of the block, in draw order: junction-spanning and deleted ones interleaved by read index
        d_idx = np.nonzero(dele)[0]
        cut = 1 + (u_cut[d_idx] * (L[d_idx] - 2)).astype(np.int64)
        all_idx = np.concatenate([multi, d_idx])
        all_n = np.concatenate([nblk, np.full(len(d_idx), 2, np.int64)])
        st = np.empty(2 * len(d_idx), np.int64); ln = np.empty(2 * len(d_idx), np.int64)
        st[0::2] = pos[d_idx]; st[1::2] = pos[d_idx] + cut + 1
        ln[0::2] = cut; ln[1::2] = L[d_idx] - cut
        all_s = np.concatenate([starts, st]); all_l = np.concatenate([lens, ln])
        if not len(all_idx):
            return
        off = np.zeros(len(all_idx) + 1, np.int64)
        np.cumsum(all_n, out=off[1:])
        o = np.argsort(all_idx, kind="stable")
        kk = keep(tid[all_idx[o]], all_s[off[:-1][o]])
        o = o[kk]
        if not len(o):
            return
        nn = all_n[o]
        dst = np.zeros(len(o) + 1, np.int64)
        np.cumsum(nn, out=dst[1:])
        idx = np.repeat(off[:-1][o] - dst[:-1], nn) + np.arange(int(dst[-1]))
        m_parts.append((tid[all_idx[o]], rev[all_idx[o]], nn, all_s[idx], all_l[idx]))

    def draw_len(rng, k):
        if paired:
            return rng.integers(25, 51, k)
        return _FOOTPRINT_TABLE[rng.integers(0, len(_FOOTPRINT_TABLE), k)]

    # ---- transcript units, a block of them at a time (the projection onto the genome is vectorised over the block)
    b0 = 0
    while b0 < len(t_sel):
        b1, acc = b0, 0
        while b1 < len(t_sel) and (acc == 0 or acc + layout.tx_count[t_sel[b1]] <= block_reads):
            acc += int(layout.tx_count[t_sel[b1]]); b1 += 1
        ts, xs, Ls, ud, uc, xs2, Ls2, ud2, uc2 = [], [], [], [], [], [], [], [], []
        for t in t_sel[b0:b1]:
            c = int(layout.tx_count[t])
            rng = np.random.default_rng([layout.seed, 1, int(t)])
            tlen = int(tx.length[t])
            if not paired:
                L = np.minimum(draw_len(rng, c), tlen)
                x = (rng.random(c) * (tlen - L + 1)).astype(np.int64)
            else:
                frag = np.minimum(np.maximum(rng.normal(180.0, 30.0, c), 60).astype(np.int64), tlen)
                fx = (rng.random(c) * (tlen - frag + 1)).astype(np.int64)
                L = np.minimum(draw_len(rng, c), frag)
                L2 = np.minimum(draw_len(rng, c), frag)
                rv = tx.strand[t] == 2
                x = fx + frag - L if rv else fx             # mate 1 at the fragment's 5' end (transcript orientation)
                x2 = fx if rv else fx + frag - L2
                xs2.append(x2); Ls2.append(L2); ud2.append(rng.random(c)); uc2.append(rng.random(c))
            ts.append(np.full(c, t, np.int64)); xs.append(x); Ls.append(L); ud.append(rng.random(c)); uc.append(rng.random(c))
        t_all = np.concatenate(ts)
        rev_tx = tx.strand[t_all] == 2
        ttid = tx.tid[t_all].astype(np.int64)
        L_all = np.concatenate(Ls)
        add_block(ttid, rev_tx, L_all, *_project(tx, t_all, np.concatenate(xs), L_all), np.concatenate(ud), np.concatenate(uc))
        if paired:
            L2_all = np.concatenate(Ls2)
            add_block(ttid, ~rev_tx, L2_all, *_project(tx, t_all, np.concatenate(xs2), L2_all), np.concatenate(ud2), np.concatenate(uc2))
        b0 = b1
    # ---- background units
    z = np.zeros(0, np.int64)
    b0 = 0
    while b0 < len(w_sel):
        b1, acc = b0, 0
        while b1 < len(w_sel) and (acc == 0 or acc + layout.win_count[w_sel[b1]] <= block_reads):
            acc += int(layout.win_count[w_sel[b1]]); b1 += 1
        tids, poss, Ls, revs, ud, uc = [], [], [], [], [], []
        for w in w_sel[b0:b1]:
            c = int(layout.win_count[w]) * (2 if paired else 1)
            rng = np.random.default_rng([layout.seed, 1, int(tx.n + w)])
            L = draw_len(rng, c).astype(np.int64)
            t = int(layout.win_tid[w])
            start = layout.win_start[w] + (rng.random(c) * layout.win_len[w]).astype(np.int64)
            start = np.maximum(0, np.minimum(start, lengths[t] - L))
            tids.append(np.full(c, t, np.int64)); poss.append(start); Ls.append(L); revs.append(rng.random(c) < 0.5)
            ud.append(rng.random(c)); uc.append(rng.random(c))
        add_block(np.concatenate(tids), np.concatenate(revs), np.concatenate(Ls), np.concatenate(poss), z, z, z, z,
                  np.concatenate(ud), np.concatenate(uc))
        b0 = b1
    key = np.concatenate(keys) if keys else np.zeros(0, np.int64)
    del keys
    key.sort()
    if m_parts:
        mt = np.concatenate([p[0] for p in m_parts]); mr = np.concatenate([p[1] for p in m_parts])
        mn = np.concatenate([p[2] for p in m_parts]); ms = np.concatenate([p[3] for p in m_parts]); ml = np.concatenate([p[4] for p in m_parts])
        moff = np.zeros(len(mt) + 1, np.int64)
        np.cumsum(mn, out=moff[1:])
        o = np.argsort((mt << 40) | (ms[moff[:-1]] << 9), kind="stable")     # ties stay in (unit, draw) order
        dst = np.zeros(len(o) + 1, np.int64)
        np.cumsum(mn[o], out=dst[1:])
        idx = np.repeat(moff[:-1][o] - dst[:-1], mn[o]) + np.arange(int(dst[-1]))
        mt, mr, mn, ms, ml = mt[o], mr[o], mn[o], ms[idx], ml[idx]
    else:
        mt = mr = mn = ms = ml = z
    return _assemble(genome, key, mt, mr, mn, ms, ml)


def job_layout(name, scale=1.0, tx_scale=None):
    """``(genome, transcripts, JobLayout, mapping)`` of BASELINE config `name` under the blocked generator."""
    gname, style, n_tx, tx_seed, n_reads, r_seed, paired, mapping = CONFIGS[name]
    genome = YEAST if gname == "yeast" else HUMAN
    n_tx = max(1, int(round(n_tx * (scale if tx_scale is None else tx_scale))))
    n_reads = max(2, int(round(n_reads * scale)))
    tx = make_transcripts(genome, n_tx, tx_seed, style)
    return genome, tx, JobLayout(genome, tx, n_reads, r_seed, paired=paired), mapping


CONFIGS = {
    # name: (genome, tx style, n_tx, tx seed, n_reads, read seed, paired, mapping)
    "C1": ("yeast", "yeast", 200, 2001, 1_000_000, 1001, False, ("fiveprime", 0)),
    "C2": ("yeast", "yeast", 20_000, 2001, 100_000_000, 1002, False, ("fiveprime", 12)),
    "C3": ("yeast", "yeast", 20_000, 2001, 100_000_000, 1003, False, ("center", 0)),
    "C4": ("human", "human", 60_000, 2004, 500_000_000, 1004, False, ("variable", VARIABLE_OFFSETS)),
    "C5": ("human", "human", 60_000, 2004, 1_000_000_000, 1005, True, ("stratified", VARIABLE_OFFSETS, 25, 35)),
}


def make_config(name, scale=1.0, tx_scale=None, seed_shift=0):
    """Build ``(genome, transcripts, packed reads, mapping)`` for BASELINE config
    `name`; `scale` shrinks the read count (and `tx_scale` the transcript count)."""
    gname, style, n_tx, tx_seed, n_reads, r_seed, paired, mapping = CONFIGS[name]
    genome = YEAST if gname == "yeast" else HUMAN
    n_tx = max(1, int(round(n_tx * (scale if tx_scale is None else tx_scale))))
    n_reads = max(2, int(round(n_reads * scale)))
    tx = make_transcripts(genome, n_tx, tx_seed, style)
    cache = os.environ.get("PC_SYNTH_CACHE")   # profiling aid: several rocprofv3 passes over the same large config
    path = None
    if cache:
        path = os.path.join(cache, "%s_%d_%d_%d.npz" % (name, n_reads, n_tx, r_seed + seed_shift))
        if os.path.exists(path):
            z = np.load(path)
            reads = PackedAlignments(z["tid"], z["pos"], z["alen"], z["flags"], z["nblk"], z["blk_start"], z["blk_len"],
                                     references=genome[0], lengths=[int(x) for x in genome[1]], mapped=len(z["tid"]),
                                     validate=False)
            return genome, tx, reads, mapping
    reads = make_reads(genome, tx, n_reads, r_seed + seed_shift, paired=paired)
    if path:
        os.makedirs(cache, exist_ok=True)
        np.savez(path, tid=reads.tid, pos=reads.pos, alen=reads.alen, flags=reads.flags, nblk=reads.nblk,
                 blk_start=reads.blk_start, blk_len=reads.blk_len)
    return genome, tx, reads, mapping


def mapping_factory(mapping):
    """``("fiveprime", 12)`` -> the map factory instance."""
    from . import map_factories as mf
    kind = mapping[0]
    if kind == "fiveprime":
        return mf.FivePrimeMapFactory(mapping[1])
    if kind == "threeprime":
        return mf.ThreePrimeMapFactory(mapping[1])
    if kind == "center":
        return mf.CenterMapFactory(mapping[1])
    if kind == "variable":
        return mf.VariableFivePrimeMapFactory(mapping[1])
    if kind == "stratified":
        return mf.StratifiedVariableFivePrimeMapFactory(mapping[1], mapping[2], mapping[3])
    raise ValueError(kind)
