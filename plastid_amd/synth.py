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
