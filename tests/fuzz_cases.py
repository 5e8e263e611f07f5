"""Randomised differential cases: HIP engine vs the oracle (helper, not collected by pytest).

Every case draws contigs, 1-3 alignment files (ungapped, gapped, long-span, empty and
host-excluded records, pile-ups), a batch of segments (all strand codes, unknown contigs,
empty and out-of-range intervals, overlapping), a mapping rule with random parameters, an
optional size filter, an output layout (forward / 5'->3' reversed / summed slices) and the
engine's scheduling knobs.  ``run_case`` raises AssertionError on the first mismatch.
"""
import os
import warnings

import numpy as np

NOFILTER = 0x10
KNOBS = ("PC_TILE_G", "PC_WORK_R", "PC_PILE", "PC_NO_SMALL", "PC_STAGE_SLICE", "PC_HIST_LAZY_BYTES", "PC_RANGES_CG1")


def random_file(rng, pa, names, lens, n, max_len, p_gapped, p_long_gap, pile):
    tid = np.sort(rng.integers(0, len(names), n)).astype(np.int32)
    pos = np.zeros(n, np.int64)
    for t in range(len(names)):
        m = tid == t
        k = int(m.sum())
        if not k:
            continue
        p = rng.integers(0, max(1, lens[t] - 10), k)
        if pile and k > 10:  # a pile-up: a third of the contig's reads inside 40 nt
            hot = rng.integers(0, max(1, lens[t] - 50))
            sel = rng.random(k) < 0.35
            p[sel] = hot + rng.integers(0, 40, int(sel.sum()))
        pos[m] = np.sort(p)
    alen = np.zeros(n, np.int64)
    nblk = np.zeros(n, np.int64)
    bs, bl = [], []
    kind = rng.random(n)
    for i in range(n):
        if kind[i] < 0.01:            # no aligned bases at all (e.g. 30S): L == 0
            continue
        if kind[i] < 0.01 + p_gapped:
            nr = int(rng.integers(2, 6))
            start = int(pos[i])
            for _ in range(nr):
                ln = int(rng.integers(1, 40))
                bs.append(start)
                bl.append(ln)
                alen[i] += ln
                gap = int(rng.integers(100, 4000)) if rng.random() < p_long_gap else int(rng.integers(1, 40))
                start += ln + gap
            nblk[i] = nr
        else:
            alen[i] = int(rng.integers(1, max_len + 1)) if rng.random() < 0.03 else int(rng.integers(15, 45))
            nblk[i] = 1
    flags = (rng.random(n) < 0.5).astype(np.uint8)
    flags[rng.random(n) < 0.03] |= 0x80
    return pa.PackedAlignments(tid, pos, alen, flags, nblk, np.array(bs, np.int32), np.array(bl, np.int32),
                               references=names, lengths=lens, validate=False)


def lengthen_last_runs(rng, pa, f):
    multi = np.nonzero(f.nblk >= 2)[0]
    if not len(multi):
        return f
    pick = multi[rng.random(len(multi)) < 0.3]
    if not len(pick):
        return f
    off = f.block_offsets()
    bl = f.blk_len.copy()
    alen = f.alen.astype(np.int64)
    for i in pick:
        extra = int(rng.integers(230, 700))
        bl[off[i] + int(f.nblk[i]) - 1] += extra
        alen[i] += extra
    return pa.PackedAlignments(f.tid, f.pos, alen, f.flags, f.nblk, f.blk_start, bl, references=f.references,
                               lengths=f.lengths, validate=False)


def random_case(seed, pa, size="small"):
    rng = np.random.default_rng(seed)
    ntid = int(rng.integers(1, 5))
    names = ["c%d" % i for i in range(ntid)]
    lens = [int(rng.integers(300, 60000 if size == "small" else 400000)) for _ in range(ntid)]
    nfiles = int(rng.choice([1, 1, 1, 2, 3]))
    max_len = int(rng.choice([44, 200, 255, 256, 2500]))
    files = []
    for _ in range(nfiles):
        n = int(rng.choice([0, 1, 2, 50, 3000, 20000 if size == "small" else 200000]))
        files.append(random_file(rng, pa, names, lens, n, max_len, float(rng.choice([0.0, 0.05, 0.4])),
                                 float(rng.choice([0.0, 0.1, 0.5])), bool(rng.random() < 0.4)))
    if seed % 3 == 1:
        # Multi-run reads longer than the run stream carries (aligned length > 255): they stay on the gapped /
        # long-span side lists.  Made from the drawn files by lengthening the LAST run of some gapped reads, with
        # a generator of its own, so that every case of earlier revisions keeps its draws.
        rng2 = np.random.default_rng(seed + 1000003)
        files = [lengthen_last_runs(rng2, pa, f) for f in files]
    nseg = int(rng.integers(1, 60))
    seg_tid = rng.integers(-1, ntid + 1, nseg).astype(np.int32)
    seg_start = np.zeros(nseg, np.int64)
    seg_end = np.zeros(nseg, np.int64)
    for s in range(nseg):
        ln = lens[seg_tid[s]] if 0 <= seg_tid[s] < ntid else 5000
        a = int(rng.integers(0, ln))
        mode = rng.random()
        if mode < 0.1:
            b = a                                   # empty segment
        elif mode < 0.3:
            b = a + int(rng.integers(1, 40))        # tiny exon
        elif mode < 0.4:
            a, b = 0, ln + int(rng.integers(0, 300))  # whole contig (and past its end)
        else:
            b = a + int(rng.integers(1, 9000))
        seg_start[s], seg_end[s] = a, b
    seg_strand = rng.choice(np.array([0, 1, 1, 2, 2, 3, 1 | NOFILTER, 2 | NOFILTER], np.uint8), nseg)
    kind = str(rng.choice(["fiveprime", "threeprime", "center", "variable", "stratified"]))
    if kind in ("fiveprime", "threeprime"):
        mapping = (kind, int(rng.choice([0, 3, 12, 30, 60])))
    elif kind == "center":
        mapping = (kind, int(rng.choice([0, 2, 12, 25])))
    else:
        has_default = bool(rng.random() < 0.7)
        # without a "default" every entry must be usable (off < L): the reference's constructor
        # trips over its own error path otherwise
        od = {int(L): int(rng.integers(0, L + 3 if has_default else L))
              for L in rng.integers(10, 60, int(rng.integers(0 if has_default else 1, 12)))}
        if has_default:
            od["default"] = int(rng.integers(0, 40))
        if max_len > 100 and rng.random() < 0.5:
            od[int(rng.integers(101, max_len + 1))] = int(rng.integers(0, 100))
        if kind == "variable":
            mapping = (kind, od)
        else:
            lo = int(rng.integers(1, 40))
            mapping = (kind, od, lo, lo + int(rng.integers(1, 12)))
    size_filter = None
    if rng.random() < 0.3:
        lo = int(rng.integers(1, 40))
        size_filter = (lo, int(rng.choice([-1, lo, lo + 10, 400])))
    knobs = {}
    if rng.random() < 0.7:
        knobs["PC_TILE_G"] = str(int(rng.choice([256, 512, 768, 1024, 4096])))
        knobs["PC_WORK_R"] = str(int(rng.choice([64, 512, 4096, 32768])))
        knobs["PC_PILE"] = str(int(rng.choice([64, 2048, 1000000])))
        if rng.random() < 0.3:
            knobs["PC_NO_SMALL"] = "1"
        rng.random()  # (keeps the stream of random numbers of earlier revisions)
    if seed % 2 == 0:   # staging in tiny slices: every slice boundary of the pipelined pass (not drawn from rng:
        knobs["PC_STAGE_SLICE"] = str((1, 2, 5, 64)[(seed // 2) % 4])   # the cases of earlier revisions stay the same)
    if seed % 3 == 1:   # the compact histogram cleared slice by slice (k_clear_split), as for plans of 64 MB and more (not drawn from rng)
        knobs["PC_HIST_LAZY_BYTES"] = "1"
    if seed % 5 == 2:   # one lane per window in k_tile_ranges, as for large plans (small plans: sixteen, sharing a cut window's sub-windows)
        knobs["PC_RANGES_CG1"] = "1"
    layout = str(rng.choice(["forward", "reversed", "mixed", "sums"]))
    if kind == "center" and layout == "sums":
        layout = "mixed"
    return dict(seed=seed, names=names, lens=lens, files=files, seg_tid=seg_tid, seg_start=seg_start, seg_end=seg_end,
                seg_strand=seg_strand, mapping=mapping, size_filter=size_filter, knobs=knobs, layout=layout,
                step_seed=int(rng.integers(0, 1 << 30)))


def oracle_expected(oracle, pa, case, spec):
    """Per-segment oracle arrays; host-excluded records are simply absent from the oracle's input,
    and the two "no strand filter" codes are expressed through codes the reference has."""
    from plastid_amd.packing import concat_file_major
    kept = [f.subset(np.nonzero((f.flags & 0x80) == 0)[0]) for f in case["files"]]
    st = case["seg_strand"]
    ostrand = np.where(st == (1 | NOFILTER), 3, st & 3).astype(np.uint8)
    arrays, warn = oracle.count_segments(concat_file_major(kept), spec, case["seg_tid"], case["seg_start"],
                                         case["seg_end"], ostrand)
    nf = np.nonzero(st == (2 | NOFILTER))[0]
    if len(nf):
        allrev = [pa.PackedAlignments(f.tid, f.pos, f.alen, f.flags | 1, f.nblk, f.blk_start, f.blk_len,
                                      references=f.references, lengths=f.lengths, validate=False) for f in kept]
        a2, w2 = oracle.count_segments(concat_file_major(allrev), spec, case["seg_tid"][nf], case["seg_start"][nf],
                                       case["seg_end"][nf], ostrand[nf])
        for k, s in enumerate(nf):
            arrays[s] = a2[k]
            warn[s] = w2[k]
    return arrays, warn


def run_case(pa, oracle, case, spec_for, engine_for):
    saved = {k: os.environ.pop(k, None) for k in KNOBS}
    os.environ.update(case["knobs"])
    try:
        mapping, sf = case["mapping"], case["size_filter"]
        rng = np.random.default_rng(case["step_seed"])
        late_flags = bool(rng.random() < 0.4)     # exclusion bits arrive after staging (pc_update_flags)
        files = case["files"]
        if late_flags:
            files = [pa.PackedAlignments(f.tid, f.pos, f.alen, f.flags & 0x7f, f.nblk, f.blk_start, f.blk_len,
                                         references=f.references, lengths=f.lengths, validate=False) for f in files]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # offset-dictionary DataWarnings of the factory constructors
            eng = engine_for(pa, files, mapping, sf)
        if late_flags:
            for fi, f in enumerate(case["files"]):
                eng.update_flags(fi, f.flags)
        rows = eng.rows
        lens_ = case["seg_end"] - case["seg_start"]
        nseg = len(lens_)
        layout = case["layout"]
        if layout == "sums":
            step = np.zeros(nseg, np.int8)
            out_off = rng.integers(0, 7, nseg).astype(np.int64) * rows     # several segments share a slot
            stride = np.ones(nseg, np.int64)
            out_elems = 7 * rows
        else:
            step = {"forward": np.ones(nseg, np.int8), "reversed": -np.ones(nseg, np.int8),
                    "mixed": rng.choice(np.array([1, -1], np.int8), nseg)}[layout]
            base = np.concatenate([[0], np.cumsum(lens_ * rows)[:-1]]).astype(np.int64)
            out_off = np.where(step > 0, base, base + lens_ - 1)
            stride = lens_.astype(np.int64)
            out_elems = int((lens_ * rows).sum())
        plan = eng.plan(case["seg_tid"], case["seg_start"], case["seg_end"], case["seg_strand"], out_off, step, stride,
                        out_elems, rows)
        spec = spec_for(oracle, mapping, sf)
        arrays, warn = oracle_expected(oracle, pa, case, spec)
        center = mapping[0] == "center"
        exp = np.zeros(out_elems, np.float64 if center else np.int64)
        for s, arr in enumerate(arrays):
            a2 = arr.reshape(rows, -1)
            for r in range(rows):
                if layout == "sums":
                    exp[out_off[s] + r] += a2[r].sum()
                else:
                    idx = out_off[s] + int(step[s]) * np.arange(a2.shape[1]) + r * stride[s]
                    exp[idx] = a2[r]
        tag = "seed %s %s sf=%s knobs=%s layout=%s files=%s" % (
            case["seed"], mapping, sf, case["knobs"], layout, [f.n for f in case["files"]])
        if not center:
            got = plan.count(np.int64)
            assert got.dtype == np.int64 and np.array_equal(got, exp), tag
            assert plan.total() == exp.sum(), tag
        if layout != "sums":
            got = plan.count(np.float64)
            assert np.array_equal(got, exp.astype(np.float64)), tag
        assert np.array_equal(plan.warn_flags(), warn), tag
        if layout != "sums" and rng.random() < 0.5:   # count / float(sum) * 1e6, genome_array.py:826-827
            total = float(rng.integers(1, 10 ** 9))
            eng.set_normalize(True, total)
            assert np.array_equal(plan.count(np.float64), exp.astype(np.float64) / total * 1e6), tag
        plan.close()
        eng.close()
    finally:
        for k in KNOBS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]
