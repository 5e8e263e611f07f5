"""Parity of the HIP path against (a) golden vectors from the reference and (b) the
oracle on seeded random inputs.  Needs a real MI355X: ``pytest -m gpu``."""
import os
import sys
import warnings

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import golden_util as gu  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import plastid_amd
    return plastid_amd


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)


def files_of(pa, g, case):
    """Split a fixture's file-major arrays back into PackedAlignments files."""
    aln = g.aln(case)
    refs, lens = case["aln"]["references"], case["aln"]["lengths"]
    files = []
    nb = aln["nblk"].astype(np.int64)
    if "wide_idx" in aln:                       # reads beyond the 16-bit / 8-bit fields: true run counts aside
        nb[aln["wide_idx"]] = aln["wide_nblk"]
    off = np.cumsum(np.where(nb >= 2, nb, 0))
    off = np.concatenate([[0], off])
    for k in range(case["aln"]["nfiles"]):
        idx = np.nonzero(aln["file_id"] == k)[0]
        lo, hi = (idx[0], idx[-1] + 1) if len(idx) else (0, 0)
        wide = {}
        if "wide_idx" in aln:
            w = (aln["wide_idx"] >= lo) & (aln["wide_idx"] < hi)
            wide = dict(wide_idx=aln["wide_idx"][w] - lo, wide_alen=aln["wide_alen"][w], wide_nblk=aln["wide_nblk"][w])
        if "flag16" in aln:                     # the SAM FLAG word and MAPQ of every read (flag_filters.npz)
            wide.update(flag16=aln["flag16"][lo:hi], mapq=aln["mapq"][lo:hi])
        if "nh" in aln:                         # the NH:i tag of every read (nh_filters.npz)
            wide.update(nh=aln["nh"][lo:hi])
        files.append(pa.PackedAlignments(
            aln["tid"][lo:hi], aln["pos"][lo:hi], aln["alen"][lo:hi], aln["flags"][lo:hi], aln["nblk"][lo:hi],
            aln["blk_start"][off[lo]:off[hi]], aln["blk_len"][off[lo]:off[hi]], references=refs, lengths=lens,
            mapped=case["aln"]["mapped"][k], **wide))
    return files


def factory_of(pa, spec):
    k = spec["kind"]
    od = gu.offset_dict_of(spec)
    if k == "fiveprime":
        return pa.FivePrimeMapFactory(spec["param"])
    if k == "threeprime":
        return pa.ThreePrimeMapFactory(spec["param"])
    if k == "center":
        return pa.CenterMapFactory(spec["param"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if k == "variable":
            return pa.VariableFivePrimeMapFactory(od)
        return pa.StratifiedVariableFivePrimeMapFactory(od, spec["min_len"], spec["max_len"])


def call_with_warnings(fn, *a, **k):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = fn(*a, **k)
    return out, [x for x in w if issubclass(x.category, UserWarning)]


# ------------------------------------------------------------------ golden
def test_kat_direct_factory_calls(pa):
    """The reference's own closed-form unit-test vectors, through the plugin API
    ``factory(reads, segment)`` (test_map_factories.py:17-200)."""
    g = gu.load("kat_map_factories")
    for case in g.cases:
        packed = files_of(pa, g, case)[0]
        reads = [packed.read(i) for i in range(packed.n)]
        fn = factory_of(pa, case["spec"])
        seg = pa.GenomicSegment(case["chrom"], case["start"], case["end"], case["strand"])
        (reads_out, arr), warns = call_with_warnings(fn, reads, seg)
        assert same(arr, g[case["expected"]]), case["spec"]
        assert [r.index for r in reads_out] == list(g[case["reads_out"]]), case["spec"]
        assert (len(warns) > 0) == case["warned"], case["spec"]


@pytest.mark.parametrize("group", ["quirks", "random_reads", "chains", "wide_reads"])
def test_golden_bamgenomearray(pa, group):
    g = gu.load(group)
    nq = 0
    for case in g.cases:
        if case["kind"] != "ga":
            continue
        files = files_of(pa, g, case)
        ga = pa.BAMGenomeArray(files, mapping=factory_of(pa, case["spec"]))
        if case.get("size_filter"):
            ga.add_filter("size", pa.SizeFilterFactory(min=case["size_filter"][0], max=case["size_filter"][1]))
        if case["normalize"]:
            ga.set_sum(case["sum"])
            ga.set_normalize(True)
        assert ga.sum() == case["sum"]
        offs = np.cumsum([0] + [f.n for f in files])
        for q in case["queries"]:
            nq += 1
            if q["type"] == "segment":
                seg = pa.GenomicSegment(q["chrom"], q["start"], q["end"], q["strand"])
                (reads, arr), warns = call_with_warnings(ga.get_reads_and_counts, seg, roi_order=q["roi_order"])
                assert same(arr, g[q["expected"]]), (case["spec"], q, case["note"])
                got = [offs[files.index(r.source)] + r.index for r in reads]
                assert got == list(g[q["reads_out"]]), (case["spec"], q)
                assert (len(warns) > 0) == q["warned"], (case["spec"], q)
                if "warn_messages" in q:   # the text too, incl. the read length the Variable rule names (:633-648)
                    assert [str(w.message) for w in warns] == q["warn_messages"], (case["spec"], q)
                assert same(ga.get(seg, roi_order=q["roi_order"]), g[q["expected"]])
                if q["roi_order"]:
                    assert same(ga[seg], g[q["expected"]])
            else:
                segs = [pa.GenomicSegment(q["chrom"], s, e, q["strand"]) for s, e in q["segments"]]
                chain = pa.SegmentChain(*segs)
                assert [(s.start, s.end) for s in chain] == [tuple(x) for x in q["merged_segments"]]
                assert chain.length == q["length"]
                if q.get("masks"):
                    chain.add_masks(*[pa.GenomicSegment(q["chrom"], s, e, q["strand"]) for s, e in q["masks"]])
                    assert chain.masked_length == q["masked_length"]
                    assert [(s.start, s.end) for s in chain.mask_segments] == [tuple(x) for x in q["mask_segments"]]
                if q.get("raises"):
                    with pytest.raises(ValueError):
                        chain.get_counts(ga)
                    continue
                arr = chain.get_counts(ga, stranded=q.get("stranded", True))
                assert same(arr, g[q["expected"]]), (case["spec"], q)
                if q.get("stranded", True):
                    assert same(ga[chain], arr) and same(ga.get(chain), arr)
                m = chain.get_masked_counts(ga)
                assert same(np.ma.getdata(m), g[q["masked_data"]])
                assert same(np.ma.getmaskarray(m), g[q["masked_mask"]])
                assert chain.get_position_list() == list(g[q["position_list"]])
                assert sorted(chain.get_masked_position_set()) == list(g[q["masked_position_set"]])
    assert nq > (100 if group != "wide_reads" else 70)


@pytest.mark.parametrize("how", ["device", "callable"])
def test_golden_flag_and_mapq_filters(pa, how):
    """Filters on the FLAG word and MAPQ against what the reference returned for plain callables on
    ``read.is_secondary`` / ``.mapping_quality`` ... (tests/golden/make_flag_golden.py; genome_array.py:697-722,
    819-820): as :class:`FlagFilterFactory` (evaluated on the GPU, pc_set_flag_filter) and as the same plain callable
    on the mirror's read objects (evaluated on the host, staged as exclusion bits) -- vectors, reads_out, warnings,
    for all five rules, one and two files, next to a size filter and normalised."""
    from plastid_amd.map_factories import FLAG_BITS
    g = gu.load("flag_filters")
    nq = 0
    for case in g.cases:
        files = files_of(pa, g, case)
        ga = pa.BAMGenomeArray(files, mapping=factory_of(pa, case["spec"]))
        req, exc, mq = case["filter"]
        if how == "device":
            ga.add_filter("flags", pa.FlagFilterFactory(req, exc, mq))
        else:
            need = [k for k, b in FLAG_BITS.items() if req & b]
            ban = [k for k, b in FLAG_BITS.items() if exc & b]
            ga.add_filter("flags", lambda r, need=need, ban=ban, mq=mq: all(getattr(r, k) for k in need) and
                          not any(getattr(r, k) for k in ban) and r.mapping_quality >= mq)
        if case["size_filter"]:
            ga.add_filter("size", pa.SizeFilterFactory(min=case["size_filter"][0], max=case["size_filter"][1]))
        if case["normalize"]:
            ga.set_normalize(True)
        assert ga.sum() == case["sum"]
        offs = np.cumsum([0] + [f.n for f in files])
        for q in case["queries"]:
            nq += 1
            if q["type"] == "segment":
                seg = pa.GenomicSegment(q["chrom"], q["start"], q["end"], q["strand"])
                (reads, arr), warns = call_with_warnings(ga.get_reads_and_counts, seg)
                assert same(arr, g[q["expected"]]), (case["spec"], case["filter_name"], q)
                assert [offs[files.index(r.source)] + r.index for r in reads] == list(g[q["reads_out"]]), (case["spec"], q)
                assert (len(warns) > 0) == q["warned"], (case["spec"], q)
            else:
                chain = pa.SegmentChain(*[pa.GenomicSegment(q["chrom"], s, e, q["strand"]) for s, e in q["segments"]])
                assert same(chain.get_counts(ga), g[q["expected"]]), (case["spec"], case["filter_name"], q)
        if how == "device":
            assert ga._engine._state["flagfilter"] == (True, req, exc, mq)     # it ran on the GPU, not read by read
    assert nq > 800


@pytest.mark.parametrize("how", ["device", "callable"])
def test_golden_nh_filters(pa, how):
    """Filters on the NH:i tag -- the unique-mapper test of the reference's users -- against what the reference returned
    for ``lambda read: read.has_tag("NH") and read.get_tag("NH") <= k`` next to FLAG / MAPQ tests
    (tests/golden/make_nh_golden.py; genome_array.py:697-722, 819-820): as ``FlagFilterFactory(max_nh=k)`` (evaluated on
    the GPU: pc_set_nh_filter) and as the same plain callable on the mirror's read objects (``PackedRead.get_tag``;
    evaluated on the host, staged as exclusion bits) -- vectors, reads_out, warnings, all five rules, one and two files,
    next to a size filter and normalised."""
    from plastid_amd.map_factories import FLAG_BITS
    g = gu.load("nh_filters")
    nq = 0
    for case in g.cases:
        files = files_of(pa, g, case)
        assert all(f.nh is not None for f in files)
        ga = pa.BAMGenomeArray(files, mapping=factory_of(pa, case["spec"]))
        req, exc, mq, max_nh = case["filter"]
        if how == "device":
            ga.add_filter("nh", pa.FlagFilterFactory(req, exc, mq, max_nh=max_nh))
        else:
            need = [k for k, b in FLAG_BITS.items() if req & b]
            ban = [k for k, b in FLAG_BITS.items() if exc & b]
            ga.add_filter("nh", lambda r, need=need, ban=ban, mq=mq, k=max_nh: all(getattr(r, x) for x in need) and
                          not any(getattr(r, x) for x in ban) and r.mapping_quality >= mq and r.has_tag("NH") and r.get_tag("NH") <= k)
        if case["size_filter"]:
            ga.add_filter("size", pa.SizeFilterFactory(min=case["size_filter"][0], max=case["size_filter"][1]))
        if case["normalize"]:
            ga.set_normalize(True)
        assert ga.sum() == case["sum"]
        offs = np.cumsum([0] + [f.n for f in files])
        for q in case["queries"]:
            nq += 1
            if q["type"] == "segment":
                seg = pa.GenomicSegment(q["chrom"], q["start"], q["end"], q["strand"])
                (reads, arr), warns = call_with_warnings(ga.get_reads_and_counts, seg)
                assert same(arr, g[q["expected"]]), (case["spec"], case["filter_name"], q)
                assert [offs[files.index(r.source)] + r.index for r in reads] == list(g[q["reads_out"]]), (case["spec"], q)
                assert (len(warns) > 0) == q["warned"], (case["spec"], q)
            else:
                chain = pa.SegmentChain(*[pa.GenomicSegment(q["chrom"], s, e, q["strand"]) for s, e in q["segments"]])
                assert same(chain.get_counts(ga), g[q["expected"]]), (case["spec"], case["filter_name"], q)
        if how == "device":
            assert ga._engine._state["nhfilter"] == max_nh     # it ran on the GPU, not read by read
    assert nq > 600


# ------------------------------------------------------------------ oracle, seeded random, through the C ABI
def aln_dict(files):
    from plastid_amd.packing import concat_file_major
    return concat_file_major(files)


def spec_for(oracle, mapping, size_filter=None):
    kind = mapping[0]
    if kind in ("fiveprime", "threeprime", "center"):
        return oracle.mapping_spec(kind, mapping[1], size_filter=size_filter)
    if kind == "variable":
        return oracle.mapping_spec(kind, 0, mapping[1], size_filter=size_filter)
    return oracle.mapping_spec(kind, 0, mapping[1], mapping[2], mapping[3], size_filter=size_filter)


def engine_for(pa, files, mapping, size_filter=None):
    from plastid_amd.engine import Engine
    from plastid_amd import synth
    eng = Engine(0)
    eng.set_alignments(files)
    synth.mapping_factory(mapping)._configure(eng)
    if size_filter:
        eng.set_size_filter(*size_filter)
    return eng


def oracle_chain_outputs(oracle, files, spec, tx, plan_arrays, rows, dtype):
    """Oracle counts for every segment, scattered into the plan's output layout."""
    aln = aln_dict(files)
    arrays, warn = oracle.count_segments(aln, spec, plan_arrays["tid"], plan_arrays["start"], plan_arrays["end"],
                                         plan_arrays["strand"])
    out = np.zeros(plan_arrays["out_elems"], dtype)
    for s, arr in enumerate(arrays):
        n = arr.shape[-1]
        idx = plan_arrays["out_off"][s] + plan_arrays["out_step"][s].astype(np.int64) * np.arange(n)
        a2 = arr.reshape(rows, n)
        for r in range(rows):
            out[idx + r * plan_arrays["row_stride"][s]] = a2[r]
    return out, warn


MAPPINGS = [
    ("fiveprime", 0), ("fiveprime", 12), ("fiveprime", 27), ("threeprime", 0), ("threeprime", 14),
    ("center", 0), ("center", 5), ("center", 13),
    ("variable", {26: 12, 27: 12, 28: 13, 29: 13, 30: 14, 31: 13, "default": 13}),
    ("variable", {28: 5, 30: 29}),
    ("stratified", {26: 12, 27: 12, 28: 13, 29: 13, 30: 14, 31: 13, "default": 13}, 25, 35),
    ("stratified", {28: 5}, 27, 30),
]


@pytest.mark.parametrize("config,scale,tx_scale", [("C2", 0.001, 0.01), ("C4", 0.0002, 0.005), ("C5", 0.0001, 0.005)])
@pytest.mark.parametrize("mapping", MAPPINGS, ids=lambda m: "%s-%s" % (m[0], str(m[1])[:12]))
def test_random_vs_oracle(pa, oracle, config, scale, tx_scale, mapping):
    """Seeded synthetic reads (incl. spliced + deleted bases) x transcripts: every
    output element equals the oracle's, int64 and float64 layouts."""
    from plastid_amd import synth
    genome, tx, reads, _ = synth.make_config(config, scale=scale, tx_scale=tx_scale)
    eng = engine_for(pa, [reads], mapping)
    rows = eng.rows
    parr = tx.plan_arrays(rows=rows)
    plan = eng.plan(parr["tid"], parr["start"], parr["end"], parr["strand"], parr["out_off"], parr["out_step"],
                    parr["row_stride"], parr["out_elems"], rows)
    spec = spec_for(oracle, mapping)
    center = mapping[0] == "center"
    exp, warn = oracle_chain_outputs(oracle, [reads], spec, tx, parr, rows, np.float64 if center else np.int64)
    got64 = plan.count(np.float64)
    assert np.array_equal(got64, exp.astype(np.float64))
    if not center:
        got = plan.count(np.int64)
        assert got.dtype == np.int64 and np.array_equal(got, exp)
        assert plan.total() == exp.sum()
    assert np.array_equal(plan.warn_flags(), warn)
    # normalisation: count / float(sum) * 1e6 (genome_array.py:826-827)
    eng.set_normalize(True, 123457.0)
    assert np.array_equal(plan.count(np.float64), exp / float(123457.0) * 1e6)
    plan.close()
    eng.close()


def test_pileup_and_multifile(pa, oracle):
    """A pile-up (> one work item per tile, merged with global atomics), two files
    (file-major center order), '.' segments, size filter, overlapping segments."""
    from plastid_amd import synth
    from plastid_amd.engine import Engine
    rng = np.random.default_rng(5)
    names, lens = ["a", "b"], [50000, 20000]
    n = 300000
    pos = np.sort(np.concatenate([rng.integers(1000, 1040, n - 20000), rng.integers(0, 19000, 20000)]))
    tid = np.zeros(n, np.int32)
    alen = rng.integers(20, 40, n)
    rev = rng.random(n) < 0.5
    f1 = pa.PackedAlignments.from_ungapped(tid, pos, alen, rev, references=names, lengths=lens)
    _, _, f2, _ = synth.make_config("C2", scale=0.0005, tx_scale=0.001)
    f2 = pa.PackedAlignments(np.zeros(f2.n, np.int32), np.sort(f2.pos % 40000), f2.alen, f2.flags,
                             np.minimum(f2.nblk, 1), references=names, lengths=lens)
    seg_start = np.array([0, 900, 1000, 1010, 1030, 0, 5000, 1000, 0], np.int64)
    seg_end = np.array([50000, 1100, 1001, 1500, 1031, 2000, 5000, 1040, 20000], np.int64)
    seg_tid = np.array([0, 0, 0, 0, 0, 0, 0, 0, 1], np.int32)
    seg_strand = np.array([1, 2, 3, 3, 1, 2, 1, 3, 3], np.uint8)
    lens_ = seg_end - seg_start
    for mapping in [("fiveprime", 3), ("center", 2), ("threeprime", 0)]:
        for sf in (None, (25, 33)):
            eng = engine_for(pa, [f1, f2], mapping, sf)
            out_off = np.concatenate([[0], np.cumsum(lens_)[:-1]])
            plan = eng.plan(seg_tid, seg_start, seg_end, seg_strand, out_off, np.ones(len(lens_), np.int8), lens_,
                            int(lens_.sum()), 1)
            spec = spec_for(oracle, mapping, sf)
            arrays, warn = oracle.count_segments(aln_dict([f1, f2]), spec, seg_tid, seg_start, seg_end, seg_strand)
            exp = np.concatenate(arrays)
            got = plan.count(exp.dtype)
            assert np.array_equal(got, exp), (mapping, sf)
            assert np.array_equal(plan.warn_flags(), warn)
            plan.close()
            eng.close()


def test_compact_histogram_cleared_slice_by_slice(pa, oracle, monkeypatch):
    """A large plan's compact histogram (what merged windows go through) is never cleared as a whole: k_clear_split zeroes
    the slices of the merged windows behind every k_tile_ranges, k_gather_split leaves them zero after every count.  With
    PC_HIST_LAZY_BYTES=1 a small plan takes that path: a plan whose device block comes back from the pool dirty (a plan of
    the same size was counted and closed before), several counts of one plan, and the SAME plan after the engine's
    alignments changed under it (the lists are rebuilt, other windows are merged) -- against the oracle every time."""
    rng = np.random.default_rng(77)
    names, lens = ["a"], [200000]

    def piled(at, n=250000):
        pos = np.sort(np.concatenate([rng.integers(at, at + 40, n - 50000), rng.integers(0, 190000, 50000)]))
        return pa.PackedAlignments.from_ungapped(np.zeros(n, np.int32), pos, rng.integers(20, 40, n), rng.random(n) < 0.5,
                                                 references=names, lengths=lens)
    f_a, f_b = piled(5000), piled(90000)
    seg_start = np.array([0, 4000, 89000, 5000, 90010], np.int64)
    seg_end = np.array([200000, 7000, 92000, 5040, 90011], np.int64)
    seg_tid = np.zeros(len(seg_start), np.int32)
    seg_strand = np.array([1, 2, 3, 3, 1], np.uint8)
    lens_ = seg_end - seg_start
    out_off = np.concatenate([[0], np.cumsum(lens_)[:-1]])
    monkeypatch.setenv("PC_WORK_R", "512")
    monkeypatch.setenv("PC_PILE", "2048")
    mapping = ("fiveprime", 3)
    spec = spec_for(oracle, mapping, None)
    expected = {}
    for key, files in (("a", [f_a]), ("b", [f_b]), ("ab", [f_a, f_b])):
        expected[key] = np.concatenate(oracle.count_segments(aln_dict(files), spec, seg_tid, seg_start, seg_end, seg_strand)[0])
        assert expected[key].max() > 3000                         # (the piles are there)
    for lazy in ("1", None):
        if lazy: monkeypatch.setenv("PC_HIST_LAZY_BYTES", lazy)
        else: monkeypatch.delenv("PC_HIST_LAZY_BYTES", raising=False)
        eng = engine_for(pa, [f_a], mapping, None)

        def mk():
            return eng.plan(seg_tid, seg_start, seg_end, seg_strand, out_off, np.ones(len(lens_), np.int8), lens_, int(lens_.sum()), 1)
        dirty = mk(); dirty.count(np.int64); dirty.close()        # its block goes back to the pool with whatever the count left
        plan = mk()
        for key, files in (("a", None), ("b", [f_b]), ("a", [f_a]), ("ab", [f_a, f_b]), ("b", [f_b])):
            if files is not None:
                eng.clear_alignments()
                eng.set_alignments(files)                         # under the plan: its lists are rebuilt at the next count
            for _ in range(2):
                assert np.array_equal(plan.count(np.int64), expected[key]), (lazy, key)
        plan.close()
        eng.close()


def test_long_aligned_lengths(pa, oracle):
    """Ungapped reads too long for the 4-byte record stream (L > 255) are binned from the side
    list, and those longer than the window halo through the long-span path; short-only data stays
    in the stream.  All must match the oracle, including the nofilter/'.' strand modes and odd
    record ranges."""
    rng = np.random.default_rng(23)
    names, lens = ["a", "b"], [60000, 9000]
    for lmax in (40, 3000):
        n = 50001
        tid = np.sort((rng.random(n) < 0.1).astype(np.int32))
        alen = np.where(rng.random(n) < 0.02, rng.integers(1, lmax + 1, n), rng.integers(18, 41, n))
        pos = np.empty(n, np.int64)
        for t in (0, 1):
            m = tid == t
            pos[m] = np.sort(rng.integers(0, lens[t] - 3100, int(m.sum())))
        rev = rng.random(n) < 0.5
        f1 = pa.PackedAlignments.from_ungapped(tid, pos, alen, rev, references=names, lengths=lens)
        f_allrev = pa.PackedAlignments.from_ungapped(tid, pos, alen, np.ones(n, bool), references=names, lengths=lens)
        seg_tid = np.array([0, 0, 0, 1, 0, 0], np.int32)
        seg_start = np.array([0, 100, 20000, 0, 31000, 7], np.int64)
        seg_end = np.array([60000, 30000, 20001, 9000, 33333, 4100], np.int64)
        seg_strand = np.array([1, 2, 3, 3, 2 | 0x10, 1 | 0x10], np.uint8)
        lens_ = seg_end - seg_start
        out_off = np.concatenate([[0], np.cumsum(lens_)[:-1]])
        voff = {26: 12, 27: 12, 28: 13, 29: 13, 30: 14, 31: 13, 600: 300, "default": 13}
        for mapping in [("fiveprime", 12), ("threeprime", 30), ("variable", voff), ("stratified", voff, 20, 32)]:
            for sf in (None, (20, 500)):
                eng = engine_for(pa, [f1], mapping, sf)
                rows = eng.rows
                plan = eng.plan(seg_tid, seg_start, seg_end, seg_strand, out_off * rows, np.ones(len(lens_), np.int8),
                                lens_, int(lens_.sum()) * rows, rows)
                spec = spec_for(oracle, mapping, sf)
                # the oracle has no "unfiltered" strand code: all reads under the forward rule is what
                # '.' does, all reads under the reverse rule is a '-' query over reads all flagged reverse
                ostrand = np.where(seg_strand == (1 | 0x10), 3, seg_strand & 3).astype(np.uint8)
                arrays, warn = oracle.count_segments(aln_dict([f1]), spec, seg_tid, seg_start, seg_end, ostrand)
                nf = np.nonzero(seg_strand == (2 | 0x10))[0]
                arr2, warn2 = oracle.count_segments(aln_dict([f_allrev]), spec, seg_tid[nf], seg_start[nf], seg_end[nf],
                                                    ostrand[nf])
                for k, s_ in enumerate(nf):
                    arrays[s_] = arr2[k]
                    warn[s_] = warn2[k]
                exp = np.concatenate([a.reshape(-1) for a in arrays])
                got = plan.count(np.int64)
                assert np.array_equal(got, exp), (lmax, mapping, sf)
                assert np.array_equal(plan.warn_flags(), warn)
                plan.close()
                eng.close()


def test_genome_partition_through_the_engine(pa, oracle):
    """every genome range counted by its own engine instance (what one rank per GPU does)
    and assembled on the host equals the single-engine result; per-chain sums complete by addition."""
    from plastid_amd import multigpu, synth
    genome, tx, reads, _ = synth.make_config("C4", scale=0.0005, tx_scale=0.005)
    for mapping, rows in ((("fiveprime", 12), 1), (("center", 2), 1), (("stratified", synth.VARIABLE_OFFSETS, 27, 31), 5)):
        center = mapping[0] == "center"
        dtype = np.float64 if center else np.int64
        p = tx.plan_arrays(rows=rows)
        eng = engine_for(pa, [reads], mapping)
        plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"],
                        p["out_elems"], rows)
        want = plan.count(dtype)
        plan.close()
        eng.close()
        world = 4
        part = multigpu.GenomePartition([reads], p, world)
        got = np.zeros(p["out_elems"], dtype)
        for r in range(world):
            lp = part.local_plan_arrays(r, rows)
            eng = engine_for(pa, part.records(r), mapping)
            plan = eng.plan(lp["tid"], lp["start"], lp["end"], lp["strand"], lp["out_off"], lp["out_step"],
                            lp["row_stride"], lp["out_elems"], rows)
            part.scatter_local(got, r, plan.count(dtype), rows)
            plan.close()
            eng.close()
        assert np.array_equal(got, want), mapping
    # fused per-chain sums: each rank fills its share of one [n_chains] vector; the all-reduce is a sum
    p = tx.plan_arrays(rows=1)
    seg = dict(p, out_off=tx.ex_tx.astype(np.int64), out_step=np.zeros(len(p["tid"]), np.int8),
               row_stride=np.ones(len(p["tid"]), np.int64))
    eng = engine_for(pa, [reads], ("threeprime", 0))
    plan = eng.plan(seg["tid"], seg["start"], seg["end"], seg["strand"], seg["out_off"], seg["out_step"],
                    seg["row_stride"], tx.n, 1)
    want = plan.count(np.int64)
    plan.close()
    eng.close()
    part = multigpu.GenomePartition([reads], seg, 3)
    acc = np.zeros(tx.n, np.int64)
    for r in range(3):
        sg = part.segments(r)
        eng = engine_for(pa, part.records(r), ("threeprime", 0))
        plan = eng.plan(sg["tid"], sg["start"], sg["end"], sg["strand"], sg["out_off"], sg["out_step"],
                        sg["row_stride"], tx.n, 1)
        acc += plan.count(np.int64)
        plan.close()
        eng.close()
    assert np.array_equal(acc, want)


def test_threaded_staging_is_deterministic(pa, monkeypatch):
    """pc_add_alignment_file examines the contig column on several host threads and sends the other columns through a
    ring of page-locked pieces: the staged result (hence every count) and the reported error must not depend on the
    thread count or the piece size."""
    from plastid_amd import synth
    from plastid_amd.engine import Engine
    genome, tx, reads, _ = synth.make_config("C4", scale=0.005, tx_scale=0.01)      # 2.5 M reads, spliced
    p = tx.plan_arrays(rows=1)
    outs = []
    for threads, slice_records in (("1", None), ("5", None), ("5", "300000"), ("2", "77777")):
        monkeypatch.setenv("PC_STAGE_THREADS", threads)
        if slice_records is None:
            monkeypatch.delenv("PC_STAGE_SLICE", raising=False)
        else:
            monkeypatch.setenv("PC_STAGE_SLICE", slice_records)    # the columns in pieces through the page-locked ring
        eng = engine_for(pa, [reads], ("threeprime", 3))
        plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"],
                        p["out_elems"], 1)
        outs.append(plan.count(np.int64))
        plan.close()
        eng.close()
    assert all(np.array_equal(outs[0], o) for o in outs[1:]) and outs[0].sum() > 0
    monkeypatch.delenv("PC_STAGE_SLICE", raising=False)
    # two defects in different thread chunks: the one with the lower record index is reported
    bad_pos = reads.pos.copy()
    i_lo, i_hi = reads.n // 3, (2 * reads.n) // 3
    for i in (i_lo, i_hi):
        j = i
        while reads.tid[j] != reads.tid[j - 1] or reads.nblk[j] >= 2 or reads.nblk[j - 1] >= 2:
            j += 1
        bad_pos[j] = bad_pos[j - 1] - 1 if bad_pos[j - 1] > 0 else bad_pos[j]
        if i == i_lo:
            first = j
    broken = pa.PackedAlignments(reads.tid, bad_pos, reads.alen, reads.flags, reads.nblk, reads.blk_start, reads.blk_len,
                                 references=reads.references, lengths=reads.lengths, validate=False)
    for threads in ("1", "5"):
        monkeypatch.setenv("PC_STAGE_THREADS", threads)
        eng = Engine(0)
        with pytest.raises(ValueError) as ei:
            eng.set_alignments([broken])
        assert "record %d" % first in str(ei.value), str(ei.value)
        eng.close()


@pytest.mark.parametrize("slice_records", [None, "3"])
def test_staging_rejects_every_defect_of_the_columns(pa, monkeypatch, slice_records):
    """The caller's columns are validated on the GPU (stage_kernels.hip.h: k_cols_pack<true>; the contig column on the
    host, where it stays): every defect the packed-record contract names is reported with its record index and the
    message of the host-side validator (packing.PackedAlignments), the defect of the LOWEST record first -- also when one
    of two defects is in the contig column and the other is not.  With PC_STAGE_SLICE the columns cross PCIe through
    the ring of page-locked pieces, twelve bytes at a time."""
    from plastid_amd.engine import Engine
    if slice_records:
        monkeypatch.setenv("PC_STAGE_SLICE", slice_records)
    refs, lens = ["a", "b", "c"], [5000, 5000, 5000]
    tid = np.array([0, 0, 0, 1, 1, 1, 1, 2, 2], np.int32)
    pos = np.array([10, 20, 20, 5, 30, 40, 41, 0, 7], np.int32)
    alen = np.array([30, 25, 28, 30, 40, 30, 30, 30, 29], np.uint16)
    nblk = np.array([1, 1, 2, 1, 3, 1, 1, 1, 1], np.uint8)
    flags = np.zeros(9, np.uint8)
    blk_start = np.array([20, 60, 30, 50, 90], np.int32)
    blk_len = np.array([10, 18, 10, 20, 10], np.int32)
    cols = dict(tid=tid, pos=pos, alen=alen, flags=flags, nblk=nblk, blk_start=blk_start, blk_len=blk_len)

    def stage(**changed):
        c = {k: v.copy() for k, v in cols.items()}
        for k, (i, v) in changed.items():
            if i is None:
                c[k] = v
            else:
                c[k][i] = v
        reads = pa.PackedAlignments(c["tid"], c["pos"], c["alen"], c["flags"], c["nblk"], c["blk_start"], c["blk_len"],
                                    references=refs, lengths=lens, validate=False)
        eng = Engine(0)
        try:
            eng.set_alignments([reads])
            return None
        except ValueError as err:
            return str(err)
        finally:
            eng.close()

    assert stage() is None
    cases = [
        (dict(pos=(1, -3)), "record 1: negative position"),
        (dict(pos=(5, 29)), "not sorted by (tid, pos) at record 5"),
        (dict(tid=(4, 7)), "record 4: tid 7 out of range"),
        (dict(tid=(4, -1)), "record 4: tid -1 out of range"),
        (dict(tid=(0, -2)), "record 0: tid -2 out of range"),
        (dict(tid=(5, 0)), "not sorted by (tid, pos) at record 5"),
        (dict(blk_start=(0, 21)), "record 2: first run must start at pos"),
        (dict(blk_len=(1, 19)), "record 2: run lengths do not sum to alen"),
        (dict(blk_start=(3, 40)), "record 4: aligned runs must be non-empty, ascending and non-adjacent"),
        (dict(blk_len=(3, 0)), "record 4: aligned runs must be non-empty"),
        (dict(nblk=(6, 0)), "record 6: nblk/alen mismatch"),
        (dict(alen=(8, 0)), "record 8: nblk/alen mismatch"),
        (dict(pos=(8, 2**31 - 20)), "record 8: alignment end beyond 2^31-1"),
        (dict(nblk=(0, 2)), "run arrays shorter than sum of nblk"),
        (dict(blk_start=(None, np.append(blk_start, 7).astype(np.int32)), blk_len=(None, np.append(blk_len, 7).astype(np.int32))),
         "run arrays longer than sum of nblk"),
        # two defects: the lower record is the one reported, whichever column it is in
        (dict(tid=(6, 0), pos=(1, -3)), "record 1: negative position"),
        (dict(tid=(3, 2), pos=(5, 29)), "not sorted by (tid, pos) at record 4"),
        (dict(tid=(6, 0), pos=(5, 29)), "not sorted by (tid, pos) at record 5"),
        (dict(tid=(5, 9), blk_len=(1, 19)), "record 2: run lengths do not sum to alen"),
        # ... and of two checks that fail for one record, the first in the validator's order
        (dict(tid=(5, 0), pos=(5, -1)), "record 5: negative position"),
        (dict(pos=(4, -9)), "record 4: negative position"),
    ]
    for changed, want in cases:
        got = stage(**changed)
        assert got is not None and want in got, (changed, want, got)
    # contig starts are not defects: positions start over where the contig changes, and only there
    assert stage(pos=(3, 0)) is None and stage(pos=(7, 0)) is None


def test_transfer_ring_large_files_pinned_memory_and_two_engines(pa, monkeypatch):
    """Staging and read-back of a file large enough for the ring of page-locked pieces by itself (12 M records: 96 MB of
    columns up, 100 MB of counts down), against the same file sent in 40-byte pieces; page-locked caller memory (a pinned
    torch tensor as the output buffer) bypasses the ring; two engines staging and counting on two host threads at once --
    one ring per device, one transfer at a time -- give what each gives alone; and an engine made after another was
    destroyed (its device blocks come from the process-wide reservoir) counts the same."""
    import threading
    import torch
    from plastid_amd.engine import Engine
    from plastid_amd.packing import PackedAlignments
    n = 12_000_000
    rng = np.random.default_rng(5)
    pos = np.sort(rng.integers(0, 12_500_000, n)).astype(np.int32)
    alen = rng.integers(25, 36, n).astype(np.uint16)
    rev = rng.random(n) < 0.5
    tid = (pos >= 6_000_000).astype(np.int32)
    pos = np.where(tid == 1, pos - 6_000_000, pos).astype(np.int32)
    reads = PackedAlignments.from_ungapped(tid, pos, alen, rev, references=["a", "b"], lengths=[6_100_000, 6_600_000], validate=False)
    seg = dict(tid=np.array([0, 1], np.int32), start=np.array([0, 0], np.int64), end=np.array([6_050_000, 6_550_000], np.int64),
               strand=np.array([3, 3], np.uint8))     # PC_STRAND_UNS: reads of both strands
    seg["out_off"] = np.array([0, 6_050_000], np.int64)
    seg["out_step"] = np.array([1, 1], np.int8)
    seg["row_stride"] = np.array([0, 0], np.int64)
    out_elems = 12_600_000

    def count(out=None):
        eng = Engine(0)
        eng.set_alignments([reads])
        pa.FivePrimeMapFactory(3)._configure(eng)
        plan = eng.plan(seg["tid"], seg["start"], seg["end"], seg["strand"], seg["out_off"], seg["out_step"], seg["row_stride"], out_elems, 1)
        got = plan.count(np.int64, out=out)
        res = got.copy()
        plan.close()
        eng.close()
        return res

    want = count()
    # every read lands somewhere inside the two segments (5' end + 3 for forward reads, 3' end - 3 for reverse ones)
    assert want.sum() == n
    monkeypatch.setenv("PC_STAGE_SLICE", "10")            # 40-byte pieces up, 40 KiB pieces down
    small = count()
    monkeypatch.delenv("PC_STAGE_SLICE")
    assert np.array_equal(small, want)
    pinned = torch.zeros(out_elems, dtype=torch.int64, pin_memory=True)
    assert np.array_equal(count(out=pinned.numpy()), want)
    results = [None, None]
    def worker(k):
        results[k] = count()
    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert np.array_equal(results[0], want) and np.array_equal(results[1], want)
    assert np.array_equal(count(), want)
    # ... and after the process has given the kept blocks back to the driver
    import torch.cuda
    from plastid_amd.engine import release_cached_memory
    free_before = torch.cuda.mem_get_info(0)[0]
    release_cached_memory(0)
    assert torch.cuda.mem_get_info(0)[0] > free_before + (200 << 20)       # the records alone were 96 MB + 48 MB + ...
    assert np.array_equal(count(), want)


def test_stratified_with_many_rows(pa, oracle):
    """36 length rows x both strands need more than 64 KiB of LDS bins per window: gfx950 lets a
    workgroup have it."""
    from plastid_amd import synth
    genome, tx, reads, _ = synth.make_config("C2", scale=0.002, tx_scale=0.004)
    mapping = ("stratified", {"default": 12, 30: 15}, 15, 50)
    eng = engine_for(pa, [reads], mapping)
    rows = eng.rows
    assert rows == 36
    parr = tx.plan_arrays(rows=rows)
    plan = eng.plan(parr["tid"], parr["start"], parr["end"], parr["strand"], parr["out_off"], parr["out_step"],
                    parr["row_stride"], parr["out_elems"], rows)
    exp, _ = oracle_chain_outputs(oracle, [reads], spec_for(oracle, mapping), tx, parr, rows, np.int64)
    assert np.array_equal(plan.count(np.int64), exp) and exp.sum() > 0
    plan.close()
    eng.close()


def test_to_genome_array(pa):
    """BAMGenomeArray.to_genome_array (genome_array.py:965-988): the dense copy answers segment and
    chain queries like the BAM-backed array (last nucleotide of every contig excepted, as in the
    reference)."""
    from plastid_amd import synth
    genome, tx, reads, _ = synth.make_config("C2", scale=0.001, tx_scale=0.002)
    small = reads.subset(np.nonzero(reads.tid < 2)[0])          # two contigs are enough
    ga = pa.BAMGenomeArray(small, mapping=pa.FivePrimeMapFactory(offset=12))
    dense = ga.to_genome_array()
    assert sorted(dense.chroms()) == sorted(ga.chroms()) and tuple(dense.strands()) == tuple(ga.strands())
    rng = np.random.default_rng(3)
    names = list(ga.chroms())[:2]
    for _ in range(20):
        chrom = names[int(rng.integers(0, 2))]
        a = int(rng.integers(0, ga.lengths()[chrom] - 2000))
        seg = pa.GenomicSegment(chrom, a, a + int(rng.integers(1, 1500)), "+-."[int(rng.integers(0, 3))])
        assert np.array_equal(dense[seg], ga[seg].astype(float))
        assert np.array_equal(dense.get(seg, roi_order=False), ga.get(seg, roi_order=False).astype(float))
    chains = [c for c in tx.chains() if c.chrom in names][:10]
    for c in chains:
        assert np.array_equal(c.get_counts(dense), c.get_counts(ga))
    total = sum(float(ga[pa.GenomicSegment(ch, 0, ga.lengths()[ch] - 1, st)].sum()) for ch in ga.chroms() for st in ga.strands())
    assert dense.sum() == total


def test_inverse_table_is_ieee(pa):
    """1.0/m used by the center kernel is the host's correctly rounded quotient;
    a lone read of aligned length m contributes exactly 1.0/m at each position."""
    for m in (1, 3, 7, 25, 29, 33, 49, 97, 1001):
        packed = pa.PackedAlignments.from_ungapped(0, [10], [m], [False], references=["c"], lengths=[5000])
        ga = pa.BAMGenomeArray(packed, mapping=pa.CenterMapFactory(0))
        arr = ga.get(pa.GenomicSegment("c", 0, 2000, "+"))
        assert arr.dtype == np.float64
        exp = np.zeros(2000)
        exp[10:10 + m] = 1.0 / m
        assert np.array_equal(arr, exp)


def test_errors_and_edges(pa):
    packed = pa.PackedAlignments.from_ungapped(0, [5, 9], [30, 31], [False, True], references=["c"], lengths=[100])
    ga = pa.BAMGenomeArray(packed, mapping=pa.FivePrimeMapFactory(0))
    # unknown chromosome: zeros([1]) float64 (genome_array.py:795-798)
    arr = ga[pa.GenomicSegment("nope", 0, 50, "+")]
    assert arr.shape == (1,) and arr.dtype == np.float64
    # empty segment, segment beyond the contig end
    assert ga[pa.GenomicSegment("c", 7, 7, "+")].shape == (0,)
    assert ga[pa.GenomicSegment("c", 90, 400, "+")].sum() == 0
    # unsorted input is rejected like an unindexed BAM (ValueError)
    with pytest.raises(ValueError):
        pa.PackedAlignments.from_ungapped(0, [9, 5], [30, 30], [False, False], references=["c"], lengths=[100])
    # custom (plugin) mapping function and custom filter keep the reference's contract
    def my_map(reads, seg):
        out = np.zeros(len(seg), int)
        for r in reads:
            out[r.positions[0] - seg.start] += 2
        return reads, out
    ga.set_mapping(my_map)
    assert ga[pa.GenomicSegment("c", 0, 50, "+")][5] == 2
    ga.set_mapping(pa.FivePrimeMapFactory(0))
    ga.add_filter("only31", lambda r: len(r.positions) == 31)
    assert ga[pa.GenomicSegment("c", 0, 50, ".")].sum() == 1
    assert ga.remove_filter("only31") is not None
    assert ga[pa.GenomicSegment("c", 0, 50, ".")].sum() == 2
    # empty read list through the plugin API
    reads_out, arr = pa.FivePrimeMapFactory(0)([], pa.GenomicSegment("c", 0, 10, "+"))
    assert reads_out == [] and arr.shape == (10,) and arr.dtype == np.int64 and arr.sum() == 0


def test_bam_file_end_to_end(pa, oracle, tmp_path):
    """BAM on disk -> native stager -> HBM -> counts, vs the oracle on the same records;
    count_table / get_counts_batch / chain.get_counts agree."""
    from plastid_amd import synth
    from plastid_amd.annotation import IntervalTable
    from tests import bam_writer
    genome, tx, reads, _ = synth.make_config("C2", scale=0.0004, tx_scale=0.004)
    path = str(tmp_path / "reads.bam")
    bam_writer.write_bam(path, list(reads.references), list(reads.lengths), bam_writer.packed_to_records(reads))
    ga = pa.BAMGenomeArray(path, mapping=pa.FivePrimeMapFactory(12))
    ga.add_filter("size", pa.SizeFilterFactory(min=26, max=32))
    assert ga.sum() == reads.n and ga.chroms() == sorted(reads.references)
    table = IntervalTable(tx.references, tx.ref_lengths, tx.tid, tx.strand, tx.ex_off, tx.ex_start, tx.ex_end)
    flat, per_chain = ga.count_table(table)
    spec = spec_for(oracle, ("fiveprime", 12), (26, 32))
    parr = tx.plan_arrays(rows=1)
    exp, _ = oracle_chain_outputs(oracle, [reads], spec, tx, parr, 1, np.float64)
    assert flat.dtype == np.float64 and np.array_equal(flat, exp)
    chains = tx.chains(limit=25)
    batch = ga.get_counts_batch(chains)
    for c, a, b in zip(chains, batch, per_chain):
        assert np.array_equal(a, b) and np.array_equal(c.get_counts(ga), a) and np.array_equal(ga[c], a)
    # stratified through the same objects: [rows, length] per chain
    ga.set_mapping(pa.StratifiedVariableFivePrimeMapFactory(synth.VARIABLE_OFFSETS, 25, 35))
    flat2, per_chain2 = ga.count_table(table)
    assert per_chain2[0].shape == (11, int(tx.length[0]))
    assert np.array_equal(chains[3].get_counts(ga), per_chain2[3])
    # bedGraph / wiggle export run off the same path
    import io
    ga.set_mapping(pa.FivePrimeMapFactory(0))
    ga.remove_filter("size")
    buf = io.StringIO()
    ga.to_bedgraph(buf, "t", "+", window_size=50000)
    total = 0
    for line in buf.getvalue().splitlines()[1:]:
        chrom, s, e, v = line.split("\t")
        total += (int(e) - int(s)) * int(v)
    assert total == int(((reads.flags & 1) == 0).sum())


def test_region_limited_staging_counts_like_the_whole_file(pa, tmp_path):
    """``BAMGenomeArray(path, regions=...)`` stages only what the BAI index says overlaps the regions;
    every count inside the regions equals the whole-file count (all five rules see the same reads
    there, htslib fetch semantics), and ``sum()`` is the whole file's, from the index."""
    from plastid_amd import synth
    from tests import bam_writer
    genome, tx, reads, _ = synth.make_config("C4", scale=0.0002, tx_scale=0.002)      # 100 k reads, spliced
    path = str(tmp_path / "reads.bam")
    bam_writer.write_bam(path, list(reads.references), list(reads.lengths), bam_writer.packed_to_records(reads),
                         block_bytes=20000, index=True)
    chains = tx.chains(limit=40)
    regions = [seg for c in chains for seg in c]
    for mapping in (pa.FivePrimeMapFactory(12), pa.CenterMapFactory(2), pa.VariableFivePrimeMapFactory(synth.VARIABLE_OFFSETS)):
        whole = pa.BAMGenomeArray(path, mapping=mapping)
        part = pa.BAMGenomeArray(path, mapping=mapping, regions=regions)
        assert part._packed[0].n < whole._packed[0].n and part.sum() == whole.sum()
        a, b = whole.get_counts_batch(chains), part.get_counts_batch(chains)
        assert sum(x.sum() for x in a) > 0
        for x, y in zip(a, b):
            assert np.array_equal(x.view(np.uint64), y.view(np.uint64))
        assert np.array_equal(whole[chains[0][0]], part[chains[0][0]])


def _windows(ga, pa, strand, window_size):
    """(chrom, window start, genome-order vector) for every export window, chromosomes sorted --
    the iteration order of the reference's exporters (genome_array.py:990-1111)."""
    for chrom in sorted(ga.chroms()):
        size = ga.lengths()[chrom]
        for w0 in range(0, size, window_size):
            yield chrom, w0, ga.get(pa.GenomicSegment(chrom, w0, min(w0 + window_size, size), strand), roi_order=False)


def _expected_bedgraph(ga, pa, trackname, strand, window_size):
    """What to_bedgraph must write: per window, maximal runs of equal positive values as
    ``chrom start end value`` lines; windows without counts write nothing."""
    lines = ["track type=bedGraph name=%s\n" % trackname]
    for chrom, w0, vec in _windows(ga, pa, strand, window_size):
        if not vec.sum() > 0:
            continue
        edges = np.concatenate([[0], np.nonzero(np.diff(vec))[0] + 1, [len(vec)]])
        for lo, hi in zip(edges[:-1], edges[1:]):
            if vec[lo] > 0:
                lines.append("%s\t%s\t%s\t%s\n" % (chrom, w0 + lo, w0 + hi, vec[lo]))
    return "".join(lines)


def _expected_variable_step(ga, pa, trackname, strand, window_size):
    """What to_variable_step must write: a header per chromosome, then ``position(1-based) value``
    for every position with a count."""
    lines = ["track type=wiggle_0 name=%s\n" % trackname]
    last = None
    for chrom in sorted(ga.chroms()):
        lines.append("variableStep chrom=%s span=1\n" % chrom)
        for c2, w0, vec in _windows(ga, pa, strand, window_size):
            if c2 != chrom or not vec.sum() > 0:
                continue
            for i in np.flatnonzero(vec):
                lines.append("%s\t%s\n" % (w0 + i + 1, vec[i]))
    return "".join(lines)


def test_export_is_the_reference_window_loop(pa):
    """to_bedgraph / to_variable_step (one launch + GPU run-length encoding per chromosome) write,
    byte for byte, what the reference's window loop writes -- runs cut at window borders, int and
    float formatting -- for integer, normalised and center mappings and every strand."""
    import io
    from plastid_amd import synth
    from plastid_amd.engine import Engine
    genome, tx, reads, _ = synth.make_config("C2", scale=0.002, tx_scale=0.002)
    small = reads.subset(np.nonzero(reads.tid < 3)[0])
    names = list(small.references)
    lens = list(small.lengths)
    # shrink the unused contigs so the reference-style loop stays quick
    for t in range(3, len(lens)):
        lens[t] = 1000
    small = pa.PackedAlignments(small.tid, small.pos, small.alen, small.flags, small.nblk, small.blk_start, small.blk_len,
                                references=names, lengths=lens)
    for mapping, norm in ((pa.FivePrimeMapFactory(12), False), (pa.ThreePrimeMapFactory(0), True), (pa.CenterMapFactory(3), False)):
        ga = pa.BAMGenomeArray(small, mapping=mapping)
        ga.set_normalize(norm)
        for strand in ("+", "-", "."):
            for window in (100000, 7777):
                got = io.StringIO()
                ga.to_bedgraph(got, "t", strand, window_size=window)
                assert got.getvalue() == _expected_bedgraph(ga, pa, "t", strand, window), (mapping, norm, strand, window)
            got = io.StringIO()
            ga.to_variable_step(got, "t", strand)
            assert got.getvalue() == _expected_variable_step(ga, pa, "t", strand, 100000), (mapping, norm, strand)
    # the encoder itself against numpy, with and without a period
    eng = Engine(0)
    eng.set_alignments([small])
    pa.FivePrimeMapFactory(0)._configure(eng)
    size = lens[0]
    plan = eng.plan([0], [0], [size], [3], [0], np.ones(1, np.int8), [size], size, 1)
    vec = plan.count(np.int64)
    for period in (0, 1, 4096, 1000):
        starts, values = plan.rle(period)
        head = np.ones(size, bool)
        head[1:] = vec[1:] != vec[:-1]
        if period:
            head[::period] = True
        assert np.array_equal(starts, np.nonzero(head)[0]) and np.array_equal(values, vec[head]), period
    plan.close()
    eng.close()


def test_fused_region_statistics(pa, oracle):
    """count_in_regions == numpy.nansum(chain.get_masked_counts(ga)) per chain (counts_in_region.py:113-124)."""
    from plastid_amd import synth
    genome, tx, reads, _ = synth.make_config("C4", scale=0.0002, tx_scale=0.003)
    ga = pa.BAMGenomeArray(reads, mapping=pa.VariableFivePrimeMapFactory(synth.VARIABLE_OFFSETS))
    ga.add_filter("size", pa.SizeFilterFactory(25, 100))
    chains = tx.chains(limit=60)
    rng = np.random.default_rng(3)
    for c in chains[::2]:   # mask a random stretch of every other chain
        span = c.spanning_segment
        a = int(rng.integers(span.start, span.end))
        c.add_masks(pa.GenomicSegment(c.chrom, a, min(span.end, a + 200), c.strand))
    chains.append(pa.SegmentChain(pa.GenomicSegment("nope", 5, 50, "+")))   # unknown contig -> 0
    stats = ga.count_in_regions(chains)
    for i, c in enumerate(chains):
        exp = np.nansum(c.get_masked_counts(ga))
        assert stats["counts"][i] == exp, i
        assert stats["length"][i] == c.masked_length
        if c.masked_length:
            assert stats["counts_per_nucleotide"][i] == float(exp) / c.masked_length
            assert stats["rpkm"][i] == float(exp) / c.masked_length * (1000.0 * 1e6 / ga.sum())
    # stratified: one sum per read length
    ga.set_mapping(pa.StratifiedVariableFivePrimeMapFactory(synth.VARIABLE_OFFSETS, 25, 35))
    st2 = ga.count_in_regions(chains[:10])
    assert st2["counts"].shape == (10, 11)
    for i, c in enumerate(chains[:10]):
        assert np.array_equal(st2["counts"][i], np.nansum(c.get_masked_counts(ga), axis=-1))
    ga.set_mapping(pa.CenterMapFactory())
    with pytest.raises(TypeError):
        ga.count_in_regions(chains[:2])


@pytest.mark.parametrize("knobs", [
    {},                                                   # defaults
    {"PC_WORK_R": "1024", "PC_PILE": "1000000"},          # every dense window cut into sub-windows
    {"PC_WORK_R": "1024", "PC_PILE": "1024"},             # pile-up fallback: record slices merged via hist
    {"PC_TILE_G": "512", "PC_WORK_R": "2048"},            # small windows
    {"PC_NO_SMALL": "1"},                                 # no single-wave class
    {"PC_TILE_G": "768", "PC_WORK_R": "512", "PC_PILE": "100000"},  # odd window size, odd/even record ranges
])
def test_work_list_paths_vs_oracle(pa, oracle, knobs, monkeypatch):
    """Every scheduling path of the tile kernel (sub-windows, heavy/light/small classes, merged
    windows, 3 files -> FileView array path) gives the oracle's counts: dense pile-up + sparse
    annotation + whole-contig and '.' segments + gapped/long reads, all five... integer rules."""
    from plastid_amd import synth
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(11)
    genome, tx, reads, _ = synth.make_config("C2", scale=0.003, tx_scale=0.01)      # 300 k reads, 200 tx
    # a second, sparse file with spliced reads on the same contigs and a third tiny one
    _, tx4, r4, _ = synth.make_config("C4", scale=0.0001, tx_scale=0.002)
    names, lens = list(reads.references), list(reads.lengths)
    sparse = pa.PackedAlignments(r4.tid % len(names), r4.pos % 200000, r4.alen, r4.flags, r4.nblk,
                                 r4.blk_start, r4.blk_len, references=names, lengths=lens, validate=False)
    # rebuild runs consistently after the modulo: simplest is to drop gapped structure there
    order = np.lexsort((sparse.pos, sparse.tid))
    sparse = pa.PackedAlignments(sparse.tid[order], sparse.pos[order], sparse.alen[order], sparse.flags[order],
                                 np.minimum(sparse.nblk[order], 1), references=names, lengths=lens)
    third = reads.subset(np.arange(0, reads.n, 97))
    # segments: the transcripts, plus whole contigs on '.', plus tiny exons far apart
    p = tx.plan_arrays(rows=1)
    extra_tid = np.array([0, 1, 2, 3, 3], np.int32)
    extra_start = np.array([0, 0, 1000, 50000, 50300], np.int64)
    extra_end = np.array([lens[0], lens[1], 300000, 50030, 50310], np.int64)
    extra_strand = np.array([3, 3, 1, 2, 2], np.uint8)
    for files in ([reads], [reads, sparse, third]):
        for mapping in [("fiveprime", 12), ("threeprime", 3), ("variable", synth.VARIABLE_OFFSETS),
                        ("stratified", synth.VARIABLE_OFFSETS, 27, 31)]:
            eng = engine_for(pa, files, mapping)
            rows = eng.rows
            pp = tx.plan_arrays(rows=rows)
            base = pp["out_elems"]
            elen = extra_end - extra_start
            eoff = base + np.concatenate([[0], np.cumsum(elen * rows)[:-1]])
            tid = np.concatenate([pp["tid"], extra_tid]); start = np.concatenate([pp["start"], extra_start])
            end = np.concatenate([pp["end"], extra_end]); strand = np.concatenate([pp["strand"], extra_strand])
            out_off = np.concatenate([pp["out_off"], eoff]); step = np.concatenate([pp["out_step"], np.ones(5, np.int8)])
            rstride = np.concatenate([pp["row_stride"], elen])
            total = int(base + (elen * rows).sum())
            plan = eng.plan(tid, start, end, strand, out_off, step, rstride, total, rows)
            got = plan.count(np.int64)
            full = dict(tid=tid, start=start, end=end, strand=strand, out_off=out_off, out_step=step,
                        row_stride=rstride, out_elems=total)
            exp, warn = oracle_chain_outputs(oracle, files, spec_for(oracle, mapping), None, full, rows, np.int64)
            assert np.array_equal(got, exp), (knobs, len(files), mapping)
            assert np.array_equal(plan.warn_flags(), warn)
            plan.close()
            eng.close()


# ---------------------------------------------------------------------------- export + region statistics vs the reference's own output
def _export_golden():
    import json
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "export_regions.npz"))
    man = json.loads(str(z["manifest"]))
    return man, {k[4:]: z[k] for k in z.files if k.startswith("aln_")}


def _export_factory(pa, spec):
    from tests import golden_util as gu
    k = spec["kind"]
    if k == "fiveprime":
        return pa.FivePrimeMapFactory(spec["param"])
    if k == "threeprime":
        return pa.ThreePrimeMapFactory(spec["param"])
    if k == "center":
        return pa.CenterMapFactory(spec["param"])
    return pa.VariableFivePrimeMapFactory(gu.offset_dict_of(spec))


def test_export_text_equals_reference_output(pa):
    """to_bedgraph / to_variable_step write, byte for byte, the text the REFERENCE's own
    ``BAMGenomeArray.to_bedgraph`` / ``to_variable_step`` (genome_array.py:990-1111) wrote for the
    same reads (tests/golden/export_regions.npz, made by make_export_golden.py from the scratch
    reference): integer, normalised, center and variable rules, every strand, two window sizes,
    extra track keywords."""
    import io
    man, aln = _export_golden()
    packed = pa.PackedAlignments(aln["tid"], aln["pos"], aln["alen"], aln["flags"], aln["nblk"], aln["blk_start"], aln["blk_len"],
                                 references=man["references"], lengths=man["lengths"], mapped=man["mapped"])
    assert len(man["exports"]) == 36
    for ex in man["exports"]:
        ga = pa.BAMGenomeArray(packed, mapping=_export_factory(pa, ex["spec"]))
        ga.set_normalize(ex["normalize"])
        fh = io.StringIO()
        if ex["what"] == "bedgraph":
            ga.to_bedgraph(fh, "trk", ex["strand"], window_size=ex["window"], **ex["kwargs"])
        else:
            ga.to_variable_step(fh, "trk", ex["strand"], window_size=ex["window"], **ex["kwargs"])
        assert fh.getvalue() == ex["text"], (ex["what"], ex["spec"], ex["normalize"], ex["strand"], ex["window"])


def test_region_statistics_equal_reference_output(pa):
    """count_in_regions reproduces the numbers AND the formatted lines of the reference's
    ``bin/counts_in_region.py:113-124`` loop (masked sums, masked length, reads per nucleotide, RPKM),
    including a fully masked chain (nan), an unknown chromosome and the CLI size filter."""
    from tests import golden_util as gu
    man, aln = _export_golden()
    packed = pa.PackedAlignments(aln["tid"], aln["pos"], aln["alen"], aln["flags"], aln["nblk"], aln["blk_start"], aln["blk_len"],
                                 references=man["references"], lengths=man["lengths"], mapped=man["mapped"])
    chains = []
    for c in man["chains"]:
        ch = pa.SegmentChain(*[pa.GenomicSegment(c["chrom"], s, e, c["strand"]) for s, e in c["segments"]], ID=c["name"])
        ch.add_masks(*[pa.GenomicSegment(c["chrom"], s, e, c["strand"]) for s, e in c["masks"]])
        chains.append(ch)
    for tab in man["regions"]:
        ga = pa.BAMGenomeArray(packed, mapping=_export_factory(pa, tab["spec"]))
        if tab["size_filter"] is not None:
            ga.add_filter("size", pa.SizeFilterFactory(tab["size_filter"][0], tab["size_filter"][1]))
        assert float(ga.sum()) == tab["sum"]
        st = ga.count_in_regions(chains)
        lines = ga.counts_in_region_lines(chains, st, names=[c["name"] for c in man["chains"]])
        for i, row in enumerate(tab["rows"]):
            assert int(st["length"][i]) == row["length"]
            if row["length"] == 0:
                assert np.isnan(st["counts_per_nucleotide"][i]) and np.isnan(st["rpkm"][i])
            else:
                assert float(st["counts"][i]) == row["counts"]
                assert st["counts_per_nucleotide"][i] == row["rpnt"] and st["rpkm"][i] == row["rpkm"]
            assert lines[i] == row["line"], (tab["spec"], i)


def test_custom_filters_are_evaluated_like_the_reference(pa, oracle):
    """An arbitrary filter callable sees what the reference would pass to it (genome_array.py:800-820):
    only reads ``fetch`` returns for the queried region, only the strand the region keeps, and -- here --
    each read at most once; the counts equal the oracle's on the reads the filter keeps."""
    from plastid_amd import synth
    genome, tx, reads, _ = synth.make_config("C2", scale=0.0005, tx_scale=0.005)
    ga = pa.BAMGenomeArray(reads, mapping=pa.FivePrimeMapFactory(5))
    seen = []

    def keep_even_starts(read):
        seen.append((read.index, read.is_reverse))
        return read.positions[0] % 2 == 0
    ga.add_filter("even", keep_even_starts)
    chrom = reads.references[3]
    seg = pa.GenomicSegment(chrom, 1000, 9000, "+")
    got = ga[seg]
    fetched = reads.fetch_indices(chrom, 1000, 9000)
    fwd = fetched[(reads.flags[fetched] & 1) == 0]
    assert sorted(i for i, _ in seen) == list(fwd) and not any(rev for _, rev in seen)   # fetched, forward only, once each
    n_first = len(seen)
    assert np.array_equal(ga[seg], got) and len(seen) == n_first                          # cached verdicts
    # expected: the oracle on the records the filter keeps
    keep = np.ones(reads.n, bool)
    keep[reads.pos % 2 == 1] = False
    sub = reads.subset(np.nonzero(keep)[0])
    aln = aln_dict([sub])
    arrays, _ = oracle.count_segments(aln, oracle.mapping_spec("fiveprime", 5), [3], [1000], [9000], [1])
    assert np.array_equal(got, arrays[0])
    # an overlapping '-' query asks about the reverse reads only, and only the new ones
    seg2 = pa.GenomicSegment(chrom, 5000, 12000, "-")
    got2 = ga.get(seg2, roi_order=False)
    new = seen[n_first:]
    f2 = reads.fetch_indices(chrom, 5000, 12000)
    assert sorted(i for i, _ in new) == list(f2[(reads.flags[f2] & 1) == 1]) and all(rev for _, rev in new)
    arrays, _ = oracle.count_segments(aln, oracle.mapping_spec("fiveprime", 5), [3], [5000], [12000], [2])
    assert np.array_equal(got2, arrays[0])
    # removing the filter restores every read
    ga.remove_filter("even")
    arrays, _ = oracle.count_segments(aln_dict([reads]), oracle.mapping_spec("fiveprime", 5), [3], [1000], [9000], [1])
    assert np.array_equal(ga[seg], arrays[0])


def test_exact_grid_guard_and_cache_invalidation(pa, monkeypatch):
    """Large plans launch exactly the work-item counts a previous count of the plan left.  (1) What invalidates
    those counts -- other alignment files, re-read knobs -- must drop the cache: results stay right.  (2) Should a
    count ever queue MORE items than the cached counts launched, the last kernel of the call notices and the read-back
    reports it instead of returning incomplete vectors (PC_TEST_STALE_COUNTS makes the cache one item short)."""
    from plastid_amd import synth
    from plastid_amd.engine import Engine
    from plastid_amd.exceptions import EngineError
    genome, tx, reads, mapping = synth.make_config("C4", scale=0.004, tx_scale=0.25)
    p = tx.plan_arrays(rows=1)
    half = reads.slice(0, reads.n // 2)

    def count_twice(eng, plan):
        a = plan.count(np.int64)
        eng.sync()
        b = plan.count(np.int64)            # by now the work counts of the first count have arrived: exact grids
        assert np.array_equal(a, b)
        return b
    eng = Engine(0)
    eng.set_alignments([reads])
    synth.mapping_factory(mapping)._configure(eng)
    plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
    assert plan.tiles >= 4096
    full = count_twice(eng, plan)
    eng.set_alignments([half])              # same plan, other records: the cached counts must not be used
    part = count_twice(eng, plan)
    eng.set_alignments([reads])
    monkeypatch.setenv("PC_WORK_R", "2048")
    eng.reload_knobs()                      # other work-item size: more items than before
    assert np.array_equal(count_twice(eng, plan), full)
    monkeypatch.delenv("PC_WORK_R")
    eng.reload_knobs()
    # the work lists live in the plan and are reused from count to count; another mapping rule on the same plan (another
    # halo in front of the windows) must rebuild them, and going back must rebuild them again
    from plastid_amd import map_factories as mf
    e_ref = Engine(0)
    e_ref.set_alignments([reads])
    mf.ThreePrimeMapFactory(3)._configure(e_ref)
    plan_ref = e_ref.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
    three = plan_ref.count(np.int64)
    plan_ref.close()
    e_ref.close()
    mf.ThreePrimeMapFactory(3)._configure(eng)
    assert np.array_equal(count_twice(eng, plan), three)
    synth.mapping_factory(mapping)._configure(eng)
    assert np.array_equal(count_twice(eng, plan), full)
    assert np.array_equal(count_twice(eng, plan), full)
    plan.close()
    e2 = Engine(0)
    e2.set_alignments([half])
    synth.mapping_factory(mapping)._configure(e2)
    plan2 = e2.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
    assert np.array_equal(plan2.count(np.int64), part)
    plan2.close()
    e2.close()
    eng.close()
    # (2) the guard
    monkeypatch.setenv("PC_TEST_STALE_COUNTS", "1")
    e3 = Engine(0)
    monkeypatch.delenv("PC_TEST_STALE_COUNTS")
    e3.set_alignments([reads])
    synth.mapping_factory(mapping)._configure(e3)
    plan3 = e3.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
    assert np.array_equal(plan3.count(np.int64), full)       # first count: the whole capacity is launched
    e3.sync()
    plan3.launch(np.int64)                                   # second: "cached" counts, one light item short
    with pytest.raises(EngineError):
        plan3.read()
    assert np.array_equal(plan3.count(np.int64), full)       # the cache was dropped: whole capacity again
    plan3.close()
    e3.close()


@pytest.mark.gpu
@pytest.mark.parametrize("group", ["quirks", "random_reads", "wide_reads"])
def test_golden_reads_out_through_the_batch_call(pa, group):
    """`reads_out` of every golden segment query (the reference's own lists), reproduced by ONE pass per case over all
    its segments: `BAMGenomeArray.get_reads_batch` -> `pc_mapped_reads_batch` (a CSR over (segment, file) pairs)."""
    g = gu.load(group)
    nq = 0
    for case in g.cases:
        if case["kind"] != "ga":
            continue
        files = files_of(pa, g, case)
        ga = pa.BAMGenomeArray(files, mapping=factory_of(pa, case["spec"]))
        if case.get("size_filter"):
            ga.add_filter("size", pa.SizeFilterFactory(min=case["size_filter"][0], max=case["size_filter"][1]))
        offs = np.cumsum([0] + [f.n for f in files])
        qs = [q for q in case["queries"] if q["type"] == "segment"]
        segs = [pa.GenomicSegment(q["chrom"], q["start"], q["end"], q["strand"]) for q in qs]
        if not segs:
            continue
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            lists = ga.get_reads_batch(segs)
            idx_lists = ga.get_reads_batch(segs, as_indices=True)
        assert len(lists) == len(segs)
        for q, reads, parts in zip(qs, lists, idx_lists):
            nq += 1
            got = [offs[files.index(r.source)] + r.index for r in reads]
            assert got == list(g[q["reads_out"]]), (case["spec"], q)
            flat = [int(offs[f] + i) for f, idx in parts for i in idx]
            assert flat == got
    assert nq >= 60


@pytest.mark.gpu
def test_reads_out_of_twenty_thousand_segments_in_one_pass(pa, oracle):
    """20 000 exon-sized segments over a spliced, two-file data set: the batch lists equal the oracle's `mapped`
    masks (fetch order, file-major) for a sample of segments and the per-segment engine path for another."""
    from plastid_amd import synth
    from plastid_amd.packing import concat_file_major
    genome, tx, reads, _ = synth.make_config("C4", scale=0.0004, tx_scale=0.05)
    halves = [reads.subset(np.arange(k, reads.n, 2)) for k in (0, 1)]
    ga = pa.BAMGenomeArray(halves, mapping=pa.FivePrimeMapFactory(12))
    p = tx.plan_arrays(rows=1)
    nseg = min(20000, len(p["tid"]))
    names = list(reads.references)
    strand = {1: "+", 2: "-", 3: "."}
    segs = [pa.GenomicSegment(names[int(p["tid"][s])], int(p["start"][s]), int(p["end"][s]), strand[int(p["strand"][s]) & 3]) for s in range(nseg)]
    parts = ga.get_reads_batch(segs, as_indices=True)
    assert len(parts) == nseg
    aln = concat_file_major(ga._packed)
    spec = oracle.mapping_spec("fiveprime", 12)
    rng = np.random.default_rng(3)
    pick = rng.choice(nseg, 60, replace=False)
    offs = np.cumsum([0] + [f.n for f in ga._packed])
    _, _, mapped = oracle.count_segments(aln, spec, p["tid"][pick], p["start"][pick], p["end"][pick], p["strand"][pick], want_mapped=True)
    nonempty = 0
    for k, s in enumerate(pick):
        got = [int(offs[f] + i) for f, idx in parts[s] for i in idx]
        assert got == list(np.nonzero(mapped[k])[0]), s
        nonempty += bool(got)
    assert nonempty > 10
    for s in rng.choice(nseg, 15, replace=False):          # the per-segment path of the mirror
        reads_one = ga.get_reads(segs[int(s)])
        got = [(ga._packed.index(r.source), r.index) for r in reads_one]
        assert got == [(f, int(i)) for f, idx in parts[int(s)] for i in idx]


@pytest.mark.gpu
def test_one_window_plans_count_in_a_single_launch(pa, oracle):
    """A plan of one window over one file (`ga[segment]`) takes the fused path: the tile kernel looks its record ranges
    up itself.  Its counts equal the oracle's and those of the general path (work lists, window classes, merge pass;
    forced with PC_NO_SINGLE) -- for every rule, forward / reversed / summed layouts, a spliced data set, a pile-up of
    more records than one work item of the general path takes, an empty window and a window beyond the last record."""
    from plastid_amd import synth
    from plastid_amd.packing import PackedAlignments
    genome, tx, reads, _ = synth.make_config("C4", scale=0.0005, tx_scale=0.01)
    # a pile-up: 120 000 extra reads of mixed lengths at a handful of positions of the first contig
    rng = np.random.default_rng(11)
    npile = 120000
    ppos = np.sort(rng.choice(np.arange(5000, 5040), npile)).astype(np.int32)
    plen = rng.integers(25, 35, npile).astype(np.uint16)
    extra = PackedAlignments(np.zeros(npile, np.int32), ppos, plen, rng.integers(0, 2, npile).astype(np.uint8), np.ones(npile, np.uint8),
                             references=reads.references, lengths=reads.lengths)
    first = reads.slice(0, int(np.searchsorted(reads.tid, 1)))
    # (the pile-up file holds the first contig's single-run reads + the pile, in coordinate order)
    order = np.argsort(np.concatenate([first.pos, extra.pos]), kind="stable")
    cat = lambda k: np.concatenate([getattr(first, k), getattr(extra, k)])[order]   # noqa: E731
    keep = cat("nblk") == 1
    dense = PackedAlignments(cat("tid")[keep], cat("pos")[keep], cat("alen")[keep], cat("flags")[keep], cat("nblk")[keep],
                             references=reads.references, lengths=reads.lengths)
    names = list(reads.references)
    cases = []
    p = tx.plan_arrays(rows=1)
    for s in rng.choice(len(p["tid"]), 12, replace=False):
        cases.append((reads, int(p["tid"][s]), int(p["start"][s]), int(p["end"][s]), int(p["strand"][s])))
    cases += [(dense, 0, 4900, 5200, 1), (dense, 0, 4990, 5100, 2), (dense, 0, 5000, 5001, 3),
              (reads, 0, 10, 10, 1), (reads, len(names) - 1, int(reads.lengths[-1]) + 5000, int(reads.lengths[-1]) + 5300, 3)]
    mappings = [("fiveprime", 12), ("threeprime", 3), ("variable", synth.VARIABLE_OFFSETS), ("stratified", synth.VARIABLE_OFFSETS, 27, 31)]
    saved = os.environ.pop("PC_NO_SINGLE", None)
    try:
        for files in (reads, dense):
            for mapping in mappings:
                eng = engine_for(pa, [files], mapping)
                rows = eng.rows
                spec = spec_for(oracle, mapping)
                aln = aln_dict([files])
                for (fl, tid, start, end, strand) in cases:
                    if fl is not files:
                        continue
                    n = end - start
                    for step in (1, -1, 0):
                        off = [0] if step >= 0 else [max(n - 1, 0)]
                        stride = [max(n, 1)] if step else [1]
                        elems = max(rows * n, 1) if step else rows
                        got = {}
                        for path in ("single", "general"):
                            if path == "general":
                                os.environ["PC_NO_SINGLE"] = "1"
                            else:
                                os.environ.pop("PC_NO_SINGLE", None)
                            eng.reload_knobs()
                            plan = eng.plan([tid], [start], [end], [strand], off, [step], stride, elems, rows)
                            got[path] = plan.count(np.int64).copy()
                            plan.close()
                        assert np.array_equal(got["single"], got["general"]), (mapping, tid, start, end, strand, step)
                        arrays, _ = oracle.count_segments(aln, spec, [tid], [start], [end], [strand])
                        a2 = arrays[0].reshape(rows, n)
                        if step == 0:
                            exp = a2.sum(axis=1)
                        else:
                            exp = np.zeros(elems, np.int64)
                            for r in range(rows):
                                exp[off[0] + r * stride[0] + step * np.arange(n)] = a2[r]
                        assert np.array_equal(got["single"][:len(exp)], exp), (mapping, tid, start, end, strand, step)
                        # the same window in ONE call (pc_query_segment: argument-borne window, counts through page-locked memory)
                        if step != 0 and mapping[0] != "stratified" and 0 < n <= 4096 and 0 <= start:
                            for dt in (np.int64, np.float64):
                                q = eng.query_segment(tid, start, end, strand, step < 0, dt)
                                assert q.dtype == dt and np.array_equal(q, exp.astype(dt)), (mapping, tid, start, end, strand, step)
                eng.close()
    finally:
        os.environ.pop("PC_NO_SINGLE", None)
        if saved is not None:
            os.environ["PC_NO_SINGLE"] = saved


def test_single_segment_queries_in_one_call(pa, oracle):
    """``ga[segment]`` through the mirror: segments of up to 4 096 positions over one file take ``pc_query_segment`` (no
    plan object); the vectors equal those of the plan path (PC_NO_SINGLE) -- point rules, both strands and '.', roi_order,
    a size filter, a FLAG / MAPQ filter, normalisation -- and a rule that can warn keeps to the plan path and warns."""
    from plastid_amd import synth
    genome, tx, reads, _ = synth.make_config("C2", scale=0.002, tx_scale=0.02)
    rng = np.random.default_rng(3)
    reads.flag16 = (np.where(reads.flags & 1, 0x10, 0) | np.where(rng.random(reads.n) < 0.3, 0x100, 0)).astype(np.uint16)
    reads.mapq = rng.integers(0, 40, reads.n).astype(np.uint8)
    ga = pa.BAMGenomeArray(reads, mapping=pa.FivePrimeMapFactory(12))
    segs = [c[0] for c in tx.chains(limit=60)]
    segs += [pa.GenomicSegment(s.chrom, s.start, s.end, ".") for s in segs[:10]]
    saved = os.environ.pop("PC_NO_SINGLE", None)
    try:
        calls = []
        real = ga._engine.query_segment
        ga._engine.query_segment = lambda *a: (calls.append(1), real(*a))[1]
        for step, setup in enumerate((lambda: None, lambda: ga.set_mapping(pa.ThreePrimeMapFactory(0)),
                                      lambda: ga.add_filter("size", pa.SizeFilterFactory(27, 32)),
                                      lambda: ga.add_filter("flags", pa.FlagFilterFactory(exclude="is_secondary", min_mapq=10)),
                                      lambda: ga.set_normalize(True),
                                      lambda: ga.set_mapping(pa.VariableFivePrimeMapFactory(synth.VARIABLE_OFFSETS)))):
            setup()
            for s_ in segs:
                for ro in (True, False):
                    os.environ.pop("PC_NO_SINGLE", None)
                    n0 = len(calls)
                    fast = ga.get(s_, roi_order=ro)
                    assert len(calls) == n0 + 1 or len(s_) > 4096
                    os.environ["PC_NO_SINGLE"] = "1"
                    slow = ga.get(s_, roi_order=ro)
                    assert fast.dtype == slow.dtype and np.array_equal(np.asarray(fast).view(np.uint64), np.asarray(slow).view(np.uint64)), (step, s_)
        assert sum(int(np.asarray(ga[s_]).sum() > 0) for s_ in segs) > 20
        # a rule that can warn (offset beyond the shortest reads) keeps to the plan path and warns as ever
        os.environ.pop("PC_NO_SINGLE", None)
        ga.set_normalize(False)
        ga.set_mapping(pa.FivePrimeMapFactory(int(reads.alen.max())))
        n0 = len(calls)
        dense = max(segs, key=lambda s_: np.asarray(ga.get(s_)).size)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            for s_ in segs:
                ga[s_]
        assert len(calls) == n0 and any(issubclass(x.category, UserWarning) for x in w) and dense is not None
    finally:
        os.environ.pop("PC_NO_SINGLE", None)
        if saved is not None:
            os.environ["PC_NO_SINGLE"] = saved


def test_sixteen_bit_bins_do_not_overflow(pa, oracle):
    """The stratified rule bins into 16-bit LDS counters, two positions per word (k_hist_point, B16); what keeps a bin
    from carrying into its neighbour is k_tile_ranges cutting or merging every work item that could add 65 536 times.
    200 000 reads of ONE length at ONE position (an rRNA pile-up), 70 000 at the neighbouring position that shares its
    LDS word, 100 000 on the first position of the next window and 90 000 reverse reads of another length: every element
    against the oracle -- one file, the same reads dealt into two files (joint windows), '+', '-', '.', segments that end
    on and run across the window edge, forward / reversed / summed layouts.  A library compiled with
    -DPC_KMAX16=1000000000 (no guard) fails this test."""
    from plastid_amd import synth
    names, lens = ["a", "b"], [60000, 20000]
    G = 768                                  # window of a multi-row plan (choose_window)
    edge = G * 10
    off28 = synth.VARIABLE_OFFSETS[28]       # forward rule: read.positions[13] of a 28-mer
    off30 = synth.VARIABLE_OFFSETS[30]
    rng = np.random.default_rng(16)
    parts = [
        (np.full(200000, edge - 2 - off28), 28, False),          # -> position edge - 2 (even: low half of its word)
        (np.full(70000, edge - 1 - off28), 28, False),           # -> edge - 1: the high half of the same word
        (np.full(100000, edge - off28), 28, False),              # -> edge: first position of the next window
        (np.full(90000, edge - 3 - (30 - 1 - off30)), 30, True),  # reverse rule -> edge - 3
        (rng.integers(0, 50000, 30000), 28, False),              # background, both strands
        (rng.integers(0, 50000, 30000), 31, True),
    ]
    pos = np.concatenate([p_[0] for p_ in parts]).astype(np.int64)
    alen = np.concatenate([np.full(len(p_[0]), p_[1]) for p_ in parts])
    rev = np.concatenate([np.full(len(p_[0]), p_[2]) for p_ in parts])
    order = np.argsort(pos, kind="stable")
    pos, alen, rev = pos[order], alen[order], rev[order]
    whole = pa.PackedAlignments.from_ungapped(0, pos, alen, rev, references=names, lengths=lens)
    even, odd = np.arange(0, len(pos), 2), np.arange(1, len(pos), 2)
    halves = [pa.PackedAlignments.from_ungapped(0, pos[k], alen[k], rev[k], references=names, lengths=lens) for k in (even, odd)]
    seg_start = np.array([edge - 80, edge - 2, edge - 300, 0, edge - 3, edge - 1, edge - 80], np.int64)
    seg_end = np.array([edge + 70, edge, edge - 1, 50000, edge + 1, edge + 1, edge + 70], np.int64)
    seg_tid = np.zeros(len(seg_start), np.int32)
    seg_strand = np.array([1, 1, 1, 3, 2, 3, 2], np.uint8)      # PC_STRAND_FWD / REV / UNS
    mapping = ("stratified", synth.VARIABLE_OFFSETS, 25, 35)
    spec = spec_for(oracle, mapping)
    n = seg_end - seg_start
    for files in ([whole], halves):
        eng = engine_for(pa, files, mapping)
        rows = eng.rows
        aln = aln_dict(files)
        arrays, _ = oracle.count_segments(aln, spec, seg_tid, seg_start, seg_end, seg_strand)
        assert max(int(a.max()) for a in arrays) >= 200000       # the pile really is deeper than a 16-bit bin
        for step in (1, -1, 0):
            if step:
                base = np.concatenate([[0], np.cumsum(n * rows)[:-1]])
                out_off = base if step > 0 else base + n - 1
                stride, elems = n, int((n * rows).sum())
            else:
                out_off = np.arange(len(n), dtype=np.int64) * rows
                stride, elems = np.ones(len(n), np.int64), len(n) * rows
            plan = eng.plan(seg_tid, seg_start, seg_end, seg_strand, out_off, np.full(len(n), step, np.int8), stride, elems, rows)
            got = plan.count(np.int64)
            exp = np.zeros(elems, np.int64)
            for s, arr in enumerate(arrays):
                a2 = arr.reshape(rows, int(n[s]))
                for r in range(rows):
                    if step:
                        exp[out_off[s] + r * stride[s] + step * np.arange(int(n[s]))] = a2[r]
                    else:
                        exp[out_off[s] + r] = a2[r].sum()
            assert np.array_equal(got, exp), (len(files), step, np.nonzero(got != exp)[0][:8])
            plan.close()
        # plans of ONE window (what `ga[segment]` builds): the stratified rule never takes the one-launch path of
        # pc_query_segment / SINGLE (pc_count: `single` excludes PC_MAP_STRAT5), so these go through the same guarded lists
        for s in (1, 4, 5):
            one = slice(s, s + 1)
            plan = eng.plan(seg_tid[one], seg_start[one], seg_end[one], seg_strand[one], np.zeros(1, np.int64), np.ones(1, np.int8),
                            n[one], int(n[s]) * rows, rows)
            assert np.array_equal(plan.count(np.int64), arrays[s].reshape(-1)), (len(files), s)
            plan.close()
        eng.close()
