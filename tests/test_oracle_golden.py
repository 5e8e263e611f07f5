"""The oracle (oracle/plastid_oracle.c + oracle/oracle.py) against golden vectors
generated from the reference itself.  CPU only."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402
from tests import golden_util as gu  # noqa: E402


def spec_of(case):
    s = case["spec"]
    return oracle.mapping_spec(s["kind"], s.get("param", 0), gu.offset_dict_of(s),
                               s.get("min_len", 25), s.get("max_len", 35),
                               size_filter=tuple(case["size_filter"]) if case.get("size_filter") else None)


def same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)


def test_kat_mapfn_direct():
    """test_map_factories.py closed-form vectors, via direct map-function calls."""
    g = gu.load("kat_map_factories")
    assert len(g.cases) == 24
    for case in g.cases:
        aln = g.aln(case)
        spec = spec_of(case)
        arrays, warn, mapped = oracle.count_segments(
            aln, spec, [0], [case["start"]], [case["end"]], [gu.STRAND_CODE[case["strand"]]], want_mapped=True)
        exp = g[case["expected"]]
        assert same(arrays[0], exp), case["spec"]
        assert np.array_equal(np.nonzero(mapped[0])[0], g[case["reads_out"]])
        assert bool(warn[0]) == case["warned"]


@pytest.mark.parametrize("group", ["quirks", "random_reads", "chains", "wide_reads"])
def test_ga_cases(group):
    g = gu.load(group)
    nq = 0
    for case in g.cases:
        if case["kind"] != "ga":
            continue
        aln = g.aln(case)
        spec = spec_of(case)
        norm = case["sum"] if case["normalize"] else None
        for q in case["queries"]:
            nq += 1
            tid = gu.tid_of(case, q["chrom"])
            if q["type"] == "segment":
                exp = g[q["expected"]]
                got = oracle.get_segment(aln, spec, tid, q["start"], q["end"], q["strand"],
                                         roi_order=q["roi_order"], normalize_sum=norm, known_chrom=tid >= 0)
                assert same(got, exp), (case["spec"], q)
                if tid >= 0:
                    _, warn, mapped = oracle.count_segments(
                        aln, spec, [tid], [q["start"]], [q["end"]], [gu.STRAND_CODE[q["strand"]]],
                        want_mapped=True)
                    assert np.array_equal(np.nonzero(mapped[0])[0], g[q["reads_out"]]), (case["spec"], q)
                    assert bool(warn[0]) == q["warned"], (case["spec"], q)
            else:
                segs = [tuple(x) for x in q["merged_segments"]]
                if q.get("raises"):
                    with pytest.raises(ValueError):
                        if tid < 0:
                            # unknown chrom + stratified: (1, rows) cannot broadcast (Q9)
                            arrs = [oracle.get_segment(aln, spec, tid, s, e, q["strand"], False, norm, False)
                                    for s, e in segs]
                            out = np.empty(list(arrs[0].shape[:-1]) + [q["length"]])
                            out[..., 0:segs[0][1] - segs[0][0]] = arrs[0]
                    continue
                if tid < 0:
                    continue  # broadcast quirk covered at the product level
                got = oracle.chain_get_counts(aln, spec, tid, segs, q["strand"],
                                              stranded=q.get("stranded", True), normalize_sum=norm)
                assert same(got, g[q["expected"]]), (case["spec"], q)
                m = oracle.chain_get_masked_counts(aln, spec, tid, segs, q["strand"],
                                                   [tuple(x) for x in q.get("mask_segments", [])],
                                                   normalize_sum=norm)
                assert same(np.ma.getdata(m), g[q["masked_data"]])
                assert same(np.ma.getmaskarray(m), g[q["masked_mask"]])
    assert nq > (100 if group != "wide_reads" else 70)


def flag_excluded(aln, req, exc, mq, max_nh=0):
    """The alignments WITHOUT the reads a FLAG / MAPQ filter drops -- (flag & require) == require, (flag & exclude) == 0
    and mapq >= min_mapq keeps -- as the reference's fetch loop leaves them (genome_array.py:819-820); also the mask of
    the dropped reads and the original index of every kept one."""
    f = aln["flag16"].astype(np.int64)
    drop = ((f & req) != req) | ((f & exc) != 0) | (aln["mapq"] < mq)
    if max_nh:   # ... and read.has_tag("NH") and read.get_tag("NH") <= max_nh
        drop |= (aln["nh"] == 0) | (aln["nh"] > max_nh)
    keep = np.nonzero(~drop)[0]
    nb = aln["nblk"].astype(np.int64)
    cnt = np.where(nb >= 2, nb, 0)
    off = np.concatenate([[0], np.cumsum(cnt)])
    runs = np.concatenate([np.arange(off[i], off[i + 1]) for i in keep]).astype(np.int64) if len(keep) else np.zeros(0, np.int64)
    out = {k: aln[k][keep] for k in ("tid", "pos", "alen", "flags", "nblk", "file_id")}
    out["blk_start"], out["blk_len"] = aln["blk_start"][runs], aln["blk_len"][runs]
    return out, drop, keep


def test_flag_and_mapq_filters():
    """Filters on read.flag / read.mapping_quality / the is_* properties (genome_array.py:697-722, 819-820), run by
    the reference as plain callables (tests/golden/make_flag_golden.py): the oracle on the reads the filter keeps gives the reference's vectors and reads_out for all five rules, with a size filter and normalised."""
    g = gu.load("flag_filters")
    nq, ndrop = 0, 0
    for case in g.cases:
        aln, drop, keep = flag_excluded(g.aln(case), *case["filter"])
        ndrop += int(drop.sum())
        spec = spec_of(case)
        norm = case["sum"] if case["normalize"] else None
        for q in case["queries"]:
            nq += 1
            tid = gu.tid_of(case, q["chrom"])
            if q["type"] == "segment":
                got = oracle.get_segment(aln, spec, tid, q["start"], q["end"], q["strand"], roi_order=True, normalize_sum=norm)
                assert same(got, g[q["expected"]]), (case["spec"], case["filter_name"], q)
                _, warn, mapped = oracle.count_segments(aln, spec, [tid], [q["start"]], [q["end"]], [gu.STRAND_CODE[q["strand"]]],
                                                        want_mapped=True)
                assert np.array_equal(keep[np.nonzero(mapped[0])[0]], g[q["reads_out"]]), (case["spec"], q)
                assert bool(warn[0]) == q["warned"]
            else:
                got = oracle.chain_get_counts(aln, spec, tid, [tuple(x) for x in q["segments"]], q["strand"], normalize_sum=norm)
                assert same(got, g[q["expected"]]), (case["spec"], case["filter_name"], q)
    assert nq > 800 and ndrop > 10000
    assert any(g[q["expected"]].sum() == 0 for c in g.cases for q in c["queries"])     # the filter nothing passes


def test_nh_filters():
    """Filters on the NH:i tag (``read.has_tag("NH") and read.get_tag("NH") <= k``, the unique-mapper test; run by the
    reference as plain callables: tests/golden/make_nh_golden.py): the oracle on the reads the filter keeps gives the
    reference's vectors and reads_out for all five rules, with a size filter and normalised."""
    g = gu.load("nh_filters")
    nq, ndrop = 0, 0
    for case in g.cases:
        aln, drop, keep = flag_excluded(g.aln(case), *case["filter"])
        ndrop += int(drop.sum())
        spec = spec_of(case)
        norm = case["sum"] if case["normalize"] else None
        for q in case["queries"]:
            nq += 1
            tid = gu.tid_of(case, q["chrom"])
            if q["type"] == "segment":
                got = oracle.get_segment(aln, spec, tid, q["start"], q["end"], q["strand"], roi_order=True, normalize_sum=norm)
                assert same(got, g[q["expected"]]), (case["spec"], case["filter_name"], q)
                _, warn, mapped = oracle.count_segments(aln, spec, [tid], [q["start"]], [q["end"]], [gu.STRAND_CODE[q["strand"]]],
                                                        want_mapped=True)
                assert np.array_equal(keep[np.nonzero(mapped[0])[0]], g[q["reads_out"]]), (case["spec"], q)
                assert bool(warn[0]) == q["warned"]
            else:
                got = oracle.chain_get_counts(aln, spec, tid, [tuple(x) for x in q["segments"]], q["strand"], normalize_sum=norm)
                assert same(got, g[q["expected"]]), (case["spec"], case["filter_name"], q)
    assert nq > 600 and ndrop > 8000


def test_offset_tables():
    g = gu.load("offset_tables")
    for case in g.cases:
        if case["kind"] != "table":
            continue
        od = {(k if k == "default" else int(k)): v for k, v in case["offset_dict"].items()}
        fw, rc = oracle.variable_offset_tables(od)
        efw, erc = g[case["fw"]], g[case["rc"]]
        n = len(efw)
        # behavioural probe: a lone L-mer at 0 lands on fw[L] ('+') / rc[L] ('-'), or is dropped (-1)
        assert np.array_equal(fw[1:n], efw[1:]), od
        assert np.array_equal(rc[1:n], erc[1:]), od


def test_hand_checked_cigars():
    g = gu.load("quirks")
    case = [c for c in g.cases if c["kind"] == "hand_cigars"][0]
    from plastid_amd.packing import parse_cigar_string
    for i in g["hand_indices"]:
        runs, L = oracle.cigar_to_runs(int(g["hand_pos"][i]), parse_cigar_string(case["cigars"][i]))
        pos = [p for s, n in runs for p in range(s, s + n)]
        assert pos == list(g["hand_positions_%d" % i]) and L == len(pos)
