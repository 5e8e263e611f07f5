"""Length-stratified consumers (psite / phase_by_size): host logic on CPU against the reference's
``psite.do_count`` golden fixture; the counting itself on the GPU (marked)."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import plastid_amd as pa  # noqa: E402
from plastid_amd import stratified  # noqa: E402


def _golden():
    z = np.load(os.path.join(HERE, "golden", "psite_do_count.npz"), allow_pickle=False)
    return z, json.loads(str(z["meta"]))


def _raw_from_fixture(z, ci, case):
    raw = {}
    for k in range(case["min_len"], case["max_len"] + 1):
        raw[k] = np.ma.MaskedArray(z["c%d_raw_%d" % (ci, k)], mask=z["c%d_rawmask_%d" % (ci, k)], dtype=float)
    return raw


def test_stratify_mapping_tables():
    fac, valid, nrows = stratified.stratify_mapping(pa.FivePrimeMapFactory(16), 14, 19)
    assert nrows == 6 and valid.tolist() == [False, False, False, True, True, True]
    assert [int(fac.forward_offsets[L]) for L in (17, 18, 19)] == [16, 16, 16]
    assert [int(fac.reverse_offsets[L]) for L in (17, 18, 19)] == [0, 1, 2]
    fac, valid, nrows = stratified.stratify_mapping(pa.ThreePrimeMapFactory(2), 26, 26)   # single length
    assert nrows == 2 and valid.tolist() == [True] and int(fac.forward_offsets[26]) == 23
    fac, valid, _ = stratified.stratify_mapping(pa.VariableFivePrimeMapFactory({28: 12, 30: 29}), 27, 30)
    assert valid.tolist() == [False, True, False, True]
    with pytest.raises(TypeError):
        stratified.stratify_mapping(pa.CenterMapFactory(0), 25, 30)


def test_psite_profiles_match_reference_do_count():
    """The normalisation / median part of do_count, fed with the reference's own raw matrices."""
    z, meta = _golden()
    zero_point = meta["table"]["zero_point"][0]
    for ci, case in enumerate(meta["cases"]):
        raw = _raw_from_fixture(z, ci, case)
        norm, prof = stratified.psite_profiles(raw, case["norm_start"], case["norm_end"], case["min_counts"],
                                               zero_point, aggregate=case["aggregate"])
        for k in raw:
            assert np.array_equal(np.ma.getdata(norm[k]), z["c%d_norm_%d" % (ci, k)], equal_nan=True), (ci, k)
        for col in case["columns"]:
            got = np.ma.filled(np.ma.masked_invalid(np.ma.asarray(prof[col], float)), np.nan)
            assert np.array_equal(got, z["c%d_prof_%s" % (ci, col)], equal_nan=True), (ci, col)


def _ga_and_rois(z, meta, case):
    packed = pa.PackedAlignments(z["tid"], z["pos"], z["alen"], z["flags"], z["nblk"], z["blk_start"], z["blk_len"],
                                 references=meta["references"], lengths=meta["lengths"])
    fac = pa.FivePrimeMapFactory(case["offset"]) if case["kind"] == "fiveprime" else pa.ThreePrimeMapFactory(case["offset"])
    ga = pa.BAMGenomeArray(packed, mapping=fac)
    rois = []
    for region, masked in zip(meta["table"]["region"], meta["table"]["masked"]):
        roi = pa.SegmentChain.from_str(region)
        roi.add_masks(*pa.SegmentChain.from_str(masked))
        rois.append(roi)
    return ga, rois


@pytest.mark.gpu
def test_psite_raw_counts_match_reference_do_count():
    """[length, ROI, window] matrices from ONE launch == the reference's per-ROI, per-length loop."""
    z, meta = _golden()
    window = meta["table"]["window_size"][0]
    for ci, case in enumerate(meta["cases"]):
        ga, rois = _ga_and_rois(z, meta, case)
        raw = stratified.psite_raw_counts(ga, rois, meta["table"]["alignment_offset"], window, case["min_len"],
                                          case["max_len"])
        for k in range(case["min_len"], case["max_len"] + 1):
            assert np.array_equal(np.ma.getdata(raw[k]), z["c%d_raw_%d" % (ci, k)], equal_nan=True), (ci, k)
            assert np.array_equal(np.ma.getmaskarray(raw[k]), z["c%d_rawmask_%d" % (ci, k)]), (ci, k)
        # the array's own mapping rule is back in place afterwards
        seg = pa.GenomicSegment(meta["references"][0], 100, 400, "+")
        assert ga[seg].shape == (300,)


@pytest.mark.gpu
def test_phase_by_size_matches_oracle():
    from oracle import oracle
    from plastid_amd import synth
    from plastid_amd.packing import concat_file_major
    genome, tx, reads, _ = synth.make_config("C2", scale=0.002, tx_scale=0.004)
    chains = tx.chains()
    for fac, spec_args in ((pa.FivePrimeMapFactory(12), ("fiveprime", 12)), (pa.ThreePrimeMapFactory(27), ("threeprime", 27))):
        ga = pa.BAMGenomeArray(reads, mapping=fac)
        lengths = list(range(26, 32))
        got = stratified.phase_by_size(ga, chains, lengths, codon_buffer=5, batch_positions=30000)
        p = tx.plan_arrays(rows=1)
        want = {k: np.zeros(3) for k in lengths}
        for k in lengths:
            spec = oracle.mapping_spec(*spec_args, size_filter=(k, k))
            arrays, _ = oracle.count_segments(concat_file_major([reads]), spec, p["tid"], p["start"], p["end"], p["strand"])
            for c in range(tx.n):
                vec = np.concatenate([arrays[s] for s in range(tx.ex_off[c], tx.ex_off[c + 1])])
                if tx.strand[c] == 2:
                    vec = vec[::-1]
                n3 = len(vec) // 3
                want[k] += vec[:3 * n3].reshape(n3, 3)[5:-5].sum(0)
        for k in lengths:
            assert np.array_equal(got[k], want[k]), (spec_args, k)
        tab = stratified.phase_table(got)
        assert tab["reads_counted"].sum() == sum(v.sum() for v in want.values())
        tot = tab["phase0"] + tab["phase1"] + tab["phase2"]          # NaN for lengths the rule cannot map (0/0, as in the script)
        assert np.allclose(tot[tab["reads_counted"] > 0], 1.0) and np.isnan(tot[tab["reads_counted"] == 0]).all()
