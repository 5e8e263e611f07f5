"""The CIGAR -> positions step, the BAM / BGZF / BAI reader and the fetch emulations against
fixtures produced by the REFERENCE-HELD htslib itself (``tests/golden/hts_fixture.npz``, made in the
build container by ``tests/golden/make_hts_golden.py`` from htslib 1.3 as vendored under
``/root/reference/kent/src/htslib``: ``sam_write1`` / ``sam_index_build`` wrote the bytes,
``bam_endpos``, a ``bam_cigar_type``-driven CIGAR walk and ``sam_itr_queryi`` produced the
expectations).  pysam delegates exactly these steps to htslib (genome_array.py:800-809,
map_factories.pyx:243), so this pins what SURVEY 8(c) lists as unpinned by runnable reference tests."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plastid_amd import packing  # noqa: E402
from plastid_amd.bam import read_bam  # noqa: E402

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hts_fixture.npz")


@pytest.fixture(scope="module")
def hts():
    return np.load(FIX)


@pytest.fixture(scope="module")
def bam_path(hts, tmp_path_factory):
    d = tmp_path_factory.mktemp("hts")
    path = str(d / "htslib.bam")
    open(path, "wb").write(hts["bam"].tobytes())
    open(path + ".bai", "wb").write(hts["bai"].tobytes())
    return path


def _placed(hts):
    return np.nonzero(hts["tid"] >= 0)[0]


def _htslib_runs(hts, i):
    """Maximal runs of the aligned reference positions htslib reports for record i."""
    p = hts["positions"][hts["positions_off"][i]:hts["positions_off"][i + 1]]
    return packing.positions_to_runs(p), len(p)


def _expected_nh(hts):
    """The `nh` column both decoders must produce for the fixture: htslib's bam_aux2i value of the NH tag (version 1.3 hands
    a uint32 back through an int32: 4 000 000 000 reads -294 967 296), clamped to 0 .. 65 535; 0 without an integer NH."""
    v = hts["nh"].astype(np.int64).copy()
    v[v < -(1 << 20)] += 1 << 32
    return np.where(hts["has_nh"] == 1, np.clip(v, 0, 65535), 0).astype(np.uint16)


def test_fixture_covers_every_cigar_operation(hts):
    assert set(np.unique(hts["cigar_op"])) == set(range(9))                 # M I D N S H P = X
    n_ops = np.diff(hts["cigar_off"])
    assert (n_ops == 0).sum() >= 5 and n_ops.max() >= 8                     # CIGAR-less placed reads, long CIGARs
    assert (hts["flag"] & 0x4).any() and (hts["flag"] & 0x100).any() and (hts["flag"] & 0x400).any()
    assert (hts["tid"] < 0).sum() == int(hts["n_no_coor"]) == 2


def test_cigar_to_runs_python_and_oracle_match_htslib(hts):
    """packing.cigar_to_runs (product) and po_cigar_to_runs (oracle) give htslib's positions, and the
    reference end they imply is bam_endpos (sam.c:329-341)."""
    from oracle import oracle
    for i in _placed(hts):
        ops = hts["cigar_op"][hts["cigar_off"][i]:hts["cigar_off"][i + 1]]
        lens = hts["cigar_len"][hts["cigar_off"][i]:hts["cigar_off"][i + 1]]
        cig = [(int(o), int(n)) for o, n in zip(ops, lens)]
        want_runs, want_len = _htslib_runs(hts, i)
        runs, L = packing.cigar_to_runs(int(hts["pos"][i]), cig)
        assert runs == want_runs and L == want_len, i
        oruns, oL = oracle.cigar_to_runs(int(hts["pos"][i]), cig) if cig else ([], 0)
        assert oruns == want_runs and oL == want_len, i
        # end coordinate: htslib's bam_endpos counts every reference-consuming op, pos + 1 without any
        ref_len = sum(n for o, n in cig if o in (0, 2, 3, 7, 8))
        assert int(hts["endpos"][i]) == int(hts["pos"][i]) + (ref_len if ref_len > 0 else 1)


def test_native_reader_decodes_the_htslib_written_bam(hts, bam_path):
    """bam_stager.cpp on bytes written by htslib: records, aligned runs, strand bit, mapped count."""
    for threads in (1, 4):
        got = read_bam(bam_path, threads=threads)
        keep = _placed(hts)                                                  # unplaced reads are never fetched
        assert got.n == len(keep)
        assert list(got.references) == [str(x) for x in hts["references"]]
        assert list(got.lengths) == [int(x) for x in hts["lengths"]]
        assert np.array_equal(got.tid, hts["tid"][keep]) and np.array_equal(got.pos, hts["pos"][keep])
        assert np.array_equal(got.flags & 1, (hts["flag"][keep] >> 4) & 1)   # FLAG 0x10 -> is_reverse
        assert got.mapped == int(hts["index_stat"][:, 1].sum())             # pysam AlignmentFile.mapped
        assert got.mapped == int(((hts["flag"][keep] & 0x4) == 0).sum())
        # what filter functions may look at (genome_array.py:697-722): the FLAG word, MAPQ and l_seq as htslib reads them back
        assert np.array_equal(got.flag16, hts["flag"][keep]) and np.array_equal(got.mapq, hts["mapq"][keep])
        assert np.array_equal(got.qlen, hts["l_qseq"][keep])
        assert len(np.unique(hts["mapq"])) > 50 and (hts["mapq"] == 255).any()
        # the NH:i tag as htslib's bam_aux_get / bam_aux2i read it back (pysam's has_tag / get_tag), clamped to 0 .. 65 535; 0 for
        # a record without one -- and for an NH of a non-integer type, which bam_aux2i answers with 0 too
        want_nh = _expected_nh(hts)[keep]
        assert np.array_equal(got.nh, want_nh)
        assert (want_nh > 0).sum() > 800 and (want_nh == 65535).sum() > 10 and ((hts["has_nh"][keep] == 1) & (want_nh == 0)).sum() > 50
        for j in np.nonzero(want_nh > 0)[0][:5].tolist() + np.nonzero(want_nh == 0)[0][:3].tolist():
            r = got.read(j)
            assert r.has_tag("NH") == bool(want_nh[j]) and (r.get_tag("NH") == int(want_nh[j]) if want_nh[j] else True)
            if not want_nh[j]:
                with pytest.raises(KeyError):
                    r.get_tag("NH")
        for j in (0, 7, len(keep) - 1):
            r, f = got.read(j), int(hts["flag"][keep[j]])
            assert r.flag == f and r.mapping_quality == int(hts["mapq"][keep[j]]) and r.query_length == int(hts["l_qseq"][keep[j]])
            assert (r.is_secondary, r.is_duplicate, r.is_qcfail, r.is_unmapped, r.is_paired, r.is_read2, r.mate_is_reverse, r.is_reverse) == \
                tuple(bool(f & b) for b in (0x100, 0x400, 0x200, 0x4, 0x1, 0x80, 0x20, 0x10))
        for j, i in enumerate(keep):
            runs, L = _htslib_runs(hts, i)
            assert int(got.alen[j]) == L, i
            assert got.runs_of(j) == (runs if L else []), i
        # reference end as the fetch emulation uses it == bam_endpos
        assert np.array_equal(got.ref_end(), hts["endpos"][keep])


@pytest.mark.parametrize("knobs", [{"PB_CHUNK": "1"}, {"PB_CHUNK": "700", "PB_HEAD": "24"}, {"PB_CHUNK": "1", "PB_HEAD": "5", "PB_ZLIB": "1"}])
def test_chunked_decoding_of_the_htslib_written_bam(hts, bam_path, monkeypatch, knobs):
    """The whole-file reader decodes chunks of BGZF members from guessed record boundaries and chains them;
    on htslib's own records (names, sequences, qualities, tags -- not the bare records of tests/bam_writer)
    tiny chunks, heads shorter than a record and the zlib path give the arrays of the default decode."""
    whole = read_bam(bam_path, threads=1)
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    for threads in (1, 3):
        got = read_bam(bam_path, threads=threads)
        for name in ("tid", "pos", "alen", "flags", "nblk", "blk_start", "blk_len", "flag16", "mapq", "qlen", "nh"):
            assert np.array_equal(getattr(got, name), getattr(whole, name)), (name, knobs)
        assert got.mapped == whole.mapped


def test_region_reads_through_the_bai_match_hts_itr_query(hts, bam_path):
    """read_bam(regions=...) (BAI bins + linear index + overlap test) returns exactly the records
    htslib's iterator yields (hts.c:1924-1960), for every seeded region; so do the pure-Python fetch of
    PackedAlignments and the oracle's fetch emulation."""
    from oracle import oracle
    from plastid_amd.packing import concat_file_major
    whole = read_bam(bam_path)
    keep = _placed(hts)
    rec_of = {int(i): j for j, i in enumerate(keep)}
    refs = [str(x) for x in hts["references"]]
    aln = concat_file_major([whole])
    spec = oracle.mapping_spec("center", 0)      # reads_out of the center rule = every fetched read with L > 0
    for q, (tid, beg, end) in enumerate(hts["regions"]):
        want = hts["region_records"][hts["region_off"][q]:hts["region_off"][q + 1]]
        want_j = np.array([rec_of[int(i)] for i in want], np.int64)
        # the Python duck type of AlignmentFile.fetch
        got_idx = whole.fetch_indices(refs[tid], int(beg), int(end))
        assert np.array_equal(got_idx, want_j), (q, tid, beg, end)
        # the native reader through the index
        part = read_bam(bam_path, regions=[(refs[tid], int(beg), int(end))])
        assert part.n == len(want_j), (q, tid, beg, end)
        assert np.array_equal(part.pos, whole.pos[want_j]) and np.array_equal(part.alen, whole.alen[want_j])
        assert np.array_equal(part.flags, whole.flags[want_j])
        # the oracle's fetch emulation (what po_segment hands to the map function)
        if q % 8 == 0:
            _, _, mapped = oracle.count_segments(aln, spec, [tid], [int(beg)], [int(end)], [3], want_mapped=True)
            assert np.array_equal(np.nonzero(mapped[0])[0], want_j[whole.alen[want_j] > 0]), (q, tid, beg, end)


@pytest.mark.gpu
def test_htslib_bam_end_to_end_on_the_gpu(hts, bam_path):
    """The htslib-written BAM staged by the native reader and counted by the HIP kernels equals the
    oracle run on htslib's own positions -- all five rules, whole contigs and short segments."""
    from oracle import oracle
    import plastid_amd as pa
    from plastid_amd import synth
    keep = _placed(hts)
    runs = [_htslib_runs(hts, i)[0] for i in keep]
    ref = pa.PackedAlignments.from_runs([int(t) for t in hts["tid"][keep]], [bool(f & 16) for f in hts["flag"][keep]], runs,
                                        references=[str(x) for x in hts["references"]], lengths=[int(x) for x in hts["lengths"]],
                                        positions=[int(p) for p in hts["pos"][keep]])
    aln = pa.packing.concat_file_major([ref])
    lens = [int(x) for x in hts["lengths"]]
    seg_tid = np.array([0, 0, 1, 1, 2, 0, 0], np.int32)
    seg_start = np.array([0, 0, 0, 1000, 0, 20000, 150000], np.int64)
    seg_end = np.array([lens[0], lens[0], lens[1], 30000, lens[2], 20300, 190000], np.int64)
    seg_strand = np.array([1, 2, 3, 2, 1, 3, 1], np.uint8)
    ga_bam = read_bam(bam_path)
    eng = pa.engine.Engine(0)
    eng.set_alignments([ga_bam])
    for mapping in (("fiveprime", 3), ("threeprime", 0), ("center", 1), ("variable", synth.VARIABLE_OFFSETS),
                    ("stratified", synth.VARIABLE_OFFSETS, 20, 40)):
        synth.mapping_factory(mapping)._configure(eng)
        rows = eng.rows
        ln = seg_end - seg_start
        off = np.zeros(len(ln) + 1, np.int64)
        np.cumsum(ln * rows, out=off[1:])
        plan = eng.plan(seg_tid, seg_start, seg_end, seg_strand, off[:-1], np.ones(len(ln), np.int8), ln, int(off[-1]), rows)
        got = plan.count(np.float64)
        kind = mapping[0]
        spec = oracle.mapping_spec(kind, mapping[1] if kind in ("fiveprime", "threeprime", "center") else 0,
                                   mapping[1] if kind in ("variable", "stratified") else None,
                                   *(mapping[2:4] if kind == "stratified" else (25, 35)))
        arrays, _ = oracle.count_segments(aln, spec, seg_tid, seg_start, seg_end, seg_strand)
        exp = np.concatenate([a.astype(np.float64).reshape(-1) for a in arrays])
        assert np.array_equal(got, exp), mapping
        plan.close()
    eng.close()


def test_odd_and_damaged_auxiliary_fields_on_the_host(tmp_path):
    """The NH walk over auxiliary fields it cannot size (an unterminated string, an array longer than the record, an unknown
    type, a tag cut off) ends without a value instead of reading past the record; every integer type, negative and
    oversized values, fields of every other kind before the tag (tests/bam_writer.odd_aux_records)."""
    from tests import bam_writer
    from plastid_amd.bam import read_bam
    recs, want = bam_writer.odd_aux_records()
    path = str(tmp_path / "aux.bam")
    bam_writer.write_bam(path, ["c"], [100000], recs)
    got = read_bam(path)
    assert got.n == len(recs) and got.nh.tolist() == want
    assert [got.read(i).has_tag("NH") for i in range(got.n)] == [bool(v) for v in want]
