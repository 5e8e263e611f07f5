"""CPU-only tests: host logic (intervals, factories' parameter handling, packing,
synthetic generators) and the C ABI surface.  No compute calls (no GPU here)."""
import ctypes
import io
import os
import re
import sys
import warnings

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plastid_amd as pa  # noqa: E402
from plastid_amd import _lib, map_factories as mf, packing, synth  # noqa: E402
from plastid_amd.packing import PackedAlignments  # noqa: E402
from plastid_amd.build import build_library  # noqa: E402
from tests import golden_util as gu  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------- C ABI
def test_library_builds_loads_and_exports_every_declared_symbol():
    build_library()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "plastid_counts.h")).read()
    declared = set(re.findall(r"\b(pc_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "library does not export %s" % name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().pc_abi_version() == _lib.ABI_VERSION
    assert re.search(r"#define PC_ABI_VERSION (\d+)", header).group(1) == str(_lib.ABI_VERSION)


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("GPU present")
    from plastid_amd.engine import Engine
    with pytest.raises(pa.EngineError) as e:
        Engine(0)
    assert "no CPU fallback" in str(e.value)
    with pytest.raises(pa.EngineError):
        pa.FivePrimeMapFactory(0)([], pa.GenomicSegment("c", 0, 10, "+"))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "plastid_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("the oracle", "").lower() or f == "__none__", \
                    "%s mentions the oracle" % f


# -------------------------------------------------------------- factories
def test_ctor_errors_match_reference():
    g = gu.load("offset_tables")
    errs = [c for c in g.cases if c["kind"] == "ctor_errors"][0]["errors"]
    ctors = {
        "FivePrime(-1)": lambda: pa.FivePrimeMapFactory(-1),
        "ThreePrime(-1)": lambda: pa.ThreePrimeMapFactory(-1),
        "Center(-1)": lambda: pa.CenterMapFactory(-1),
        "Strat(min==max)": lambda: pa.StratifiedVariableFivePrimeMapFactory({}, 25, 25),
        "Strat(max<min)": lambda: pa.StratifiedVariableFivePrimeMapFactory({}, 30, 25),
        "SizeFilter(max<min)": lambda: pa.SizeFilterFactory(30, 25),
        "SizeFilter(min<1)": lambda: pa.SizeFilterFactory(0, 25),
        "Variable(bad,no default)": lambda: pa.VariableFivePrimeMapFactory({25: 30}),
        "Segment(end<start)": lambda: pa.GenomicSegment("c", 10, 5, "+"),
        "Chain(mixed strands)": lambda: pa.SegmentChain(pa.GenomicSegment("c", 0, 5, "+"),
                                                        pa.GenomicSegment("c", 10, 15, "-")),
    }
    assert set(ctors) == set(errs)
    for name, ctor in ctors.items():
        if errs[name] is None:
            ctor()
        else:
            with pytest.raises(Exception) as e:
                ctor()
            assert type(e.value).__name__ == errs[name], (name, type(e.value).__name__, errs[name])


def test_offset_tables_match_reference_behaviour():
    g = gu.load("offset_tables")
    n = 0
    for case in g.cases:
        if case["kind"] != "table":
            continue
        od = {(k if k == "default" else int(k)): v for k, v in case["offset_dict"].items()}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fn = pa.VariableFivePrimeMapFactory(od)
        efw, erc = g[case["fw"]], g[case["rc"]]
        m = len(efw)
        assert np.array_equal(fn.forward_offsets[1:m], efw[1:]), od
        assert np.array_equal(fn.reverse_offsets[1:m], erc[1:]), od
        assert (fn.forward_offsets == -1).tolist() == (fn.reverse_offsets == -1).tolist()
        n += 1
    assert n >= 8


def test_offset_file_grammar():
    g = gu.load("offset_tables")
    case = [c for c in g.cases if c["kind"] == "offset_file"][0]
    d = mf._parse_variable_offset_file(io.StringIO(case["text"]))
    assert {str(k): v for k, v in d.items()} == case["expected"]
    fn = pa.VariableFivePrimeMapFactory.from_file(io.StringIO("# comment\n" + case["text"]))
    assert fn.forward_offsets[28] == 13 and fn.forward_offsets[40] == 13 and fn.reverse_offsets[30] == 15
    for bad in ("28\t12\t1\n", "x\t3\n", "28\tq\n", "28\t1\n28\t2\n"):
        with pytest.raises(pa.MalformedFileError):
            mf._parse_variable_offset_file(io.StringIO(bad))


def test_factory_properties():
    f = pa.FivePrimeMapFactory(3)
    assert f.offset == 3
    f.offset = 5
    assert f.offset == 5
    c = pa.CenterMapFactory()
    assert c.nibble == 0
    c.nibble = 4
    assert c.nibble == 4
    with pytest.raises(OverflowError):
        c.nibble = -2
    s = pa.StratifiedVariableFivePrimeMapFactory(None, 25, 35)
    assert s.shape == [11] and list(s.row_keys) == list(range(25, 36))
    sf = pa.SizeFilterFactory(25, 30)

    class R(object):
        positions = list(range(27))
    assert sf(R()) is True
    R.positions = list(range(31))
    assert sf(R()) is False
    assert pa.SizeFilterFactory(25)(R()) is True
    with pytest.raises(TypeError):
        pa.FivePrimeMapFactory(1.5)
    with pytest.raises(TypeError):
        pa.FivePrimeMapFactory(0)(None, pa.GenomicSegment("c", 0, 1, "+"))


# ---------------------------------------------------------------- intervals
def test_genomic_segment():
    a = pa.GenomicSegment("chrI", 10, 20, "+")
    assert len(a) == 10 and str(a) == "chrI:10-20(+)" and a.c_strand == 1
    assert pa.GenomicSegment.from_str(str(a)) == a and hash(pa.GenomicSegment.from_str(str(a))) == hash(a)
    b = pa.GenomicSegment("chrI", 15, 30, "+")
    assert a.overlaps(b) and not a.contains(b) and a < b
    assert pa.GenomicSegment("chrI", 12, 18, "+") in a
    assert not a.overlaps(pa.GenomicSegment("chrI", 15, 30, "-"))
    with pytest.raises(ValueError):
        pa.GenomicSegment("chrI", 10, 20, "x")
    assert sorted([b, a]) == [a, b]


def test_segment_chain_structure_and_masks():
    S = pa.GenomicSegment
    chain = pa.SegmentChain(S("c", 300, 340, "-"), S("c", 0, 90, "-"), S("c", 90, 120, "-"), S("c", 100, 130, "-"))
    assert [(s.start, s.end) for s in chain] == [(0, 130), (300, 340)]   # adjacent + overlapping merged (Q14)
    assert chain.length == 170 and len(chain) == 2 and chain.strand == "-" and chain.chrom == "c"
    assert str(chain) == "c:0-130^300-340(-)" and pa.SegmentChain.from_str(str(chain)) == chain
    assert chain.get_position_list() == list(range(0, 130)) + list(range(300, 340))
    assert chain.get_position_set() == set(chain.get_position_list())
    chain.add_masks(S("c", 120, 310, "-"), S("c", 5, 6, "-"))
    assert [(s.start, s.end) for s in chain.mask_segments] == [(5, 6), (120, 130), (300, 310)]
    assert chain.masked_length == 170 - 21
    chain.add_masks(S("c", 4, 7, "-"))
    assert [(s.start, s.end) for s in chain.mask_segments] == [(4, 7), (120, 130), (300, 310)]
    assert 5 not in chain.get_masked_position_set() and 8 in chain.get_masked_position_set()
    assert chain.get_segmentchain_coordinate(0, stranded=False) == 0
    assert chain.get_segmentchain_coordinate(0) == 169 and chain.get_segmentchain_coordinate(339) == 0
    with pytest.raises(KeyError):
        chain.get_segmentchain_coordinate(200)
    chain.reset_masks()
    assert chain.masked_length == 170 and chain.mask_segments == []
    with pytest.raises(ValueError):
        chain.add_masks(S("c", 4, 7, "+"))
    assert len(pa.SegmentChain()) == 0 and str(pa.SegmentChain()) == "na"


def test_get_counts_duck_typed_array():
    """SegmentChain.get_counts works on ANY object with .get(seg, roi_order=False)
    (the reference's tests pass a GenomeArray there, test_roitools.py:1275-1291)."""
    class FakeGA(object):
        def get(self, seg, roi_order=True):
            return np.arange(seg.start, seg.end, dtype=np.int64)
    S = pa.GenomicSegment
    plus = pa.SegmentChain(S("c", 10, 13, "+"), S("c", 20, 22, "+"))
    minus = pa.SegmentChain(S("c", 10, 13, "-"), S("c", 20, 22, "-"))
    ga = FakeGA()
    assert plus.get_counts(ga).tolist() == [10, 11, 12, 20, 21] and plus.get_counts(ga).dtype == np.float64
    assert minus.get_counts(ga).tolist() == [21, 20, 12, 11, 10]
    assert minus.get_counts(ga, stranded=False).tolist() == [10, 11, 12, 20, 21]
    minus.add_masks(S("c", 10, 11, "-"))
    m = minus.get_masked_counts(ga)
    assert np.ma.getmaskarray(m).tolist() == [False, False, False, False, True]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert pa.SegmentChain().get_counts(ga).shape == (0,)
    assert len(w) == 1 and issubclass(w[0].category, pa.DataWarning)


# ------------------------------------------------------------------ packing
def test_cigar_to_runs_hand_checked():
    g = gu.load("quirks")
    case = [c for c in g.cases if c["kind"] == "hand_cigars"][0]
    for i in g["hand_indices"]:
        runs, L = packing.cigar_to_runs(int(g["hand_pos"][i]), packing.parse_cigar_string(case["cigars"][i]))
        pos = [p for s, n in runs for p in range(s, s + n)]
        assert pos == list(g["hand_positions_%d" % i]) and L == len(pos)
        assert packing.positions_to_runs(pos) == runs
    with pytest.raises(ValueError):
        packing.parse_cigar_string("10M5")


def test_packed_alignments_roundtrip_and_fetch():
    p = pa.PackedAlignments.from_cigars([0, 0, 1, 0], [5, 50, 7, 20], ["10M", "5M100N5M", "3S8M", "4M1D4M"],
                                        [False, True, False, True], references=["a", "b"], lengths=[500, 100],
                                        sort=True)
    assert p.tid.tolist() == [0, 0, 0, 1] and p.pos.tolist() == [5, 20, 50, 7]
    assert p.alen.tolist() == [10, 8, 10, 8] and p.nblk.tolist() == [1, 2, 2, 1]
    assert p.blk_start.tolist() == [20, 25, 50, 155] and p.blk_len.tolist() == [4, 4, 5, 5]
    assert p.ref_end().tolist() == [15, 29, 160, 15]
    assert p.read(2).positions == list(range(50, 55)) + list(range(155, 160))
    assert [r.index for r in p.fetch("a", 100, 120)] == [2]        # spans the window through its intron
    assert [r.index for r in p.fetch("a", 15, 20)] == []
    assert [r.index for r in p.fetch("a", 14, 21)] == [0, 1]
    assert [r.index for r in p.fetch("b", 0, 100)] == [3]
    with pytest.raises(ValueError):
        p.fetch_indices("zzz", 0, 1)
    sub = p.subset([1, 3])
    assert sub.blk_start.tolist() == [20, 25] and sub.n == 2
    from_reads = pa.PackedAlignments.from_reads([p.read(i) for i in range(p.n)], references=["a", "b"],
                                                lengths=[500, 100])
    for k in ("tid", "pos", "alen", "flags", "nblk", "blk_start", "blk_len"):
        assert np.array_equal(getattr(from_reads, k), getattr(p, k)), k


def test_wide_reads_survive_a_reference_remap():
    """A source whose reference list differs from the array's chromosome list (extra contigs, another order) is
    re-labelled by ``_pack_source``; the side arrays of over-wide reads (> 65 535 aligned bases, > 255 runs) and
    a header order that is not the array's sorted one must both survive that."""
    from plastid_amd.genome_array import _pack_source
    # references in an order that is not lexicographic: chr2 before chr10
    refs = ["chr2", "chr10"]
    tid = np.array([0, 0, 1], np.int32)
    pos = np.array([5, 100, 7], np.int32)
    alen = np.array([30, 65535, 28], np.uint16)
    nblk = np.array([1, 255, 1], np.uint8)
    src = PackedAlignments(tid, pos, alen, np.zeros(3, np.uint8), nblk, references=refs, lengths=[200000, 1000],
                           wide_idx=[1], wide_alen=[70000], wide_nblk=[1])
    chroms = sorted(refs + ["chr3"])             # chr10, chr2, chr3
    index = {c: i for i, c in enumerate(chroms)}
    out = _pack_source(src, chroms, index)
    assert list(out.tid) == [0, 1, 1] and list(out.pos) == [7, 5, 100]
    assert list(out.wide_idx) == [2] and list(out.wide_alen) == [70000] and list(out.wide_nblk) == [1]
    assert out.true_alen()[2] == 70000
    # same order, one extra contig in the array: no re-ordering, wide arrays kept
    out2 = _pack_source(PackedAlignments(tid, pos, alen, np.zeros(3, np.uint8), nblk, references=["a", "c"],
                                         wide_idx=[1], wide_alen=[70000], wide_nblk=[1]), ["a", "b", "c"],
                        {"a": 0, "b": 1, "c": 2})
    assert list(out2.tid) == [0, 0, 2] and list(out2.wide_idx) == [1] and out2.true_alen()[1] == 70000


def test_synthetic_configs_are_seeded_and_valid():
    for name, scale, txs in (("C1", 0.01, 0.5), ("C2", 0.0002, 0.005), ("C4", 0.00004, 0.002), ("C5", 0.00002, 0.002)):
        g, tx, reads, mapping = synth.make_config(name, scale=scale, tx_scale=txs)
        g2, tx2, reads2, _ = synth.make_config(name, scale=scale, tx_scale=txs)
        reads.validate()
        assert np.array_equal(reads.pos, reads2.pos) and np.array_equal(tx.ex_start, tx2.ex_start)
        assert np.all(tx.ex_end > tx.ex_start)
        ends = reads.ref_end()
        assert np.all(ends <= np.asarray(g[1])[reads.tid])
        p = tx.plan_arrays(rows=1)
        assert p["out_elems"] == tx.n_positions
        # every output element is addressed exactly once
        hit = np.zeros(p["out_elems"], np.int32)
        for s in range(len(p["tid"])):
            n = p["end"][s] - p["start"][s]
            hit[p["out_off"][s] + p["out_step"][s].astype(np.int64) * np.arange(n)] += 1
        assert hit.min() == 1 and hit.max() == 1
    assert synth.mapping_factory(("stratified", synth.VARIABLE_OFFSETS, 25, 35)).shape == [11]


def test_header_is_plain_c_and_cxx(tmp_path):
    """include/plastid_counts.h is the drop-in boundary: it must compile as C99 and as C++ on
    its own (no torch / HIP types), and declare exactly the symbols the ctypes shim binds."""
    import re
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "plastid_counts.h")
    src = tmp_path / "use.c"
    src.write_text('#include "plastid_counts.h"\nint main(void) { return pc_abi_version() == PC_ABI_VERSION ? 0 : 1; }\n')
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        if shutil.which(cc) is None:
            pytest.skip("no %s" % cc)
        lang = ["-x", "c"] if cc == "gcc" else ["-x", "c++"]
        subprocess.check_call([cc, std, "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", os.path.dirname(hdr)] + lang + [str(src)])
    from plastid_amd import _lib
    declared = set(re.findall(r"\b(pc_[a-z_0-9]+)\s*\(", open(hdr).read()))
    assert declared == set(_lib.SIGNATURES)


def _build_c_client(tmp_path):
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from plastid_amd import build
    build.build_library()
    exe = str(tmp_path / "c_client")
    libdir = os.path.join(root, "plastid_amd")
    subprocess.check_call([shutil.which("gcc") or "gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "c_client.c"), "-L", libdir, "-lplastid_counts",
                           "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_c_client_links_against_the_abi(tmp_path):
    """examples/c_client.c -- a client that knows only the header -- compiles and links (C99)."""
    assert os.path.exists(_build_c_client(tmp_path))


@pytest.mark.gpu
def test_c_client_runs(tmp_path):
    import subprocess
    out = subprocess.check_output([_build_c_client(tmp_path)]).decode()
    want = [0] * 30
    for p in (102, 102, 105, 107):
        want[p - 95] += 1
    assert out.splitlines()[0] == "counts[95..125) = " + " ".join(str(v) for v in want)
    assert out.splitlines()[1].startswith("total = 4 ")


def test_no_kernel_uses_scratch_memory(tmp_path):
    """Every kernel of the built gfx950 code object has ``private_segment_fixed_size`` 0.  A run-time
    index into a small per-thread array (round 1: ``HistCfg::base[mode]``) silently moves the whole
    struct to scratch: 44 bytes per lane that every work item wrote to HBM -- 2 GB per step on C4 and
    8 GB on C5 (profiles/r02: WRITE_SIZE 2-3.7x the output bytes)."""
    import shutil
    import subprocess
    from plastid_amd import build
    lib = build.build_library()
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf not available")
    work = tmp_path / "co"
    work.mkdir()
    copy = str(work / "lib.so")
    shutil.copy(lib, copy)
    subprocess.check_call([objdump, "--offloading", copy], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=str(work))
    objs = [f for f in os.listdir(str(work)) if "gfx950" in f]
    assert objs, "no gfx950 code object found in %s" % lib
    notes = subprocess.check_output([readelf, "--notes", str(work / objs[0])]).decode()
    sizes = [int(line.split(":")[1]) for line in notes.splitlines() if ".private_segment_fixed_size" in line]
    names = [line.split(":")[1].strip() for line in notes.splitlines() if line.strip().startswith(".name:")]
    assert len(sizes) == len(names)
    # (the engine's own kernels, namespace pc; rocPRIM's radix sort -- staging time only -- is library code)
    own = [(n, s) for n, s in zip(names, sizes) if n.startswith("_ZN2pc")]
    assert len(own) >= 30
    bad = [n for n, s in own if s != 0]
    assert not bad, "kernels using scratch memory: %s" % bad


def test_center_kernel_replays_through_dpp_rows(tmp_path):
    """k_center2's replay step is three vector instructions -- v_and_b32_dpp (coverage mask of the entry & the lane's bit),
    v_lshlrev_b32 (-> the high word of 2.0 or 0.0), v_fmac_f64_dpp -- with the entry broadcast inside a 16-lane row by DPP
    (row_newbcast), and nothing scalar: checked on the built code object.  The kernel must fit eight waves per SIMD by
    its vector registers (<= 64); LDS only holds the by-length value table (plain reads), no scratch."""
    import re
    import shutil
    import subprocess
    from plastid_amd import build
    lib = build.build_library()
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf not available")
    work = tmp_path / "co"
    work.mkdir()
    copy = str(work / "lib.so")
    shutil.copy(lib, copy)
    subprocess.check_call([objdump, "--offloading", copy], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=str(work))
    obj = str(work / [f for f in os.listdir(str(work)) if "gfx950" in f][0])
    notes = subprocess.check_output([readelf, "--notes", obj]).decode()
    # (the instantiation short-read files run: no step counting, no per-batch test for indirect entries)
    block = [b for b in notes.split("- .agpr_count")[1:] if re.search(r"\.name:\s+_ZN2pc9k_center2ILb0ELb0ELb0EE", b)]
    assert len(block) == 1
    assert int(re.search(r"\.vgpr_count:\s+(\d+)", block[0]).group(1)) <= 64
    assert int(re.search(r"\.sgpr_count:\s+(\d+)", block[0]).group(1)) <= 96, "eight waves per SIMD need <= 96 SGPRs (amdgpu_waves_per_eu(8, 8))"
    assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block[0]).group(1)) == 0
    symbol = re.search(r"\.name:\s+(_ZN2pc9k_center2ILb0ELb0ELb0EE\S+)", block[0]).group(1)
    dis = subprocess.check_output([objdump, "-d", "--disassemble-symbols=" + symbol, obj]).decode()
    lines = [ln.split("//")[0].strip() for ln in dis.splitlines() if "\t" in ln]
    assert len(lines) > 500, "k_center2 not found in the disassembly"
    ops = [ln.split(None, 1)[0] for ln in lines if ln]
    n_fma = sum(1 for ln in lines if ln.startswith("v_fmac_f64_dpp") and "row_newbcast" in ln)
    assert n_fma >= 16 and n_fma % 4 == 0
    assert sum(1 for ln in lines if ln.startswith("v_and_b32_dpp") and "row_newbcast" in ln) == n_fma
    # (LDS holds the by-length value table: plain reads, no atomics; the lane permutes that hand a descriptor's row ranges
    # and the long-span candidates round happen once per chunk, outside the step blocks checked below)
    assert not any(op.startswith(("ds_add", "ds_sub", "ds_max", "ds_min")) for op in ops), "k_center2 uses no LDS atomics"
    # a 16-step block is 48 instructions of exactly these three kinds: nothing scalar, no lane crossing but the DPP
    idx = [i for i, ln in enumerate(lines) if ln.startswith("v_fmac_f64_dpp") and "row_newbcast:15" in ln]
    assert idx
    for last in idx:
        blockops = ops[last - 47:last + 1]
        assert all(o.startswith(("v_and_b32_dpp", "v_lshlrev_b32", "v_fmac_f64_dpp")) for o in blockops), blockops


def test_usable_cpus_respects_the_container_limits():
    """bench.py sizes its thread pools (and reports `cpu_baseline.usable_cores`) by what the process may use:
    hardware threads, affinity mask and the cgroup CPU quota -- the GPU boxes show 256 threads and grant 16."""
    import bench
    n = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert n <= len(os.sched_getaffinity(0))


def test_host_helpers_of_the_engine(tmp_path):
    """plastid_amd/csrc/host_util.h is plain C++: the worker pool behind the host passes (regions of every width, nested
    regions, concurrent callers), the galloping lower bound of the plan build against std::lower_bound for every hint,
    the vector without zero-fill, and the host's look at the contig column of caller-owned records (scan_contigs: the
    record bounds of every contig and the first record out of order or out of range, against a record-by-record walk,
    for every thread count) and the window size of a plan (choose_window: the sizes the published numbers were measured
    with, the knob's limits) -- compiled with the host compiler and run here."""
    import shutil
    import subprocess
    cxx = shutil.which("g++") or shutil.which("c++")
    if not cxx:
        pytest.skip("no host C++ compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_util_test")
    subprocess.check_call([cxx, "-O2", "-std=c++17", "-pthread", "-I", os.path.join(root, "plastid_amd", "csrc"),
                           os.path.join(root, "tests", "host_util_test.cpp"), "-o", exe])
    out = subprocess.run([exe], stdout=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stdout.decode()
    assert b"host_util: ok" in out.stdout


def test_inflate_and_plan_kernels_are_in_the_code_object(tmp_path):
    """Round 4's device code beside the counting kernels: the BGZF inflate kernel in both forms (batch decoder of the
    block symbols, wave-uniform decoder), without scratch memory and with an LDS footprint that leaves eleven waves per
    CU; the walk over the symbol table is the hand-written loop (one LDS read, one LDS write, seven instructions per
    symbol); every kernel of the GPU plan builder is there."""
    import re
    import shutil
    import subprocess
    from plastid_amd import build
    lib = build.build_library()
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf not available")
    work = tmp_path / "co"
    work.mkdir()
    copy = str(work / "lib.so")
    shutil.copy(lib, copy)
    subprocess.check_call([objdump, "--offloading", copy], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=str(work))
    obj = str(work / [f for f in os.listdir(str(work)) if "gfx950" in f][0])
    notes = subprocess.check_output([readelf, "--notes", obj]).decode()
    blocks = {re.search(r"\.name:\s+(\S+)", b).group(1): b for b in notes.split("- .agpr_count")[1:] if re.search(r"\.name:\s+(\S+)", b)}
    for form in ("ILb1E", "ILb0E"):
        name = [k for k in blocks if k.startswith("_ZN5pcbam14k_bgzf_inflate" + form)]
        assert len(name) == 1, form
        b = blocks[name[0]]
        assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", b).group(1)) == 0
        assert int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", b).group(1)) <= 160 * 1024 // 11
    batch = [k for k in blocks if k.startswith("_ZN5pcbam14k_bgzf_inflateILb1E")][0]
    dis = subprocess.check_output([objdump, "-d", "--disassemble-symbols=" + batch, obj]).decode()
    lines = [ln.split("//")[0].strip() for ln in dis.splitlines() if "\t" in ln]
    # (round 6: seven instructions -- the symbol word's low byte, four times its bit count, is added to the LDS address by an SDWA add)
    at = [i for i, ln in enumerate(lines) if ln.startswith("v_add_u32_sdwa") and "src1_sel:BYTE_0" in ln]
    assert at, "the walk over the symbol table was not found"
    loop = [ln.split(None, 1)[0] for ln in lines[at[0] - 4:at[0] + 3]]
    assert loop == ["ds_read_b32", "s_waitcnt", "ds_write_b32", "v_add_u32_e32", "v_add_u32_sdwa", "v_cmp_gt_u32_e32", "s_cbranch_vccnz"], loop
    for k in ("k_plan_segs", "k_group_ends", "k_island_flags", "k_island_fill", "k_island_lens", "k_island_offsets", "k_seg_island", "k_pieces_raw",
              "k_pieces_sorted", "k_tile_fill", "k_out_total", "k_out_raw", "k_out_sorted", "k_cchunk_count", "k_cchunk_fill", "k_gchunk_count",
              "k_gchunk_fill"):
        assert any(("pcplan" in name and k in name) for name in blocks), k


# -------------------------------------------------------------- FLAG / MAPQ (round 5)
def test_packed_reads_expose_what_filters_look_at():
    """Filters take the read (genome_array.py:697-722): a PackedRead answers for flag, mapping_quality, query_length
    and pysam's is_* properties -- from the SAM columns when the file has them, else from the strand bit alone."""
    p = PackedAlignments.from_cigars([0, 0, 0], [5, 9, 30], ["20M", "5S10M3N10M", "25M"], [False, True, False],
                                     references=["c"], lengths=[1000], flag16=[0x1 | 0x2 | 0x40, 0x10 | 0x100, 0x400 | 0x200 | 0x800],
                                     mapq=[60, 0, 255], qlen=[20, 25, 25])
    r0, r1, r2 = (p.read(i) for i in range(3))
    assert (r0.flag, r0.mapping_quality, r0.mapq, r0.query_length) == (0x43, 60, 60, 20)
    assert r0.is_paired and r0.is_proper_pair and r0.is_read1 and not (r0.is_read2 or r0.is_secondary or r0.is_unmapped)
    assert r1.is_reverse and r1.is_secondary and not r1.is_duplicate and r1.query_length == 25 and r1.positions[:2] == [9, 10]
    assert r2.is_duplicate and r2.is_qcfail and r2.is_supplementary and r2.mapping_quality == 255
    q = p.subset([2, 0], validate=False)
    assert list(q.flag16) == [0xe00, 0x43] and list(q.mapq) == [255, 60] and list(p.slice(1, 3).qlen) == [25, 25]
    bare = PackedAlignments.from_ungapped(0, [3, 7], [30, 31], [False, True], references=["c"], lengths=[100])
    assert bare.flag16 is None and (bare.read(1).flag, bare.read(1).mapping_quality, bare.read(1).query_length) == (0x10, 255, 31)
    assert not bare.read(0).is_secondary and bare.read(0).flag == 0
    with pytest.raises(ValueError):
        PackedAlignments.from_ungapped(0, [3, 7], [30, 31], [False, True], flag16=[0])
    # reads that know their flag (pysam.AlignedSegment does) keep it through from_reads
    class R(object):
        def __init__(self, pos, flag, mq):
            self.positions, self.is_reverse, self.flag, self.mapping_quality, self.query_length = list(range(pos, pos + 10)), bool(flag & 16), flag, mq, 12
    q = PackedAlignments.from_reads([R(5, 0x100, 3), R(8, 0x10, 40)], tids=[0, 0], references=["c"], lengths=[100])
    assert list(q.flag16) == [0x100, 0x10] and list(q.mapq) == [3, 40] and list(q.qlen) == [12, 12]
    from plastid_amd.packing import concat_file_major
    assert list(concat_file_major([q, q])["mapq"]) == [3, 40, 3, 40] and "mapq" not in concat_file_major([q, bare])


def test_flag_filter_factory():
    f = pa.FlagFilterFactory(exclude=["is_secondary", "is_duplicate"], min_mapq=10)
    assert (f.require, f.exclude, f.min_mapq) == (0, 0x500, 10)
    g = pa.FlagFilterFactory(require="is_proper_pair", exclude=0x200)
    assert (g.require, g.exclude, g.min_mapq) == (0x2, 0x200, 0)

    class R(object):
        def __init__(self, flag, mq):
            self.flag, self.mapping_quality = flag, mq
    assert f(R(0x10, 10)) and not f(R(0x10, 9)) and not f(R(0x100, 60)) and not f(R(0x400 | 0x10, 60))
    assert g(R(0x3, 0)) and not g(R(0x1, 0)) and not g(R(0x203, 0))
    with pytest.raises(TypeError):
        f(None)
    for bad in (dict(require="is_nice"), dict(exclude=0x10000), dict(min_mapq=256), dict(min_mapq=-1), dict(require=0x100, exclude=0x100)):
        with pytest.raises(ValueError):
            pa.FlagFilterFactory(**bad)


def test_auto_decode_falls_back_to_the_host_decoder(tmp_path, monkeypatch):
    """``decode="auto"`` asks the GPU decoder for large files; a file it rejects gets the host decoder's verdict (its
    arrays, or its exception); ``decode="gpu"`` keeps the hard failure.  Which decoder read the file is recorded."""
    from plastid_amd import bam, genome_array
    from tests import bam_writer
    path = str(tmp_path / "x.bam")
    bam_writer.write_bam(path, ["c"], [1000], [(0, 5, [(0, 30)], 0), (0, 50, [(0, 20)], 16)])
    calls = []

    def broken(p, engine, timing=None, regions=None):
        calls.append(p)
        raise ValueError("BGZF inflate failed in %s" % p)
    monkeypatch.setattr(bam, "read_bam_gpu", broken)
    monkeypatch.setattr(genome_array, "GPU_DECODE_MIN_BYTES", 1)
    aln = genome_array._open_alignment_source(path, engine=object(), decode="auto")
    assert calls == [path] and aln.decoder == "host" and aln.n == 2
    with pytest.raises(ValueError):
        genome_array._open_alignment_source(path, engine=object(), decode="gpu")
    monkeypatch.setattr(bam, "read_bam_gpu", lambda p, engine, timing=None, regions=None: bam.read_bam(p))
    assert genome_array._open_alignment_source(path, engine=object(), decode="auto").decoder == "gpu"
    open(path, "wb").write(b"not a bam")
    monkeypatch.setattr(bam, "read_bam_gpu", broken)
    with pytest.raises((ValueError, IOError)):
        genome_array._open_alignment_source(path, engine=object(), decode="auto")


def test_every_kernel_header_triggers_a_rebuild(tmp_path, monkeypatch):
    """``needs_build()`` looks at every header under csrc/ (a stale .so would travel to the GPU box with a newer
    kernel source): touching any of them -- the round-4 ones included -- makes it true."""
    from plastid_amd import build
    csrc = os.path.join(ROOT, "plastid_amd", "csrc")
    hdrs = [f for f in os.listdir(csrc) if f.endswith(".h")]
    assert {"pc_kernels.hip.h", "bam_kernels.hip.h", "plan_kernels.hip.h", "stage_kernels.hip.h", "host_util.h"} <= set(hdrs)
    assert set(os.path.basename(h) for h in build._headers()) >= set(hdrs)
    lib = tmp_path / "lib.so"
    lib.write_bytes(b"")
    monkeypatch.setattr(build, "LIB", str(lib))
    now = max(os.path.getmtime(h) for h in build._headers() + [build.SRC])
    os.utime(str(lib), (now + 10, now + 10))
    assert not build.needs_build()
    for h in hdrs:
        fake = tmp_path / h
        fake.write_text("// touched\n")
        os.utime(str(fake), (now + 20, now + 20))
        monkeypatch.setattr(build, "_headers", lambda fake=fake: [str(fake)])
        assert build.needs_build(), h
