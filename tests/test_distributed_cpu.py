"""World-size-2 gloo test of the N>1 path: chains are sharded across ranks with no
data-path collective; only summary totals are reduced.  The per-rank counting is done
by the oracle here (no GPU in this container); the sharding/reduction code is the
product's (plastid_amd/multigpu.py), the same bench.py uses with the nccl backend."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from plastid_amd import multigpu, synth
    from plastid_amd.packing import concat_file_major
    r, _, w = multigpu.init("gloo")
    assert (r, w) == (rank, world)
    genome, tx, reads, mapping = synth.make_config("C2", scale=0.0003, tx_scale=0.003)
    aln = concat_file_major([reads])
    mine = multigpu.shard_chains(tx.n, rank, world)
    sub = tx.subset(mine)
    p = sub.plan_arrays(rows=1)
    totals = []
    for kind, param in (("fiveprime", 12), ("center", 0)):
        spec = oracle.mapping_spec(kind, param)
        arrays, _ = oracle.count_segments(aln, spec, p["tid"], p["start"], p["end"], p["strand"])
        totals.append(sum(float(a.sum()) for a in arrays))
    n_int = multigpu.allreduce_int_totals([int(totals[0]), sub.n_positions, len(mine)])
    f_tot = multigpu.reduce_float_totals_ordered([totals[1]])
    t_max = multigpu.max_over_ranks(1.0 + rank)
    multigpu.barrier()
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.array(n_int + f_tot + [t_max, totals[1]], dtype=np.float64))


def test_two_rank_sharding_and_total_reduction(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npy" % r)) for r in range(world)]
    # every rank sees the same reduced totals
    assert np.array_equal(res[0][:5], res[1][:5])
    # and they equal the unsharded job
    from oracle import oracle
    from plastid_amd import multigpu, synth
    from plastid_amd.packing import concat_file_major
    genome, tx, reads, mapping = synth.make_config("C2", scale=0.0003, tx_scale=0.003)
    aln = concat_file_major([reads])
    p = tx.plan_arrays(rows=1)
    arrays, _ = oracle.count_segments(aln, oracle.mapping_spec("fiveprime", 12), p["tid"], p["start"], p["end"], p["strand"])
    assert int(res[0][0]) == int(sum(a.sum() for a in arrays))
    assert int(res[0][1]) == tx.n_positions and int(res[0][2]) == tx.n
    assert res[0][4] == 2.0  # max over ranks
    # float totals: fixed rank-order sum of the per-rank partials
    assert res[0][3] == res[0][5] + res[1][5]
    # shards are disjoint and cover all chains
    ids = np.concatenate([multigpu.shard_chains(tx.n, r, world) for r in range(world)])
    assert np.array_equal(ids, np.arange(tx.n))


# ---------------------------------------------------------------------------- genome-range partition
def _oracle_global(oracle, files, spec, p, rows, dtype):
    from plastid_amd.packing import concat_file_major
    arrays, _ = oracle.count_segments(concat_file_major(files), spec, p["tid"], p["start"], p["end"], p["strand"])
    out = np.zeros(p["out_elems"], dtype)
    for s, a in enumerate(arrays):
        a2 = a.reshape(rows, -1)
        for r in range(rows):
            out[p["out_off"][s] + r * p["row_stride"][s] + p["out_step"][s].astype(np.int64) * np.arange(a2.shape[1])] = a2[r]
    return out


def test_genome_partition_is_exact_for_every_rule():
    """Cutting records and segments at record-count quantiles of the genome and counting every
    range on its own reproduces the unpartitioned result bit for bit (center float sums included),
    for spliced human-scale reads whose halo spans introns."""
    from oracle import oracle
    from plastid_amd import multigpu, synth
    from plastid_amd.packing import concat_file_major
    genome, tx, reads, _ = synth.make_config("C4", scale=0.0002, tx_scale=0.003)
    second = reads.subset(np.arange(0, reads.n, 7))
    files = [reads, second]
    for kind, args, rows in (("fiveprime", (12,), 1), ("center", (3,), 1),
                             ("stratified", (0, synth.VARIABLE_OFFSETS, 27, 31), 5)):
        spec = oracle.mapping_spec(kind, *args)
        dtype = np.float64 if kind == "center" else np.int64
        p = tx.plan_arrays(rows=rows)
        want = _oracle_global(oracle, files, spec, p, rows, dtype)
        for world in (2, 5):
            part = multigpu.GenomePartition(files, p, world)
            assert len(part.cut_coordinates()) == world - 1
            got = np.zeros(p["out_elems"], dtype)
            staged = 0
            for r in range(world):
                mine = part.records(r)
                staged += sum(f.n for f in mine)
                lp = part.local_plan_arrays(r, rows)
                arr, _ = oracle.count_segments(concat_file_major(mine), spec, lp["tid"], lp["start"], lp["end"], lp["strand"])
                loc = np.concatenate([a.reshape(-1) for a in arr]) if len(arr) else np.zeros(0, dtype)
                assert len(loc) == lp["out_elems"]
                part.scatter_local(got, r, loc, rows)
            assert np.array_equal(got, want), (kind, world)
            assert staged >= sum(f.n for f in files)                      # halo duplicates, never drops
            # pieces tile every segment exactly once
            ln = np.zeros(len(p["tid"]), np.int64)
            np.add.at(ln, part.piece["owner"], part.piece["end"] - part.piece["start"])
            assert np.array_equal(ln, p["end"] - p["start"])


def _partition_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from plastid_amd import multigpu, synth
    from plastid_amd.packing import concat_file_major
    multigpu.init("gloo")
    genome, tx, reads, _ = synth.make_config("C4", scale=0.0002, tx_scale=0.003)
    # per-chain sums (out_step 0: the fused region statistics layout), chains straddle the cut
    p = tx.plan_arrays(rows=1)
    seg = dict(p, out_off=tx.ex_tx.astype(np.int64), out_step=np.zeros(len(p["tid"]), np.int8),
               row_stride=np.ones(len(p["tid"]), np.int64))
    part = multigpu.GenomePartition([reads], seg, world)
    sg = part.segments(rank)
    arr, _ = oracle.count_segments(concat_file_major(part.records(rank)), oracle.mapping_spec("threeprime", 0),
                                   sg["tid"], sg["start"], sg["end"], sg["strand"])
    sums = np.zeros(tx.n, np.int64)
    for j, a in enumerate(arr):
        sums[sg["out_off"][j]] += a.sum()
    total = multigpu.allreduce_chain_sums(sums)
    np.save(os.path.join(out_dir, "sums%d.npy" % rank), total)
    np.save(os.path.join(out_dir, "part%d.npy" % rank), sums)


def test_two_rank_genome_partition_chain_sums(tmp_path):
    world = 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_partition_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    tot = [np.load(os.path.join(str(tmp_path), "sums%d.npy" % r)) for r in range(world)]
    par = [np.load(os.path.join(str(tmp_path), "part%d.npy" % r)) for r in range(world)]
    assert np.array_equal(tot[0], tot[1]) and np.array_equal(tot[0], par[0] + par[1])
    assert ((par[0] > 0) & (par[1] > 0)).any()        # at least one chain really straddles the cut
    from oracle import oracle
    from plastid_amd import synth
    from plastid_amd.packing import concat_file_major
    genome, tx, reads, _ = synth.make_config("C4", scale=0.0002, tx_scale=0.003)
    p = tx.plan_arrays(rows=1)
    arr, _ = oracle.count_segments(concat_file_major([reads]), oracle.mapping_spec("threeprime", 0), p["tid"],
                                   p["start"], p["end"], p["strand"])
    want = np.zeros(tx.n, np.int64)
    for s, a in enumerate(arr):
        want[tx.ex_tx[s]] += a.sum()
    assert np.array_equal(tot[0], want)


# ---------------------------------------------------------------------------- the bench's one-job mode
def _bench_partition_worker(rank, world, port, out_dir):
    """What ``bench.py --gpus N --partition genome`` does per rank, with the oracle standing in for the
    engine: stage the rank's records, count its pieces in the rank-local layout, check the sampled
    elements it owns against the global expectation (``owned_elements``), sum its pieces per chain
    (``chain_sum_plan_arrays``) and complete the sums with one all-reduce."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from plastid_amd import multigpu, synth
    from plastid_amd.packing import concat_file_major
    multigpu.init("gloo")
    rows = 5
    genome, tx, reads, _ = synth.make_config("C4", scale=0.0002, tx_scale=0.003)
    spec = oracle.mapping_spec("stratified", 0, synth.VARIABLE_OFFSETS, 27, 31)
    p = tx.plan_arrays(rows=rows)
    part = multigpu.GenomePartition([reads], p, world)
    lp = part.local_plan_arrays(rank, rows)
    mine = concat_file_major(part.records(rank))
    arr, _ = oracle.count_segments(mine, spec, lp["tid"], lp["start"], lp["end"], lp["strand"])
    local = np.concatenate([a.reshape(-1) for a in arr]) if len(arr) else np.zeros(0, np.int64)
    sample = np.arange(0, len(p["tid"]), 3)
    li, gi = part.owned_elements(rank, rows, sample)
    np.save(os.path.join(out_dir, "own%d.npy" % rank), np.stack([gi, local[li]]))
    seg_chain = tx.ex_tx.astype(np.int64)
    sp = part.chain_sum_plan_arrays(rank, seg_chain, tx.n, rows)
    arr2, _ = oracle.count_segments(mine, spec, sp["tid"], sp["start"], sp["end"], sp["strand"])
    sums = np.zeros(sp["out_elems"], np.int64)
    for j, a in enumerate(arr2):
        for r in range(rows):
            sums[sp["out_off"][j] + r * sp["row_stride"][j]] += a.reshape(rows, -1)[r].sum()
    np.save(os.path.join(out_dir, "bsum%d.npy" % rank), multigpu.allreduce_chain_sums(sums))


def test_two_rank_bench_partition_path(tmp_path):
    world = 2
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_bench_partition_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from oracle import oracle
    from plastid_amd import synth
    from plastid_amd.packing import concat_file_major
    rows = 5
    genome, tx, reads, _ = synth.make_config("C4", scale=0.0002, tx_scale=0.003)
    spec = oracle.mapping_spec("stratified", 0, synth.VARIABLE_OFFSETS, 27, 31)
    p = tx.plan_arrays(rows=rows)
    want = _oracle_global(oracle, [reads], spec, p, rows, np.int64)
    covered = np.zeros(p["out_elems"], bool)
    for r in range(world):
        gi, val = np.load(os.path.join(str(tmp_path), "own%d.npy" % r))
        assert np.array_equal(want[gi], val)             # every owned element equals the unpartitioned job's
        assert not covered[gi].any()                     # and is owned by exactly one rank
        covered[gi] = True
    sample = np.arange(0, len(p["tid"]), 3)
    expect = np.zeros(p["out_elems"], bool)
    for s in sample:
        n = int(p["end"][s] - p["start"][s])
        for r in range(rows):
            expect[p["out_off"][s] + r * p["row_stride"][s] + p["out_step"][s].astype(np.int64) * np.arange(n)] = True
    assert np.array_equal(covered, expect)               # the ranks' pieces tile the sampled segments
    sums = [np.load(os.path.join(str(tmp_path), "bsum%d.npy" % r)) for r in range(world)]
    assert np.array_equal(sums[0], sums[1])
    chain_want = np.zeros(tx.n * rows, np.int64)
    arr, _ = oracle.count_segments(concat_file_major([reads]), spec, p["tid"], p["start"], p["end"], p["strand"])
    for s, a in enumerate(arr):
        chain_want[tx.ex_tx[s] * rows:(tx.ex_tx[s] + 1) * rows] += a.reshape(rows, -1).sum(axis=1)
    assert np.array_equal(sums[0], chain_want)


# ---------------------------------------------------------------------------- bench.py --gpus N, end to end
def test_range_addressable_generator_rank_slices_are_the_job():
    """``synth.make_reads_blocked``: the records a rank draws for its range (plus halo) are exactly that slice of the
    whole job, for any number of ranks; the owned records add up to the job; no record spans more than the halo."""
    from plastid_amd import synth
    for name, sc, txs in (("C2", 0.001, 0.01), ("C4", 0.0003, 0.005), ("C5", 0.0001, 0.005)):
        genome, tx, lay, _ = synth.job_layout(name, scale=sc, tx_scale=txs)
        whole = synth.make_reads_blocked(lay)
        assert whole.n == lay.n
        whole.validate()
        assert int((whole.ref_end() - whole.pos).max()) <= lay.halo
        lin = lay.tid_off[whole.tid] + whole.pos
        for world in (2, 5):
            cuts = lay.cuts(world)
            assert len(cuts) == world - 1 and np.all(np.diff(cuts) >= 0)
            owned = 0
            for r in range(world):
                lo, hi = lay.rank_range(cuts, r)
                mine = synth.make_reads_blocked(lay, max(0, lo - lay.halo), hi, block_reads=20000)
                i0, i1 = np.searchsorted(lin, max(0, lo - lay.halo), "left"), np.searchsorted(lin, hi, "left")
                ref = whole.slice(i0, i1)
                for k in ("tid", "pos", "alen", "flags", "nblk", "blk_start", "blk_len"):
                    assert np.array_equal(getattr(mine, k), getattr(ref, k)), (name, world, r, k)
                owned += int(((lay.tid_off[mine.tid] + mine.pos) >= lo).sum())
            assert owned == lay.n


def test_bench_spawns_its_ranks_and_runs_the_one_job_mode(tmp_path):
    """``python bench.py --gpus 2`` without a torchrun environment: the parent spawns the two ranks and relays ONE JSON
    line.  Rehearsed here without GPUs through tests/bench_rehearsal.py (the same main(), with the oracle-backed stand-in
    engine of tests/oracle_engine.py patched in from the test side; gloo): every rank generates only its range, passes
    its parity gates, the chain sums agree after the all-reduce, the records owned by the ranks add up to the job, C4 /
    C5 follow the headline as partitioned jobs, and the one stdout line stays within 4 KB with every config in it."""
    import json
    import subprocess
    env = dict(os.environ, PC_BENCH_BACKEND="gloo", PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    detail = str(tmp_path / "detail.json")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_rehearsal.py"), "--gpus", "2", "--scale", "0.002", "--tx-scale", "0.01",
           "--steps", "2", "--warmup", "1", "--other-configs", "C4,C5", "--parity-chains", "60", "--detail-out", detail]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert proc.returncode == 0, proc.stderr.decode()[-3000:]
    lines = [ln for ln in proc.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) <= 4096
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["metric"] == "mapped_reads_per_sec"
    assert "rehearsal" in d["config"] and d["value"] is None and d["value_first_count"] is None
    assert sum(d["config"]["partition"]["records_per_rank"]) == d["config"]["records"] == 200000
    assert set(d["configs"]) == {"C4", "C5"}
    for c in ("C4", "C5"):
        assert d["configs"][c]["parity_positions"] > 0 and d["configs"][c]["ms_per_step"] > 0
    # the prose and the full figures live in the side file
    full = json.load(open(detail))
    part = full["headline"]["partition"]
    assert sum(part["records_per_rank"]) == full["headline"]["records_total"] == 200000
    assert all(s >= o for s, o in zip(part["records_staged_per_rank"], part["records_per_rank"]))
    assert len(part["peak_host_rss_MB_per_rank"]) == 2 and part["allreduce"]["chains_checked_vs_oracle"] == 60
    for c in ("C4", "C5"):
        oc = full["other_configs"][c]
        assert sum(oc["partition"]["records_per_rank"]) == oc["records_total"]
        assert oc["partition"]["halo_positions"] > 1000 and "bit-exact" in oc["parity"]
    # bench.py has no switch that routes it around the HIP engine
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "PC_BENCH_ENGINE" not in src and "oracle_engine" not in src


def test_bench_one_job_from_one_shared_bam_file(tmp_path):
    """``bench.py --gpus 2 --from-bam``: rank 0 writes the job's records as ONE indexed BAM file, every rank stages its
    genome range of it (the BAI index; here the host reader feeds the stand-in engine, on a GPU box
    pc_add_alignment_bam_span) and passes the same parity gates as with generated records -- C2 (point rule), C3 (center
    rule: float64 sums in file order) and the spliced C4.  The line carries the per-rank file -> staged times."""
    import json
    import subprocess
    env = dict(os.environ, PC_BENCH_BACKEND="gloo", PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for cfg, n in (("C2", 100000), ("C3", 100000), ("C4", 100000)):
        detail = str(tmp_path / ("detail_%s.json" % cfg))
        cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_rehearsal.py"), "--gpus", "2", "--from-bam", "--config", cfg,
               "--scale", str(n / {"C2": 1e8, "C3": 1e8, "C4": 5e8}[cfg]), "--tx-scale", "0.01",
               "--steps", "2", "--warmup", "1", "--parity-chains", "60", "--detail-out", detail]
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert proc.returncode == 0, proc.stderr.decode()[-3000:]
        lines = [ln for ln in proc.stdout.decode().splitlines() if ln.strip()]
        assert len(lines) == 1 and len(lines[0]) <= 4096
        d = json.loads(lines[0])
        pt = d["config"]["partition"]
        assert d["n_gpus"] == 2 and pt["from_bam"] is True and len(pt["stage_ms_per_rank"]) == 2
        assert sum(pt["records_per_rank"]) == d["config"]["records"] == n
        full = json.load(open(detail))["headline"]
        assert "one shared BAM file" in full["partition"]["source"] and full["parity_positions"] > 0
        # every rank staged at least the records of its own range (reads reaching in from the left come on top)
        assert all(s >= o for s, o in zip(full["partition"]["records_staged_per_rank"], full["partition"]["records_per_rank"]))


def test_partition_element_maps_equal_the_piecewise_walk():
    """``owned_elements`` / ``scatter_local`` build their index arrays with repeats and running offsets (479 k exons x
    cuts at C4: no Python loop per piece); against the straightforward walk over pieces and rows, for laid-out
    (forward / reversed) and summed slices, several rows, a segment subset."""
    from plastid_amd import multigpu
    rng = np.random.default_rng(5)
    nseg, world, rows = 400, 3, 3
    tid = rng.integers(0, 3, nseg).astype(np.int32)
    start = rng.integers(0, 5000, nseg).astype(np.int64)
    length = rng.integers(0, 300, nseg)
    end = start + length
    step = rng.choice(np.array([1, -1, 0], np.int8), nseg, p=[0.5, 0.3, 0.2])
    off = np.zeros(nseg, np.int64)
    stride = np.where(step == 0, 1, length).astype(np.int64)
    at = 0
    for s in range(nseg):
        off[s] = at + (length[s] - 1 if step[s] < 0 else 0)
        at += rows * (1 if step[s] == 0 else int(length[s]))
    seg = dict(tid=tid, start=start, end=end, strand=np.ones(nseg, np.uint8), out_off=off, out_step=step, row_stride=stride)
    tid_off = np.array([0, 6000, 12000, 18000], np.int64)
    part = multigpu.GenomePartition.from_cuts(seg, world, tid_off, np.array([4100, 13050], np.int64), halo=40)
    total = np.zeros(at, np.int64)
    for rank in range(world):
        lp = part.local_plan_arrays(rank, rows)
        local = rng.integers(1, 1000, lp["out_elems"]).astype(np.int64)
        # the walk over pieces and rows
        want_li, want_gi = [], []
        ref = np.zeros(at, np.int64)
        for j, pi in enumerate(lp["piece_index"]):
            n = int(lp["end"][j] - lp["start"][j])
            if n <= 0:
                continue
            g0, st, rs = int(part.piece["out_off"][pi]), int(part.piece["out_step"][pi]), int(part.piece["row_stride"][pi])
            for r in range(rows):
                src = local[lp["out_off"][j] + r * n:lp["out_off"][j] + (r + 1) * n]
                if st == 0:
                    ref[g0 + r * rs] += src.sum()
                else:
                    ref[g0 + r * rs + st * np.arange(n)] = src
                want_li.append(lp["out_off"][j] + r * n + np.arange(n))
                want_gi.append(g0 + r * rs + st * np.arange(n))
        li, gi = part.owned_elements(rank, rows)
        assert np.array_equal(li, np.concatenate(want_li)) and np.array_equal(gi, np.concatenate(want_gi))
        got = part.scatter_local(np.zeros(at, np.int64), rank, local, rows)
        assert np.array_equal(got, ref)
        total += got
        some = rng.choice(nseg, 50, replace=False)
        li2, gi2 = part.owned_elements(rank, rows, segments=some)
        keep = np.isin(part.piece["owner"][lp["piece_index"]], some)
        sel = np.concatenate([np.full(rows * max(int(lp["end"][j] - lp["start"][j]), 0), keep[j]) for j in range(len(keep))]) if len(keep) else np.zeros(0, bool)
        assert np.array_equal(li2, li[sel]) and np.array_equal(gi2, gi[sel])
    assert total.sum() > 0


# ---------------------------------------------------------------------------- one shared BAM file, a region read per rank
def _file_shard_worker(rank, world, port, out_dir):
    """Every rank reads ITS genome range of one shared, indexed BAM file (``GenomePartition.rank_regions`` -> a region
    read through the BAI index: here the host reader, on a GPU box ``Engine.add_bam(path, regions=...)``), counts its
    pieces (the oracle stands in for the engine) and the chain sums are completed with one all-reduce."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from plastid_amd import multigpu, synth
    from plastid_amd.bam import read_bam
    from plastid_amd.packing import concat_file_major
    multigpu.init("gloo")
    genome, tx, reads, _ = synth.make_config("C4", scale=0.00008, tx_scale=0.003)
    path = os.path.join(out_dir, "shared.bam")
    p = tx.plan_arrays(rows=1)
    part = multigpu.GenomePartition([reads], p, world)           # (cuts from the record density; a real job takes them from the index)
    mine = read_bam(path, regions=part.rank_regions(rank, list(reads.references)))
    want = part.records(rank)[0]
    # the region read returns at least the records the partition names for the rank (plus reads reaching in from further left)
    key = lambda a: set(zip(a.tid.tolist(), a.pos.tolist(), a.alen.tolist(), a.flags.tolist()))  # noqa: E731
    assert key(want) <= key(mine) and mine.n <= reads.n
    spec = oracle.mapping_spec("center", 2)
    lp = part.local_plan_arrays(rank, 1)
    arr, _ = oracle.count_segments(concat_file_major([mine]), spec, lp["tid"], lp["start"], lp["end"], lp["strand"])
    local = np.concatenate([a.reshape(-1) for a in arr]) if len(arr) else np.zeros(0, np.float64)
    li, gi = part.owned_elements(rank, 1)
    np.save(os.path.join(out_dir, "fs_own%d.npy" % rank), np.stack([gi.astype(np.float64), local[li]]))
    np.save(os.path.join(out_dir, "fs_n%d.npy" % rank), np.array([mine.n, want.n]))


def test_two_ranks_stage_their_ranges_of_one_shared_file(tmp_path):
    from oracle import oracle
    from plastid_amd import synth
    from plastid_amd.packing import concat_file_major
    from tests import bam_writer
    genome, tx, reads, _ = synth.make_config("C4", scale=0.00008, tx_scale=0.003)
    path = os.path.join(str(tmp_path), "shared.bam")
    bam_writer.write_bam(path, list(reads.references), [int(x) for x in reads.lengths], bam_writer.packed_to_records(reads),
                         block_bytes=4000, index=True)
    world = 2
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_file_shard_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    p = tx.plan_arrays(rows=1)
    arr, _ = oracle.count_segments(concat_file_major([reads]), oracle.mapping_spec("center", 2), p["tid"], p["start"], p["end"], p["strand"])
    want = np.zeros(p["out_elems"], np.float64)
    for s, a in enumerate(arr):
        want[p["out_off"][s] + p["out_step"][s].astype(np.int64) * np.arange(a.shape[-1])] = a
    got = np.full(p["out_elems"], np.nan)
    staged = 0
    for r in range(world):
        own = np.load(os.path.join(str(tmp_path), "fs_own%d.npy" % r))
        idx = own[0].astype(np.int64)
        assert np.isnan(got[idx]).all()                 # every element is owned by exactly one rank
        got[idx] = own[1]
        n = np.load(os.path.join(str(tmp_path), "fs_n%d.npy" % r))
        staged += int(n[0])
        assert n[0] >= n[1]
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))      # center sums bit for bit: the ranks' records keep their order
    assert staged < 1.3 * reads.n                        # no rank read the whole file
