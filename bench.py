#!/usr/bin/env python
"""Headline benchmark: mapped reads/s of the per-position counting hot path.

    python bench.py --gpus 1 --steps K --warmup W            (one process)
    python bench.py --gpus N ...                             (spawns its N ranks itself: torch.distributed.run as a
                                                              child process, before this process touches a GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one pass of the hot path over one batch: every alignment record of
the synthetic BAM (already packed and resident in HBM) is counted under the
mapping rule into the per-position vectors of ALL transcripts of the annotation
(int64, every chain laid out 5'->3') -- what ``for chain in transcripts:
chain.get_counts(ga)`` does in the reference.  Headline workload = BASELINE.json
``configs[1]`` (C2): 100 M single-end reads, FivePrimeMapFactory(offset=12),
20 k yeast-scale transcripts, one MI355X.

N = 1 (default): after the headline, the same process runs the other GPU configs
of BASELINE.json -- C3 (center rule), whole C4 and whole C5 (they fit one GPU) --
each parity-gated against the oracle on a seeded sample of chains; their numbers
go to ``config.other_configs`` (``--other-configs none`` skips them).

N > 1: ONE job over N GPUs (``--partition genome``, strong scaling): the genome
is cut at quantiles of the expected record density (``synth.JobLayout.cuts``,
SURVEY 8e), every rank GENERATES and stages only its own range plus halo
(``synth.make_reads_blocked``: no rank ever holds the whole job) and counts its
segment pieces -- no data-path collective; per-chain sums are completed by one
RCCL all-reduce of the engine's device buffer, timed separately.  The headline
is followed by C4 and C5, partitioned the same way (``config.other_configs``).
``--partition replicas`` gives every rank an independent shard of the same
shape instead (weak scaling).

Before anything is timed the GPU output is compared bit-for-bit with the oracle
(``oracle/``) on a seeded sample of chains, and again after the timed steps; a
mismatch aborts the run.
"""
import argparse
import gc
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from plastid_amd import synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SIZE_FILTER = (25, 100)  # the CLI default SizeFilterFactory (argparsers.py:355-380, 678-679)

WORKLOAD_TEXT = {
    "C1": "C1: 1 M synthetic reads, FivePrimeMapFactory(offset=0), 200 yeast-scale SegmentChains",
    "C2": "C2: 100 M synthetic single-end reads, FivePrimeMapFactory(offset=12), 20 k yeast-scale transcripts",
    "C3": "C3: 100 M synthetic reads, CenterMapFactory(nibble=0), 20 k yeast-scale transcripts",
    "C4": "C4: 500 M synthetic reads, VariableFivePrimeMapFactory, 60 k human-scale transcripts",
    "C5": "C5: 1 B paired-end records, StratifiedVariableFivePrimeMapFactory(25..35), 60 k human-scale transcripts",
}


def oracle_spec(oracle, mapping, size_filter=None):
    kind = mapping[0]
    if kind in ("fiveprime", "threeprime", "center"):
        return oracle.mapping_spec(kind, mapping[1], size_filter=size_filter)
    if kind == "variable":
        return oracle.mapping_spec(kind, 0, mapping[1], size_filter=size_filter)
    return oracle.mapping_spec(kind, 0, mapping[1], mapping[2], mapping[3], size_filter=size_filter)


def sparse_expected(arrays, p, sel, rows):
    """Oracle per-segment arrays -> (sorted element indices of the plan's output layout, values)."""
    idxs, vals = [], []
    for arr, s in zip(arrays, sel):
        n = arr.shape[-1]
        idx = p["out_off"][s] + p["out_step"][s].astype(np.int64) * np.arange(n)
        a2 = arr.reshape(rows, n)
        for r in range(rows):
            idxs.append(idx + r * p["row_stride"][s])
            vals.append(a2[r])
    if not idxs:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    idx = np.concatenate(idxs)
    val = np.concatenate(vals)
    o = np.argsort(idx, kind="stable")
    return idx[o], val[o]


def segments_of_chains(tx, chains):
    return np.concatenate([np.arange(tx.ex_off[c], tx.ex_off[c + 1]) for c in chains]) if len(chains) else np.zeros(0, np.int64)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(oracle, prep, t_prep, spec, p, tx, n_records, budget_s, rng, min_chains=1000):
    """Time the oracle (C port of the reference algorithm, one thread) on a bounded, seeded sample of chains and
    report whole-job-equivalent reads/s.  The per-record arrays (end coordinates, contig ranges) are derived ONCE
    (`prep`, an oracle.Prepared; `t_prep` = what that took) -- the oracle's stand-in for opening and indexing the BAM
    file -- and enter the whole-job estimate once: time = preparation + counting time / sampled fraction.  At least
    `min_chains` chains are counted whatever the budget (the sparse configs touch few records per chain: a handful of
    chains says nothing).  Returns the sampled segments and their oracle arrays too: they are the parity gate of the
    GPU run."""
    order = rng.permutation(tx.n)

    def run(chains, threads=1):
        sel = segments_of_chains(tx, chains)
        t0 = time.perf_counter()
        arrays, _ = oracle.count_segments(prep, spec, p["tid"][sel], p["start"][sel], p["end"][sel], p["strand"][sel], threads=threads)
        return sel, arrays, time.perf_counter() - t0
    n_cal = min(tx.n, 20)
    sel0, arr0, t_cal = run(order[:n_cal])
    per_chain = max(t_cal, 1e-7) / n_cal
    n_main = int(max(0, (budget_s - t_prep - t_cal) / per_chain))
    n_main = max(n_main, min(min_chains, int(4 * budget_s / per_chain)) - n_cal)     # >= min_chains unless that alone takes 4 budgets
    n_main = int(min(tx.n - n_cal, max(0, n_main)))
    results = [(sel0, arr0)]
    t_work, done = max(t_cal, 1e-7), n_cal
    if n_main > 0:
        sel1, arr1, t_main = run(order[n_cal:n_cal + n_main])
        results.append((sel1, arr1))
        t_work += max(t_main, 1e-7)
        done += n_main
    frac = done / float(tx.n)
    sel_all = np.concatenate([s for s, _ in results])
    arrays_all = [a for _, arrs in results for a in arrs]
    cpu = {"value": n_records / (t_prep + t_work / frac), "unit": "reads/s", "cores": 1, "kind": "port",
           "value_excluding_preparation": n_records * frac / t_work, "chains_sampled": int(done),
           "sample": "oracle/plastid_oracle.c (C port of the reference algorithm, 1 thread) over %d of %d transcripts "
                     "(seeded sample) against all %d records: %.2f s of counting + %.2f s once-per-file preparation; "
                     "value = records / (preparation + counting time / sampled fraction)" % (done, tx.n, n_records, t_work, t_prep),
           "preparation_s": t_prep, "counting_s": t_work}
    # the SAME chains on every host core this process may use (usable_cpus: the GPU boxes grant a CFS quota well below
    # the hardware threads they show): whole segments dealt to one POSIX thread per core -- the reference itself is
    # single-threaded, so this is the most favourable honest scaling; the preparation stays single-threaded
    cores = usable_cpus()
    if cores >= 2:
        _, _, wall = run(order[:done], threads=cores)
        cpu["all_cores"] = {"value": n_records / (t_prep + wall / frac), "unit": "reads/s", "cores": cores,
                            "value_excluding_preparation": n_records * frac / wall, "counting_s": wall,
                            "sample": "the same %d transcripts dealt to %d threads of one process: %.2f s wall of counting (one core: "
                                      "%.2f s) + the same %.2f s of preparation" % (done, cores, wall, t_work, t_prep)}
    else:
        cpu["all_cores"] = None
    return cpu, sel_all, arrays_all, order[:done]


def usable_cpus():
    """CPUs this process may actually use: hardware threads, affinity mask and the container's CFS
    quota (cgroup v2 cpu.max / v1 cfs_quota_us), whichever is smallest."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = period = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota, period = int(q), int(per)
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            pass
    if quota and period and quota > 0:
        n = min(n, max(1, -(-quota // period)))
    return n


def warm_runtime(ctx):
    """Once per process, before anything is timed: a throwaway engine stages 100 k spliced records and counts them, so
    that the one-time costs of the process (code-object load of every kernel, the page-locked bounce buffers of pageable
    copies and the engine's own ring of page-locked pieces, hipcub's first temporary allocations) are not billed to the first config's `host_stage_s` / staged scope --
    the warm-up steps of the timed loop do the same for the kernel scope.  Reported as `config.runtime_warmup_s`."""
    Engine, rehearsal = engine_class()
    if ctx.get("runtime_warmup_s") is not None or rehearsal:
        return
    t0 = time.perf_counter()
    _g, tx, reads, mapping = synth.make_config("C4", scale=0.0002, tx_scale=0.002)
    factory = synth.mapping_factory(mapping)
    p = tx.plan_arrays(rows=1)
    eng = Engine(ctx["dev_index"])
    eng.set_alignments([reads])
    factory._configure(eng)
    plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
    plan.count(np.int64)
    plan.close()
    eng.close()
    # ... and the process-wide ring of page-locked pieces that large staging calls and read-backs go through (128 MB,
    # page-locked on first use: as long as staging ten million records) -- ten million single-run reads on one contig
    from plastid_amd.packing import PackedAlignments
    m = 10_000_000
    big = PackedAlignments.from_ungapped(0, np.arange(m, dtype=np.int32) // 4, np.full(m, 30, np.uint16), np.zeros(m, bool),
                                         references=["w"], lengths=[m], validate=False)
    eng = Engine(ctx["dev_index"])
    eng.set_alignments([big])
    eng.close()
    ctx["runtime_warmup_s"] = round(time.perf_counter() - t0, 3)


def first_count(eng, plan, out_dtype):
    """The first count of a plan also builds the plan's work lists (k_tile_ranges: which records every window scans --
    a function of the annotation, the staged alignments and the rule's halo, not of the counts); later counts of the
    plan reuse them.  Its cost, like the plan build, is paid once per (annotation, alignments) and reported apart:
    ms of the whole first count and of its work-list part (HIP events on the engine's stream)."""
    eng.set_profiling(2)
    plan.launch(out_dtype)
    eng.sync()
    t = eng.last_timing()
    eng.set_profiling(0)
    return {"total": round(t["total"], 4), "work_lists": round(t["worklist"], 4)}


def first_count_of_fresh_plans(eng, make_plan, out_dtype, n=3):
    """`first_count` on `n` plans built one after the other from the same annotation (each closed before the next is
    built, the last one kept): the figure is the MEDIAN first count, every sample is listed.  (One sample is at the mercy
    of what the first launch of a plan allocates: its output array and work lists come from the engine's pool when a
    block of the size is there and from hipMalloc -- milliseconds, on the host, with the stream idle between the events --
    when not; which of the two a fresh process meets differs from box to box.)  Returns (plan, figures)."""
    samples, plan = [], None
    for k in range(n):
        if plan is not None:
            plan.close()
        plan = make_plan()
        samples.append(first_count(eng, plan, out_dtype))
    mid = sorted(samples, key=lambda x: x["total"])[len(samples) // 2]
    return plan, dict(mid, samples=[x["total"] for x in samples])


def kernel_source_hash():
    """sha256 (first 16 hex digits) of the device code (every kernel lives in pc_kernels.hip.h): what a PMC-derived
    traffic figure belongs to.  (The host file -- staging, plan build, launches -- is not part of it.)"""
    import hashlib
    h = hashlib.sha256()
    for rel in ("plastid_amd/csrc/pc_kernels.hip.h",):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_traffic(config, n_records):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/traffic.json) -- only
    when the entry was measured on THESE kernel sources (`kernel_source_sha16`); a changed kernel reports null."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(tpath))
    except Exception:
        return None, None
    for entry in ([tj] + list(tj.get("configs", {}).values())) if isinstance(tj, dict) else []:
        if entry.get("config") == config and int(entry.get("n_records", -1)) == n_records:
            if entry.get("kernel_source_sha16") != kernel_source_hash():
                return None, entry
            return entry.get("hbm_bytes_per_launch"), entry
    return None, None


def peak_rss_mb():
    import resource
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


# Set ONLY by tests/bench_rehearsal.py (a test-side wrapper that imports this module): a stand-in engine class for
# rehearsing the multi-rank control flow in a container without GPUs.  bench.py itself has no switch, environment
# variable or flag that routes a run anywhere but through the HIP engine.
ENGINE_OVERRIDE = None


def engine_class():
    """(engine class, rehearsal?) -- plastid_amd.engine.Engine: HIP, no CPU path."""
    if ENGINE_OVERRIDE is not None:
        return ENGINE_OVERRIDE, True
    from plastid_amd.engine import Engine
    return Engine, False


def run_workload(name, args, ctx, headline):
    """One BASELINE config through the hot path.  Returns the result dict of that config."""
    from oracle import oracle
    from plastid_amd import multigpu
    from plastid_amd.packing import concat_file_major
    Engine, _ = engine_class()
    rank, world = ctx["rank"], ctx["world"]
    partition = ctx["partition"] if world > 1 else "none"
    steps = args.steps if headline else max(3, min(args.steps, 10))
    warmup = max(args.warmup, 1) if headline else 2
    scale = args.scale
    tx_scale = args.tx_scale

    # ---------------------------------------------------------------- inputs (host)
    t0 = time.perf_counter()
    genome, tx, reads, mapping = synth.make_config(name, scale=scale, tx_scale=tx_scale,
                                                   seed_shift=rank if partition == "replicas" else 0)
    gen_s = time.perf_counter() - t0
    center = mapping[0] == "center"
    out_dtype = np.float64 if (center or args.out_dtype == "float64") else np.int64
    factory = synth.mapping_factory(mapping)
    rows = getattr(factory, "_numlengths", 1)
    p = tx.plan_arrays(rows=rows)
    n_job = int(reads.n)                      # records of the whole job this rank belongs to

    # ---------------------------------------------------------------- oracle: CPU baseline and parity sample
    cpu = None
    aln = concat_file_major([reads])
    spec = oracle_spec(oracle, mapping)
    rng = np.random.default_rng(7 + (rank if partition == "replicas" else 0))
    t0 = time.perf_counter()
    prep = oracle.Prepared(aln)          # per-record arrays of the oracle, derived once (its "open the BAM file")
    t_prep = time.perf_counter() - t0
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # (beside every config: the other configs get ~6 s per figure so that the default run stays within minutes)
        budget = args.cpu_budget if headline else min(args.cpu_budget, 6.0)
        cpu, check_sel, check_arrays, _ = cpu_baseline(oracle, prep, t_prep, spec, p, tx, reads.n, budget, rng)
        cpu["cpu_model"] = cpu_model()
        cpu["host_cores"] = os.cpu_count()
        cpu["usable_cores"] = usable_cpus()
    else:
        chains = rng.permutation(tx.n)[:min(tx.n, args.parity_chains)]
        check_sel = segments_of_chains(tx, chains)
        check_arrays, _ = oracle.count_segments(prep, spec, p["tid"][check_sel], p["start"][check_sel],
                                                p["end"][check_sel], p["strand"][check_sel], threads=usable_cpus())
    # the SizeFilterFactory(25,100) variant is gated on a small sample of its own
    sf_chains = np.random.default_rng(9).permutation(tx.n)[:min(tx.n, 200)]
    sf_sel = segments_of_chains(tx, sf_chains)
    sf_arrays, _ = oracle.count_segments(prep, oracle_spec(oracle, mapping, SIZE_FILTER), p["tid"][sf_sel], p["start"][sf_sel],
                                         p["end"][sf_sel], p["strand"][sf_sel], threads=usable_cpus())
    prep.close()
    del aln, prep
    exp = sparse_expected(check_arrays, p, check_sel, rows)        # (element indices, values) of the sampled chains
    sf_exp = sparse_expected(sf_arrays, p, sf_sel, rows)
    del check_arrays, sf_arrays

    my_reads, lp = reads, p     # (N > 1 here means independent replicas; the one-job mode is run_partitioned)

    # ---------------------------------------------------------------- GPU
    import torch
    warm_runtime(ctx)
    eng = Engine(ctx["dev_index"])
    time.sleep(0.3)   # the all-cores CPU baseline just burnt the cgroup's CPU quota (CFS, 100 ms periods): staging is a host pass
    t0 = time.perf_counter()
    eng.set_alignments([my_reads])
    stage_s = time.perf_counter() - t0
    factory._configure(eng)
    t0 = time.perf_counter()
    plan = eng.plan(lp["tid"], lp["start"], lp["end"], lp["strand"], lp["out_off"], lp["out_step"], lp["row_stride"],
                    lp["out_elems"], rows)
    plan_s = time.perf_counter() - t0

    # host side of the read-back: page-locked memory from the engine itself (pc_host_alloc: what a caller that reads
    # repeatedly hands over); above 16 GiB ordinary pageable memory, which goes down through the transfer ring
    if int(lp["out_elems"]) * 8 <= (16 << 30) and hasattr(eng, "host_buffer"):
        out_buf = eng.host_buffer(int(lp["out_elems"]), out_dtype)
        out_buf[:] = 0
    else:
        out_buf = np.zeros(int(lp["out_elems"]), out_dtype)

    def gate(expected, what):
        """Bit-exact comparison of the sampled elements with the oracle."""
        got = plan.read(out_buf)
        e_idx, e_val = expected
        if not np.array_equal(got[e_idx], e_val.astype(got.dtype)):
            raise SystemExit("PARITY FAILURE (%s, %s): HIP counts differ from the oracle on the sampled chains" % (name, what))
        return len(e_idx)

    plan.close()
    plan, first_count_ms = first_count_of_fresh_plans(eng, lambda: eng.plan(lp["tid"], lp["start"], lp["end"], lp["strand"], lp["out_off"], lp["out_step"],
                                                                             lp["row_stride"], lp["out_elems"], rows), out_dtype)
    for _ in range(warmup):
        plan.launch(out_dtype)
    eng.sync()
    t0 = time.perf_counter()
    plan.read(out_buf)
    read_first_s = time.perf_counter() - t0   # the first copy into this buffer (whatever the runtime does once per buffer is in it)
    t0 = time.perf_counter()
    plan.read(out_buf)
    read_s = time.perf_counter() - t0   # D2H of every output position
    n_checked = gate(exp, "before timing")
    total_counts = plan.total()

    # ---------------------------------------------------------------- timed region
    multigpu.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.launch(out_dtype)
    eng.sync()
    torch.cuda.synchronize()
    multigpu.barrier()
    elapsed = multigpu.max_over_ranks(time.perf_counter() - t0, device=ctx["tdev"])
    gate(exp, "output of the last timed step")

    # ---------------------------------------------------------------- per-kernel timing (HIP events on the engine's stream)
    # The timed region above runs the product default (no events: each hipEventRecord costs a few
    # microseconds of stream time).  The same launches are repeated here with the engine's phase
    # events switched on; the dominant kernel's average duration feeds the roofline object.
    phases = {"total": 0.0, "worklist": 0.0, "hist": 0.0, "long": 0.0, "gather": 0.0, "zero": 0.0}
    m = max(3, min(steps, 20))
    eng.set_profiling(2)
    for _ in range(m):
        plan.launch(out_dtype)
        eng.sync()
        for k, v in eng.last_timing().items():
            phases[k] += v / m
    eng.set_profiling(0)
    alg_bytes_step = eng.last_algorithmic_bytes()

    # ---------------------------------------------------------------- the CLI-default size filter (one extra run per config)
    eng.set_size_filter(*SIZE_FILTER)
    plan.launch(out_dtype)
    eng.sync()
    n_sf = gate(sf_exp, "SizeFilterFactory(25,100)")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.launch(out_dtype)
    eng.sync()
    sf_elapsed = multigpu.max_over_ranks(time.perf_counter() - t0, device=ctx["tdev"])
    eng.set_size_filter(None)

    # ---------------------------------------------------------------- the same records dealt alternately into TWO files
    # (BAMGenomeArray(*bamfiles), genome_array.py:657-660, 800-809: several files count as one; joint windows)
    two_files = None
    if (headline or center) and world == 1 and not args.no_two_files:   # (the center rule too: its several-files form is the same kernel, one descriptor per entry and file)
        from plastid_amd.packing import PackedAlignments
        multi = np.nonzero(my_reads.nblk >= 2)[0]
        rec_of_run = np.repeat(multi, my_reads.nblk[multi])
        halves = []
        for k in (0, 1):
            sel = np.arange(k, my_reads.n, 2)
            runs = np.nonzero((rec_of_run & 1) == k)[0]
            halves.append(PackedAlignments(my_reads.tid[sel], my_reads.pos[sel], my_reads.alen[sel], my_reads.flags[sel], my_reads.nblk[sel],
                                           my_reads.blk_start[runs], my_reads.blk_len[runs], references=my_reads.references,
                                           lengths=my_reads.lengths, validate=False))
        eng.set_alignments(halves)
        for _ in range(2):
            plan.launch(out_dtype)
        eng.sync()
        if center:
            # float64 sums in FILE-MAJOR read order (itertools.chain over the files' fetches, genome_array.py:800-809): not
            # the one-file order -- the expectation is the oracle's on the two files, for a sample of chains of its own
            prep2 = oracle.Prepared(concat_file_major(halves))
            sel2 = segments_of_chains(tx, np.random.default_rng(11).permutation(tx.n)[:min(tx.n, 300)])
            arr2, _ = oracle.count_segments(prep2, spec, p["tid"][sel2], p["start"][sel2], p["end"][sel2], p["strand"][sel2], threads=usable_cpus())
            prep2.close()
            gate(sparse_expected(arr2, p, sel2, rows), "two files")
            del prep2, arr2
        else:
            gate(exp, "two files")                   # two files count as their sum: the one-file expectation
        t0 = time.perf_counter()
        for _ in range(steps):
            plan.launch(out_dtype)
        eng.sync()
        two_ms = (time.perf_counter() - t0) / steps * 1e3
        two_files = {"what": "the same records dealt alternately into two files (point rules: joint windows, gated on the one-file oracle sample; center rule: one descriptor per entry and file, gated on the oracle's file-major sums of 300 chains)",
                     "ms_per_step": two_ms, "ratio_to_one_file": two_ms / (elapsed / steps * 1e3)}
        eng.set_alignments([my_reads])
        del halves

    # ---------------------------------------------------------------- collectives
    n_records_all, counts_all, positions_all = multigpu.allreduce_int_totals(
        [int(my_reads.n), int(total_counts) if not center else 0, int(lp["out_elems"])], device=ctx["tdev"])
    if center:
        counts_all = multigpu.reduce_float_totals_ordered([float(total_counts)], device=ctx["tdev"])[0]

    # ---------------------------------------------------------------- result of this config
    n_extra_runs = int(len(my_reads.blk_start))
    n_seg_local = int(len(lp["tid"]))
    if center:
        # k_center (fused since round 3): streams the center stream of the records (8 B per aligned run), reads the
        # segment table and writes every output position once -- the same B_alg as the point rules
        kern_alg_bytes = my_reads.n * 8 + n_extra_runs * 8 + n_seg_local * 24 + int(lp["out_elems"]) * 8
        kernel_name = "k_center"
    else:
        # dominant kernel = k_hist_point (fused): streams every packed record once (8 B) + the runs of
        # gapped records (8 B each) + the segment table (24 B each) and writes every output position
        # once (8 B) -- exactly SURVEY section 8(d)'s B_alg for the step
        kern_alg_bytes = my_reads.n * 8 + n_extra_runs * 8 + n_seg_local * 24 + int(lp["out_elems"]) * 8
        kernel_name = "k_hist_point"
    kern_ms = phases["hist"]
    achieved = kern_alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    traffic, _ = load_traffic(name, int(my_reads.n)) if world == 1 else (None, None)
    ms_per_step = elapsed / steps * 1e3
    value = n_records_all * steps / elapsed
    res = {
        "workload": WORKLOAD_TEXT[name] + ("" if scale == 1.0 else " [scaled x%g]" % scale),
        "value": value, "ms_per_step": ms_per_step, "steps": steps, "warmup": warmup,
        "dtype": "float64" if out_dtype == np.float64 else "int64",
        "records_per_gpu": int(my_reads.n), "records_total": int(n_records_all),
        "chains": int(tx.n), "segments": int(tx.n_segments), "output_positions_per_gpu": int(lp["out_elems"]),
        "island_positions": int(plan.positions), "tiles": int(plan.tiles), "rows": rows,
        "mapping": [str(x) for x in mapping], "read_seed": synth.CONFIGS[name][5], "transcript_seed": synth.CONFIGS[name][3],
        "positions_per_sec": positions_all * steps / elapsed,
        "parity": "bit-exact vs oracle on %d output positions (seeded sample of chains), before the timed steps and on "
                  "the output of the last one" % n_checked,
        "parity_positions": int(n_checked),
        "sum_of_counts_all_ranks": counts_all,
        "host_generate_s": round(gen_s, 2), "host_stage_s": round(stage_s, 3), "host_read_outputs_s": round(read_s, 4),
        "host_read_outputs_first_s": round(read_first_s, 4),
        "plan_build_ms_once_per_annotation": round(plan_s * 1e3, 2),
        "first_count_ms": first_count_ms,
        "algorithmic_bytes_per_step": int(alg_bytes_step),
        "step_GBps_algorithmic": alg_bytes_step / (ms_per_step * 1e-3) / 1e9,
        "kernel_ms": {k: round(v, 4) for k, v in phases.items()},
        "size_filter_variant": {"filter": "SizeFilterFactory(25,100)", "ms_per_step": sf_elapsed / steps * 1e3,
                                "reads_per_s": n_records_all * steps / sf_elapsed,
                                "parity": "bit-exact vs oracle on %d positions" % n_sf},
        # kernel scope = `value`; staged = + H2D staging of the packed arrays and D2H of every
        # output; both from host buffers of the job (PCIe inclusive, never the headline value)
        "scopes": {"kernel_reads_per_s": value,
                   "staged_reads_per_s": my_reads.n / (stage_s + ms_per_step * 1e-3 + read_s) * (world if partition != "none" else 1),
                   # ... and with what is paid once per annotation on top: the plan build and the first count's work lists
                   "staged_with_plan_reads_per_s": my_reads.n / (stage_s + plan_s + first_count_ms["total"] * 1e-3 + read_s) *
                                                   (world if partition != "none" else 1)},
        "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "frac_traffic": (traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if (traffic and kern_ms > 0) else None,
                     "algorithmic_bytes_per_launch": int(kern_alg_bytes), "avg_launch_ms": kern_ms,
                     "basis": "achieved / frac: ALGORITHMIC bytes of SURVEY 8(d) (8 B per record of the whole file + runs + "
                              "segments + outputs) over the kernel's time -- not the bytes moved: the kernel streams 4 B per "
                              "record, so this can exceed the measured stream rate; frac_traffic is the PMC-measured figure"},
        "cpu_baseline": cpu,
        "two_files": two_files,
    }
    ctx["last_engine_objects"] = (eng, plan, my_reads)
    ctx["last_annotation"] = tx
    return res


def run_partitioned(name, args, ctx, headline):
    """One BASELINE config as ONE job over the ranks (N > 1, ``--partition genome``): every rank generates, stages
    and counts only its own genome range.  Parity: every rank compares the sampled segment pieces it
    owns with the oracle run on ITS records (the cut is exact for all rules, tests/test_distributed_cpu.py), the
    per-chain sums after the device all-reduce with the all-reduced oracle sums."""
    from oracle import oracle
    from plastid_amd import multigpu
    from plastid_amd.packing import concat_file_major
    Engine, rehearsal = engine_class()
    rank, world = ctx["rank"], ctx["world"]
    steps = args.steps if headline else max(3, min(args.steps, 10))
    warmup = max(args.warmup, 1) if headline else 2

    # ---------------------------------------------------------------- this rank's share of the inputs (host)
    t0 = time.perf_counter()
    genome, tx, lay, mapping = synth.job_layout(name, scale=args.scale, tx_scale=args.tx_scale)
    cuts = lay.cuts(world)
    lo, hi = lay.rank_range(cuts, rank)
    my_reads = synth.make_reads_blocked(lay, max(0, lo - lay.halo), hi)
    gen_s = time.perf_counter() - t0
    owned = int(((lay.tid_off[my_reads.tid] + my_reads.pos) >= lo).sum())
    center = mapping[0] == "center"
    out_dtype = np.float64 if (center or args.out_dtype == "float64") else np.int64
    factory = synth.mapping_factory(mapping)
    rows = getattr(factory, "_numlengths", 1)
    p = tx.plan_arrays(rows=rows)
    gp = multigpu.GenomePartition.from_cuts(p, world, lay.tid_off, cuts, lay.halo)
    lp = gp.local_plan_arrays(rank, rows)
    piece_owner = gp.piece["owner"][lp["piece_index"]]          # global segment of every local piece

    # ---------------------------------------------------------------- oracle on the rank's own records: parity sample
    aln = concat_file_major([my_reads])
    spec = oracle_spec(oracle, mapping)
    chains = np.random.default_rng(7).permutation(tx.n)[:min(tx.n, args.parity_chains)]      # the same chains on every rank
    sampled = np.zeros(len(p["tid"]), bool)
    sampled[segments_of_chains(tx, chains)] = True

    def local_expectation(sp, pieces):
        arrays, _ = oracle.count_segments(aln, sp, lp["tid"][pieces], lp["start"][pieces], lp["end"][pieces], lp["strand"][pieces])
        idxs, vals = [], []
        for a, j in zip(arrays, pieces):
            n = a.shape[-1]
            idxs.append((lp["out_off"][j] + (np.arange(rows)[:, None] * n + np.arange(n)[None, :])).reshape(-1))   # rank-local layout: [rows, len]
            vals.append(a.reshape(-1))
        if not idxs:
            return np.zeros(0, np.int64), np.zeros(0, out_dtype), arrays
        return np.concatenate(idxs), np.concatenate(vals), arrays
    pieces = np.nonzero(sampled[piece_owner])[0]
    exp_idx, exp_val, piece_arrays = local_expectation(spec, pieces)
    sf_chains = np.random.default_rng(9).permutation(tx.n)[:min(tx.n, 40)]
    sf_sampled = np.zeros(len(p["tid"]), bool)
    sf_sampled[segments_of_chains(tx, sf_chains)] = True
    sf_idx, sf_val, _ = local_expectation(oracle_spec(oracle, mapping, SIZE_FILTER), np.nonzero(sf_sampled[piece_owner])[0])
    # the oracle's share of the per-chain sums (completed across the ranks below)
    seg_chain = np.repeat(np.arange(tx.n, dtype=np.int64), np.diff(tx.ex_off))
    oracle_part = np.zeros(tx.n * rows, np.int64)
    if not center:
        for a, j in zip(piece_arrays, pieces):
            oracle_part[seg_chain[piece_owner[j]] * rows:(seg_chain[piece_owner[j]] + 1) * rows] += a.reshape(rows, -1).sum(axis=1)
    del aln, piece_arrays

    # ---------------------------------------------------------------- --from-bam: ONE shared file, written once (untimed)
    # What a user of the reference has is a BAM file, and the reference reads each region from it
    # (genome_array.py:800-809).  Rank 0 writes the job's records as a coordinate-sorted, indexed BAM; every rank then
    # stages ITS genome range of that file (GenomePartition.rank_regions -> the BAI index -> pc_add_alignment_bam_span:
    # only the BGZF members of the range are uploaded, inflated and decoded, on the GPU).  The records the rank
    # generated above stay the oracle's input.
    bam_path = None
    if getattr(args, "from_bam", False):
        from tests import bam_writer
        bam_path = os.path.join(tempfile.gettempdir(), "pc_bench_job_%s_%s.bam" % (name, os.environ.get("MASTER_PORT", str(os.getppid()))))
        if rank == 0:
            whole = my_reads if world == 1 else synth.make_reads_blocked(lay)
            bam_writer.write_bam_packed(bam_path, whole, threads=min(16, usable_cpus()), index=True)
            del whole
        multigpu.barrier()

    # ---------------------------------------------------------------- GPU
    import torch
    warm_runtime(ctx)
    eng = Engine(ctx["dev_index"])
    time.sleep(0.3)   # the all-cores CPU baseline just burnt the cgroup's CPU quota (CFS, 100 ms periods): staging is a host pass
    staged_n = int(my_reads.n)
    if bam_path:
        regions = gp.rank_regions(rank, list(my_reads.references))
        if hasattr(eng, "add_bam"):
            eng.clear_alignments()
            eng.add_bam(bam_path, regions=regions)             # page cache + library warm-up
            t0 = time.perf_counter()
            eng.clear_alignments()
            eng.add_bam(bam_path, regions=regions)
            stage_s = time.perf_counter() - t0
            staged_n = int(eng.num_records(0))
        else:                                                   # (the CPU rehearsal's stand-in engine: the host reader, same regions)
            from plastid_amd.bam import read_bam
            t0 = time.perf_counter()
            from_file = read_bam(bam_path, regions=regions)
            eng.set_alignments([from_file])
            stage_s = time.perf_counter() - t0
            staged_n = int(from_file.n)
        if staged_n < my_reads.n:
            raise SystemExit("--from-bam (%s, rank %d): the region read staged %d records, the rank's range holds %d" % (name, rank, staged_n, my_reads.n))
    else:
        t0 = time.perf_counter()
        eng.set_alignments([my_reads])
        stage_s = time.perf_counter() - t0
    factory._configure(eng)
    t0 = time.perf_counter()
    plan = eng.plan(lp["tid"], lp["start"], lp["end"], lp["strand"], lp["out_off"], lp["out_step"], lp["row_stride"],
                    lp["out_elems"], rows)
    plan_s = time.perf_counter() - t0
    out_buf = (eng.host_buffer(int(lp["out_elems"]), out_dtype) if (hasattr(eng, "host_buffer") and int(lp["out_elems"]) * 8 <= (16 << 30))
               else np.zeros(int(lp["out_elems"]), out_dtype))   # page-locked, as at N = 1
    out_buf[:] = 0

    def gate(idx, val, what):
        got = plan.read(out_buf)
        if not np.array_equal(got[idx], val.astype(got.dtype)):
            raise SystemExit("PARITY FAILURE (%s, rank %d, %s): HIP counts differ from the oracle on the sampled pieces" % (name, rank, what))
        return len(idx)
    plan.close()
    plan, first_count_ms = first_count_of_fresh_plans(eng, lambda: eng.plan(lp["tid"], lp["start"], lp["end"], lp["strand"], lp["out_off"], lp["out_step"],
                                                                             lp["row_stride"], lp["out_elems"], rows), out_dtype)
    for _ in range(warmup):
        plan.launch(out_dtype)
    eng.sync()
    t0 = time.perf_counter()
    plan.read(out_buf)
    read_s = time.perf_counter() - t0
    n_checked = gate(exp_idx, exp_val, "before timing")
    total_counts = plan.total()

    # ---------------------------------------------------------------- timed region
    multigpu.barrier()
    if not rehearsal:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.launch(out_dtype)
    eng.sync()
    if not rehearsal:
        torch.cuda.synchronize()
    multigpu.barrier()
    elapsed = multigpu.max_over_ranks(time.perf_counter() - t0, device=ctx["tdev"])
    gate(exp_idx, exp_val, "output of the last timed step")

    phases = {"total": 0.0, "worklist": 0.0, "hist": 0.0, "long": 0.0, "gather": 0.0, "zero": 0.0}
    m = max(3, min(steps, 20))
    eng.set_profiling(2)
    for _ in range(m):
        plan.launch(out_dtype)
        eng.sync()
        for k, v in eng.last_timing().items():
            phases[k] += v / m
    eng.set_profiling(0)
    alg_bytes_step = eng.last_algorithmic_bytes()

    eng.set_size_filter(*SIZE_FILTER)
    plan.launch(out_dtype)
    eng.sync()
    n_sf = gate(sf_idx, sf_val, "SizeFilterFactory(25,100)")
    eng.set_size_filter(None)

    # ---------------------------------------------------------------- collectives
    tot = multigpu.allreduce_int_totals([owned, int(total_counts) if not center else 0, int(lp["out_elems"]), n_checked, n_sf],
                                        device=ctx["tdev"])
    n_records_all, counts_all, positions_all, checked_all, sf_all = tot
    if n_records_all != lay.n:
        raise SystemExit("partition (%s): the ranks own %d records, the job has %d" % (name, n_records_all, lay.n))
    if center:
        counts_all = multigpu.reduce_float_totals_ordered([float(total_counts)], device=ctx["tdev"])[0]
    per_rank = lambda v: multigpu.allreduce_int_totals([int(v) if r == rank else 0 for r in range(world)], device=ctx["tdev"])
    balance, staged, rss = per_rank(owned), per_rank(staged_n), per_rank(peak_rss_mb())
    stage_ms_all = per_rank(round(stage_s * 1e3))
    if bam_path:
        multigpu.barrier()
        if rank == 0:
            for f in (bam_path, bam_path + ".bai"):
                try:
                    os.remove(f)
                except OSError:
                    pass
    gen_all = per_rank(round(gen_s * 1e3))
    allreduce = None
    if not center:
        sp = gp.chain_sum_plan_arrays(rank, seg_chain, tx.n, rows)
        splan = eng.plan(sp["tid"], sp["start"], sp["end"], sp["strand"], sp["out_off"], sp["out_step"], sp["row_stride"],
                         sp["out_elems"], rows)
        want = multigpu.allreduce_chain_sums(oracle_part, device=ctx["tdev"])     # the oracle's sums, completed the same way
        times = []
        for it in range(4):
            splan.launch(np.int64)
            eng.sync()
            multigpu.barrier()
            t0 = time.perf_counter()
            if ctx["backend"] == "nccl" and not rehearsal:
                torch.cuda.synchronize()
                t = multigpu.allreduce_device_sums(splan.device_ptr, tx.n * rows, "int64")
                torch.cuda.synchronize()
                sums = t.cpu().numpy() if it == 3 else None
            else:   # rehearsal on CPU ranks (gloo): through the host
                sums = multigpu.allreduce_chain_sums(splan.read(), device="cpu")
            times.append(time.perf_counter() - t0)
        nchk = 0
        for c in chains:
            if not np.array_equal(sums[c * rows:(c + 1) * rows], want[c * rows:(c + 1) * rows]):
                raise SystemExit("PARITY FAILURE (%s): all-reduced chain sums differ from the oracle" % name)
            nchk += 1
        allreduce = {"what": "per-chain sums, int64[%d], RCCL all-reduce of the engine's device buffer" % (tx.n * rows),
                     "ms": min(times[1:]) * 1e3, "chains_checked_vs_oracle": nchk}
        splan.close()

    n_extra_runs = int(len(my_reads.blk_start))
    kern_alg_bytes = my_reads.n * 8 + n_extra_runs * 8 + len(lp["tid"]) * 24 + int(lp["out_elems"]) * 8
    kern_ms = phases["hist"]
    achieved = kern_alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    ms_per_step = elapsed / steps * 1e3
    value = n_records_all * steps / elapsed
    res = {
        "workload": WORKLOAD_TEXT[name] + ("" if args.scale == 1.0 else " [scaled x%g]" % args.scale) + " -- one job over %d ranks" % world,
        "value": None if rehearsal else value, "ms_per_step": ms_per_step, "steps": steps, "warmup": warmup,
        "dtype": "float64" if out_dtype == np.float64 else "int64",
        "records_per_gpu": owned, "records_total": int(n_records_all),
        "chains": int(tx.n), "segments": int(tx.n_segments), "output_positions_per_gpu": int(lp["out_elems"]),
        "island_positions": int(plan.positions), "tiles": int(plan.tiles), "rows": rows,
        "mapping": [str(x) for x in mapping], "read_seed": synth.CONFIGS[name][5], "transcript_seed": synth.CONFIGS[name][3],
        "positions_per_sec": positions_all * steps / elapsed,
        "parity": "bit-exact vs oracle on %d output positions over all ranks (every rank: its pieces of a seeded sample of "
                  "chains against the oracle on its own records), before the timed steps and on the output of the last one" % checked_all,
        "parity_positions": int(checked_all),
        "sum_of_counts_all_ranks": counts_all,
        "host_generate_s": round(gen_s, 2), "host_stage_s": round(stage_s, 3), "host_read_outputs_s": round(read_s, 4),
        "plan_build_ms_once_per_annotation": round(plan_s * 1e3, 2),
        "first_count_ms": first_count_ms,
        "algorithmic_bytes_per_step": int(alg_bytes_step),
        "step_GBps_algorithmic": alg_bytes_step / (ms_per_step * 1e-3) / 1e9,
        "kernel_ms": {k: round(v, 4) for k, v in phases.items()},
        "size_filter_variant": {"filter": "SizeFilterFactory(25,100)", "parity": "bit-exact vs oracle on %d positions over all ranks" % sf_all},
        "scopes": {"kernel_reads_per_s": value,
                   "staged_reads_per_s": lay.n / (multigpu.max_over_ranks(stage_s + read_s, device=ctx["tdev"]) + ms_per_step * 1e-3)},
        "roofline": {"bound": "hbm", "kernel": "k_center" if center else "k_hist_point", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "what": "rank 0's launch: its records and outputs",
                     "algorithmic_bytes_per_launch": int(kern_alg_bytes), "avg_launch_ms": kern_ms},
        "cpu_baseline": None,
        "partition": {"mode": "genome ranges at quantiles of the expected record density (synth.JobLayout.cuts); every rank generates "
                              "and stages only its range (synth.make_reads_blocked)",
                      "records_per_rank": balance, "records_staged_per_rank": staged, "halo_positions": int(lay.halo),
                      "host_generate_ms_per_rank": gen_all, "peak_host_rss_MB_per_rank": rss, "allreduce": allreduce,
                      "stage_ms_per_rank": stage_ms_all, "stage_ms_max": max(stage_ms_all),
                      "source": ("one shared BAM file (written once by rank 0, untimed): every rank stages its genome range of it through the "
                                 "BAI index, decoded on the GPU (pc_add_alignment_bam_span); stage_ms = file -> staged") if bam_path
                                else "records generated by every rank for its own range; stage_ms = host arrays -> staged"},
    }
    if rehearsal:
        res["rehearsal"] = "engine stand-in %s: control flow only, no measurement" % getattr(Engine, "__module__", "?")
    ctx["last_engine_objects"] = (eng, plan, my_reads)
    ctx["last_annotation"] = tx
    return res


def e2e_scope(args, ctx, name, realistic=False):
    """SURVEY 8(d) t_e2e on a bounded sample: a BAM file written once (untimed) is decoded, staged, counted and read
    back.  Two files: the SKELETON records of round 1 / 2 (42 bytes each: name ``r``, no sequence -- a best case by 3x in
    inflate bytes) and, `realistic`, records as an aligner writes them (read name, sequence, qualities, NH / MD tags:
    ~120 bytes per 30-nt read).  Two decoders per file: the native host reader (zlib / libdeflate on the CPUs the box
    grants) and, `*_gpu_decode`, the file image sent to HBM as it is with BGZF inflate and record decode as HIP kernels
    (pc_bam_open)."""
    from plastid_amd.bam import read_bam, read_bam_gpu
    from tests import bam_writer
    Engine, _ = engine_class()
    want = args.e2e_realistic_records if realistic else args.e2e_records
    n = min(int(want), int(synth.CONFIGS[name][4] * args.scale))   # skeleton default: every record of the configuration
    genome, tx, reads, mapping = synth.make_config(name, scale=n / float(synth.CONFIGS[name][4]), tx_scale=args.tx_scale)
    factory = synth.mapping_factory(mapping)
    rows = getattr(factory, "_numlengths", 1)
    p = tx.plan_arrays(rows=rows)
    tmp = tempfile.mkdtemp(prefix="pc_bench_")
    path = os.path.join(tmp, "sample.bam")
    writer = bam_writer.write_bam_realistic if realistic else bam_writer.write_bam_packed
    nbytes = writer(path, reads, threads=min(16, usable_cpus()))
    fsize = os.path.getsize(path)
    eng = Engine(ctx["dev_index"])
    factory._configure(eng)
    out = {}
    out_dtype = np.float64 if mapping[0] == "center" else np.int64

    # host side of the read-back: page-locked, as in the staged scope (what a caller that reads repeatedly hands over)
    out_buf = eng.host_buffer(int(p["out_elems"]), out_dtype) if int(p["out_elems"]) * 8 <= (16 << 30) else np.zeros(int(p["out_elems"]), out_dtype)
    out_buf[:] = 0

    def counted():
        plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], rows)
        try:
            return plan.count(out_dtype, out=out_buf)
        finally:
            plan.close()

    # ---- third form: the file never leaves the GPU between its bytes and the counts (Engine.add_bam = pc_add_alignment_bam:
    # inflate, record decode AND staging as kernels); gated on the counts of the records the file was written from
    key = ("e2e_realistic" if realistic else "e2e") + "_gpu_resident"
    try:
        eng.set_alignments([reads])
        want = counted().copy()
        eng.clear_alignments()
        eng.add_bam(path)                                # page cache + library warm-up
        runs = []
        for _ in range(3):
            t0 = time.perf_counter()
            eng.clear_alignments()
            mapped = eng.add_bam(path)
            t_stage = time.perf_counter() - t0
            got = counted()
            t_all = time.perf_counter() - t0
            runs.append((t_all, t_stage))
            if mapped != reads.n or not np.array_equal(got.view(np.uint64), want.view(np.uint64)):
                raise SystemExit("e2e scope: the counts of the GPU-resident BAM path differ from those of the records it was written from")
            del got
        t_all, t_stage = sorted(runs)[1]                     # the MEDIAN pass is the figure; the best one is kept beside it
        out[key + "_reads_per_s"] = reads.n / t_all
        out[key + "_reads_per_s_best"] = reads.n / min(runs)[0]
        out[key + "_sample"] = ("the same file, decoded AND staged on the GPU (pc_add_alignment_bam: the columns never leave HBM); three whole "
                                "passes (%s s; value = median pass): file -> staged %.3f s + plan, count and read-back into page-locked memory %.3f s; counts gated on those of the "
                                "records the file was written from" % ("/".join("%.3f" % r[0] for r in runs), t_stage, t_all - t_stage))
        del want
    except SystemExit:
        raise
    except Exception as e:   # a scope that fails must not cost the bench line
        out[key + "_error"] = str(e)
    for gpu_decode in (False, True):
        key = ("e2e_realistic" if realistic else "e2e") + ("_gpu_decode" if gpu_decode else "")
        decode = (lambda: read_bam_gpu(path, eng)) if gpu_decode else (lambda: read_bam(path))
        try:
            decode()                                     # page cache + library warm-up
            runs = []
            for _ in range(3):                           # the host cores are shared with other tenants: three whole passes
                t0 = time.perf_counter()
                packed = decode()
                t_decode = time.perf_counter() - t0
                eng.set_alignments([packed])
                t_stage = time.perf_counter() - t0 - t_decode
                got = counted()
                t_all = time.perf_counter() - t0
                runs.append((t_all, t_decode, t_stage))
                ok = all(np.array_equal(getattr(packed, k), getattr(reads, k)) for k in ("tid", "pos", "alen", "flags", "nblk", "blk_start", "blk_len"))
                del packed, got                          # released outside the timed pass
                if not ok:
                    raise SystemExit("e2e scope: the decoded BAM differs from the records it was written from")
        except SystemExit:
            raise
        except Exception as e:   # a scope that fails must not cost the bench line
            out[key + "_error"] = str(e)
            continue
        t_all, t_decode, t_stage = sorted(runs)[1]           # the MEDIAN pass is the figure; the best one is kept beside it
        out[key + "_reads_per_s"] = reads.n / t_all
        out[key + "_reads_per_s_best"] = reads.n / min(runs)[0]
        out[key + "_sample"] = ("%d records of %s written once (untimed) as a BGZF-compressed BAM of %.0f MB (%.0f MB inflated, %.0f bytes per "
                                "record%s); timed: three whole passes (%s s; value = median pass, best beside it): %s decode %.3f s + staging "
                                "%.3f s + plan, count and read-back into page-locked memory %.3f s" %
                                (reads.n, name, fsize / 1e6, nbytes / 1e6, nbytes / float(reads.n),
                                 ": read name, sequence, qualities, NH and MD tags" if realistic else ": name 'r', no sequence",
                                 "/".join("%.3f" % r[0] for r in runs), "GPU (BGZF inflate + record decode as HIP kernels)" if gpu_decode else "native host",
                                 t_decode, t_stage, t_all - t_decode - t_stage))
    del out_buf          # (page-locked memory of the engine: released while the engine is still there)
    eng.close()
    try:
        os.remove(path)
        os.rmdir(tmp)
    except OSError:
        pass
    return out


def sig(x, digits=4):
    """x rounded to `digits` significant digits (the stdout line is kept short; bench_detail.json has full precision)."""
    if x is None or not isinstance(x, (int, float)) or x == 0 or x != x:
        return x
    from math import floor, log10
    return round(x, digits - 1 - int(floor(log10(abs(x)))))


def brief_config(r):
    """The figures of one config that go into the stdout line (`configs`)."""
    if "skipped" in r:
        return {"skipped": True}
    roof, cpu = r["roofline"], r.get("cpu_baseline")
    b = {"ms_per_step": sig(r["ms_per_step"]), "reads_per_s": sig(r["value"]), "kernel_ms": sig(roof["avg_launch_ms"]),
         "frac": sig(roof["frac"], 3), "frac_traffic": sig(roof.get("frac_traffic"), 3),
         "first_count_ms": r["first_count_ms"]["total"], "parity_positions": r["parity_positions"],
         "plan_build_ms": r["plan_build_ms_once_per_annotation"], "staged_reads_per_s": sig(r["scopes"]["staged_reads_per_s"]),
         "host_stage_s": r.get("host_stage_s"), "read_s": r.get("host_read_outputs_s")}
    if r["scopes"].get("staged_with_plan_reads_per_s"):
        b["staged_with_plan"] = sig(r["scopes"]["staged_with_plan_reads_per_s"])
    for k in ("issue_bound_ms", "issue_frac", "row_fill"):
        if k in roof:
            b[k] = sig(roof[k], 3)
    if r.get("size_filter_variant", {}).get("ms_per_step"):
        b["size_filter_ms"] = sig(r["size_filter_variant"]["ms_per_step"])
    if r.get("two_files"):
        b["two_files_ratio"] = sig(r["two_files"]["ratio_to_one_file"], 3)
    if cpu:
        b["cpu_1core"] = sig(cpu["value"])
        if cpu.get("all_cores"):
            b["cpu_all"] = sig(cpu["all_cores"]["value"])
            b["cores"] = cpu["all_cores"]["cores"]
    if "partition" in r:
        pt = r["partition"]
        b["records_per_rank"] = pt["records_per_rank"]
        b["stage_ms_per_rank"] = pt.get("stage_ms_per_rank")
        if pt.get("allreduce"):
            b["allreduce_ms"] = sig(pt["allreduce"]["ms"])
    return b


def single_query_latency(reads, tx, dev_index, n=300):
    """Microseconds per ``ga[segment]`` through the Python mirror (what the reference's scripts do region by region,
    genome_array.py:861-928, bin/psite.py:181-192): median of `n` queries over first exons, for the one-launch path of
    one-window plans and for the general path (PC_NO_SINGLE=1: work lists, two window classes)."""
    import plastid_amd as pa
    out = {}
    ga = pa.BAMGenomeArray(reads, mapping=pa.FivePrimeMapFactory(12), device=dev_index)
    segs = [c[0] for c in tx.chains(limit=n)]
    try:
        for key, env in (("single_query_us", None), ("single_query_general_us", "1")):
            if env is None:
                os.environ.pop("PC_NO_SINGLE", None)
            else:
                os.environ["PC_NO_SINGLE"] = env
            ga._engine.reload_knobs()
            for s_ in segs[:20]:
                ga[s_]
            ts = []
            for s_ in segs:
                t0 = time.perf_counter()
                ga[s_]
                ts.append(time.perf_counter() - t0)
            out[key] = float(np.median(ts)) * 1e6
    finally:
        os.environ.pop("PC_NO_SINGLE", None)
        ga._engine.close()
    return out


def center_issue_bound(eng, plan, kernel_ms):
    """The yardstick of the center kernel that means something (HBM does not bound an ordered float64 replay): its
    VECTOR-ISSUE floor.  A replay step is 3 single-rate vector instructions (2 cycles each on a SIMD-32) and one
    v_fmac_f64 (4 cycles): 10 cycles per step and wave; the launch's steps (counted by the kernel itself in a
    diagnostic launch, pc_center_replay_steps) over 1 024 SIMDs at 2.4 GHz."""
    try:
        steps, waves = eng.center_replay_steps(plan)
    except Exception as e:   # a diagnostic must not cost the bench line
        return {"issue_bound_error": str(e)}
    floor_ms = steps * 10.0 / (1024 * 2.4e9) * 1e3
    fill = {}
    try:   # how full the four lock-step rows of a wave are: entries of all rows / (4 x steps)
        ent, cap = eng.center_row_fill(plan)
        fill = {"row_fill": ent / float(cap) if cap else None, "row_entries": int(ent)}
    except Exception as e:
        fill = {"row_fill_error": str(e)}
    return {"replay_steps": int(steps), "replay_waves": int(waves), "issue_bound_ms": floor_ms, **fill,
            "issue_frac": floor_ms / kernel_ms if kernel_ms > 0 else None,
            "issue_basis": "replay steps x (3 x 2 + 4) cycles / (1024 SIMDs x 2.4 GHz): subrev, subrev_co, cndmask at 2 cycles, v_fmac_f64 at 4"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2", choices=sorted(WORKLOAD_TEXT))
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the read count (testing only)")
    ap.add_argument("--tx-scale", type=float, default=1.0)
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU baseline work (one core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-two-files", action="store_true", help="skip the two-file variant of the headline")
    ap.add_argument("--out-dtype", default="int64", choices=["int64", "float64"])
    ap.add_argument("--other-configs", default="auto",
                    help="comma list of further configs run after the headline (auto: C3,C4,C5 at N=1 with the default "
                         "headline, none otherwise)")
    ap.add_argument("--partition", default="auto", choices=["auto", "genome", "replicas"],
                    help="N > 1: one job cut into genome ranges (strong scaling, default) or independent replicas (weak)")
    ap.add_argument("--parity-chains", type=int, default=200, help="chains of the parity sample when no CPU baseline is timed")
    ap.add_argument("--from-bam", action="store_true",
                    help="one job (also at N = 1): rank 0 writes the config's records as ONE indexed BAM file (untimed) and every rank "
                         "stages its genome range of it -- decoded on the GPU -- instead of generated arrays")
    ap.add_argument("--one-job", action="store_true",
                    help="N = 1 through the one-job path of N > 1 (the blocked generator, the partition code with a single range): what --from-bam is compared with")
    ap.add_argument("--time-budget", type=float, default=1200.0, help="seconds after which no further config is started")
    ap.add_argument("--e2e-records", type=float, default=1e8, help="records of the BAM file of the e2e scope, at most the whole configuration (0: skip)")
    ap.add_argument("--detail-out", default=None, help="where the full result (prose included) goes; default bench_detail.json beside bench.py")
    ap.add_argument("--no-single-query", action="store_true", help="skip the single-query latency loop (profiling passes: it launches the tile kernel hundreds of times)")
    ap.add_argument("--e2e-realistic-records", type=float, default=1e8,
                    help="records of the second e2e sample, written as an aligner writes them (~120 bytes per record; 0: skip)")
    args = ap.parse_args()
    t_start = time.perf_counter()
    # this process creates one engine per config, tens of GB each, one after the other: the idle device blocks of a closed
    # engine are kept for the next one (on some boxes a hipMalloc behind the hipFree of tens of GB takes seconds); the
    # library keeps 4 GB per device by default
    os.environ.setdefault("PC_POOL_RESERVOIR_GB", "96")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: spawn the N ranks (one per GPU) as a child torch.distributed.run and relay its
        # one JSON line and exit code.  This process has not touched a GPU (no torch import, no HIP call) and never does.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(sys.argv[0])] + sys.argv[1:]
        proc = subprocess.run(cmd, stdout=subprocess.PIPE)
        line = None
        for ln in proc.stdout.decode("utf-8", "replace").splitlines():
            if ln.startswith("{") and '"metric"' in ln:
                line = ln
            elif ln.strip():
                print(ln, file=sys.stderr)
        if line:
            print(line)
        raise SystemExit(proc.returncode if (proc.returncode or line) else 1)

    from plastid_amd import multigpu
    rank, local_rank, world = multigpu.env_rank()
    if args.gpus != world:
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))
    partition = "genome" if args.partition == "auto" else args.partition

    import torch
    import torch.distributed as dist
    # one rank per GPU over RCCL.  PC_BENCH_BACKEND=gloo is a rehearsal aid only: it lets the
    # multi-rank flow run on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("PC_BENCH_BACKEND", "nccl")
    rehearsal = engine_class()[1]
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    ctx = {"rank": rank, "world": world, "partition": partition, "backend": backend, "dev_index": dev_index,
           "tdev": "cuda" if backend == "nccl" else "cpu"}

    # The CPU baseline of the headline runs before this process initialises the GPU only in the
    # sense that it never touches it; run_workload does oracle work first, then creates the engine.
    if not rehearsal:
        torch.cuda.set_device(dev_index)
    multigpu.init(backend, device=None if rehearsal else torch.device("cuda", dev_index))  # "nccl" is RCCL on ROCm
    one_job = (world > 1 or args.from_bam or args.one_job) and partition == "genome"
    run_config = run_partitioned if one_job else run_workload

    head = run_config(args.config, args, ctx, headline=True)
    eng, plan, _reads = ctx.pop("last_engine_objects")
    if rank == 0 and not rehearsal and head["roofline"]["kernel"] == "k_center":
        head["roofline"].update(center_issue_bound(eng, plan, head["roofline"]["avg_launch_ms"]))
    stream_peak = None
    if rank == 0 and not rehearsal:
        try:
            rd, wr = eng.stream_probe(1 << 30, 5)
            stream_peak = {"read_GBps": rd, "write_GBps": wr,
                           "what": "pc_stream_probe: 16-byte contiguous loads / 8-byte contiguous stores per lane over 1 GiB, "
                                   "measured in this process"}
        except Exception as e:  # a probe failure must not cost the bench line
            stream_peak = {"error": str(e)}
    plan.close()
    eng.close()
    single_query = None
    if rank == 0 and world == 1 and not rehearsal and not args.no_single_query and not args.from_bam and not args.one_job:
        try:   # (the headline's own records, staged once more by the mirror's own engine)
            single_query = single_query_latency(_reads, ctx["last_annotation"], dev_index)
        except Exception as e:   # a diagnostic must not cost the bench line
            single_query = {"single_query_error": str(e)}
    del eng, plan, _reads
    gc.collect()

    others = {}
    want = args.other_configs
    if want == "auto":
        # N = 1: the other single-GPU configs; N > 1: the two configs BASELINE.json labels 8 x MI355X, as one job each
        want = ("C3,C4,C5" if world == 1 else ("C4,C5" if one_job else "none")) if (args.config == "C2" and args.scale == 1.0 and not args.from_bam and not args.one_job) else "none"
    names = [c for c in want.split(",") if c and c != "none" and c != args.config]
    for c in names:
        if time.perf_counter() - t_start > args.time_budget:
            others[c] = {"skipped": "time budget of %.0f s used up before this config" % args.time_budget}
            continue
        r = run_config(c, args, ctx, headline=False)
        e2, p2, _r2 = ctx.pop("last_engine_objects")
        if rank == 0 and not rehearsal and r["roofline"]["kernel"] == "k_center":
            r["roofline"].update(center_issue_bound(e2, p2, r["roofline"]["avg_launch_ms"]))
        p2.close()
        e2.close()
        del e2, p2, _r2
        gc.collect()
        others[c] = r
    e2e = None
    if rank == 0 and world == 1 and not rehearsal and not args.from_bam and not args.one_job and time.perf_counter() - t_start <= args.time_budget:
        e2e = {}
        for realistic, want in ((False, args.e2e_records), (True, args.e2e_realistic_records)):
            if want <= 0:
                continue
            try:
                e2e.update(e2e_scope(args, ctx, args.config, realistic=realistic))
            except SystemExit:
                raise
            except Exception as e:
                e2e["e2e_realistic_error" if realistic else "e2e_error"] = str(e)

    if rank == 0:
        roof = dict(head["roofline"])
        if stream_peak and "read_GBps" in stream_peak:
            roof["stream_peak_measured"] = stream_peak
            roof["frac_of_measured_stream"] = roof["achieved"] / stream_peak["read_GBps"]
            if roof.get("traffic"):
                roof["frac_traffic_of_measured_stream"] = roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9 / stream_peak["read_GBps"]
        scopes = dict(head["scopes"])
        if e2e:
            scopes.update(e2e)
        if single_query:
            scopes.update(single_query)
        # ---- everything, prose included, goes to a side file; the ONE stdout line stays small enough for a
        # tail-capturing driver (<= 4 KB) and carries every config's figures
        detail = {"headline": dict(head, roofline=roof, scopes=scopes), "other_configs": others,
                  "kernel_source_sha16": kernel_source_hash(), "runtime_warmup_s": ctx.get("runtime_warmup_s"),
                  "bench_wall_s": round(time.perf_counter() - t_start, 1), "argv": sys.argv[1:]}
        detail_path = args.detail_out or os.path.join(ROOT, "bench_detail.json")
        try:
            with open(detail_path, "w") as f:
                json.dump(detail, f, indent=1, default=str)
        except OSError as e:
            print("bench: could not write %s: %s" % (detail_path, e), file=sys.stderr)
            detail_path = None
        config = {"workload": head["workload"], "records": head["records_total"], "chains": head["chains"],
                  "segments": head["segments"], "output_positions": head["output_positions_per_gpu"], "tiles": head["tiles"],
                  "mapping": head["mapping"], "parity_positions": head["parity_positions"],
                  "plan_build_ms": head["plan_build_ms_once_per_annotation"], "host_stage_s": head["host_stage_s"],
                  "read_s": head.get("host_read_outputs_s"),
                  "kernel_source_sha16": kernel_source_hash(), "bench_wall_s": detail["bench_wall_s"],
                  "detail": os.path.basename(detail_path) if detail_path else None}
        if head.get("two_files"):
            config["two_files_ratio"] = sig(head["two_files"]["ratio_to_one_file"], 3)
        if head.get("size_filter_variant", {}).get("ms_per_step"):
            config["size_filter_ms_per_step"] = sig(head["size_filter_variant"]["ms_per_step"])
        if "partition" in head:
            pt = head["partition"]
            config["partition"] = {"records_per_rank": pt["records_per_rank"], "halo": pt["halo_positions"],
                                   "stage_ms_per_rank": pt.get("stage_ms_per_rank"), "from_bam": bool(args.from_bam),
                                   "allreduce_ms": sig(pt["allreduce"]["ms"]) if pt.get("allreduce") else None}
        if "rehearsal" in head:
            config["rehearsal"] = head["rehearsal"]
        fc = head["first_count_ms"]["total"]
        cpu = head["cpu_baseline"]
        result = {
            "metric": "mapped_reads_per_sec",
            "value": head["value"],
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak" if (world == 1 or partition == "replicas") else "strong",
            "vs_baseline": None,
            "dtype": head["dtype"],
            "data": "synthetic" if not one_job else "synthetic (every rank generates its own genome range of the one job)",
            # the first count of a plan also builds its work lists -- the only count a one-shot run does
            "first_count_ms": fc,
            "value_first_count": sig(head["records_total"] / (fc * 1e-3)) if (fc and head["value"] is not None) else None,
            "config": config,
            "roofline": {k: (sig(v) if isinstance(v, float) else v) for k, v in roof.items()
                         if k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_traffic", "avg_launch_ms",
                                  "frac_of_measured_stream", "issue_bound_ms", "issue_frac", "replay_steps", "row_fill")},
            "cpu_baseline": None if not cpu else {
                "value": sig(cpu["value"]), "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
                "sample": "oracle (C port of the reference algorithm) on %d of %d chains, all %d records; see %s" %
                          (cpu.get("chains_sampled", 0), head["chains"], head["records_total"], config["detail"]),
                "all_cores": None if not cpu.get("all_cores") else {"value": sig(cpu["all_cores"]["value"]), "cores": cpu["all_cores"]["cores"]},
                "cpu_model": cpu.get("cpu_model"), "usable_cores": cpu.get("usable_cores")},
            "scopes": {k: sig(v) for k, v in scopes.items() if isinstance(v, (int, float)) and not k.endswith("_best")},
            "configs": {c: brief_config(r) for c, r in others.items()},
        }
        if roof.get("stream_peak_measured"):
            result["roofline"]["stream_read_GBps"] = sig(roof["stream_peak_measured"]["read_GBps"])
        line = json.dumps(result, separators=(",", ":"))
        if len(line) > 4096:   # never expected; keep the line parseable AND short whatever happens
            result["config"] = {k: config[k] for k in ("workload", "records", "chains", "parity_positions", "detail")}
            line = json.dumps(result, separators=(",", ":"))
        print(line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
