#!/usr/bin/env python
"""Headline benchmark: mapped reads/s of the per-position counting hot path.

    python bench.py --gpus 1 --steps K --warmup W            (one process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one pass of the hot path over one batch: every alignment record of
the synthetic BAM (already packed and resident in HBM) is counted under the
mapping rule into the per-position vectors of ALL transcripts of the annotation
(int64, every chain laid out 5'->3') -- what ``for chain in transcripts:
chain.get_counts(ga)`` does in the reference.  Default workload = BASELINE.json
``configs[1]`` (C2): 100 M single-end reads, FivePrimeMapFactory(offset=12),
20 k yeast-scale transcripts, one MI355X.

At N > 1 every rank owns an independent shard of the same shape (its own
records and the same annotation: weak scaling, no data-path collective); the
only collective is an RCCL all-reduce of the summary totals.

Before anything is timed the GPU output is compared bit-for-bit with the oracle
(``oracle/``) on a seeded sample of chains; a mismatch aborts the run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from plastid_amd import synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOAD_TEXT = {
    "C1": "C1: 1 M synthetic reads, FivePrimeMapFactory(offset=0), 200 yeast-scale SegmentChains",
    "C2": "C2: 100 M synthetic single-end reads, FivePrimeMapFactory(offset=12), 20 k yeast-scale transcripts",
    "C3": "C3: 100 M synthetic reads, CenterMapFactory(nibble=0), 20 k yeast-scale transcripts",
    "C4": "C4: 500 M synthetic reads, VariableFivePrimeMapFactory, 60 k human-scale transcripts",
    "C5": "C5: 1 B paired-end records, StratifiedVariableFivePrimeMapFactory(25..35), 60 k human-scale transcripts",
}


def oracle_spec(oracle, mapping):
    kind = mapping[0]
    if kind in ("fiveprime", "threeprime", "center"):
        return oracle.mapping_spec(kind, mapping[1])
    if kind == "variable":
        return oracle.mapping_spec(kind, 0, mapping[1])
    return oracle.mapping_spec(kind, 0, mapping[1], mapping[2], mapping[3])


def scatter_expected(arrays, p, sel, rows, dtype, out_elems):
    """Oracle per-segment arrays -> the plan's output layout (only segments `sel`)."""
    out = np.zeros(out_elems, dtype)
    for arr, s in zip(arrays, sel):
        n = arr.shape[-1]
        idx = p["out_off"][s] + p["out_step"][s].astype(np.int64) * np.arange(n)
        a2 = arr.reshape(rows, n)
        for r in range(rows):
            out[idx + r * p["row_stride"][s]] = a2[r]
    return out


def cpu_baseline(oracle, aln, spec, p, tx, n_records, budget_s, rng):
    """Time the oracle (single-threaded C port of the reference algorithm) on a
    bounded, seeded sample of chains; report whole-job-equivalent reads/s."""
    order = rng.permutation(tx.n)
    seg_of_chain = [np.arange(tx.ex_off[c], tx.ex_off[c + 1]) for c in order]
    # pilot to size the sample
    done, t_used, results = 0, 0.0, []
    batch = 50
    while done < tx.n and t_used < budget_s:
        chains = order[done:done + batch]
        sel = np.concatenate([seg_of_chain[k] for k in range(done, done + len(chains))])
        t0 = time.perf_counter()
        arrays, _ = oracle.count_segments(aln, spec, p["tid"][sel], p["start"][sel], p["end"][sel], p["strand"][sel])
        t_used += time.perf_counter() - t0
        results.append((sel, arrays))
        done += len(chains)
        if t_used > 0:
            rate = done / t_used
            batch = int(max(50, min(tx.n - done, rate * max(budget_s - t_used, 0.5) * 0.5)))
    frac = done / float(tx.n)
    value = n_records * frac / t_used
    sel_all = np.concatenate([s for s, _ in results])
    arrays_all = [a for _, arrs in results for a in arrs]
    return {"value": value, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": "oracle/plastid_oracle.c (C port of the reference algorithm, 1 thread) over %d of %d "
                      "transcripts (seeded sample) against all %d records: %.1f s; value = records x "
                      "sampled fraction / time" % (done, tx.n, n_records, t_used)}, sel_all, arrays_all, order[:done]


_MP = {}


def _mp_worker(sel):
    t0 = time.perf_counter()
    _MP["oracle"].count_segments(_MP["aln"], _MP["spec"], _MP["p"]["tid"][sel], _MP["p"]["start"][sel],
                                 _MP["p"]["end"][sel], _MP["p"]["strand"][sel])
    return time.perf_counter() - t0


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_all_cores(oracle, aln, spec, p, tx, n_records, chains_done):
    """The same oracle on every host core: one process per core over disjoint chain shards
    (the reference itself is single-threaded; this is the most favourable honest scaling)."""
    import multiprocessing as mp
    # every worker re-derives the oracle's per-record arrays (16 B/record), so the process count
    # is bounded by memory, not only by cores: at most 16 workers
    cores = min(os.cpu_count() or 1, 16)
    if cores < 2 or len(chains_done) < cores:
        return None
    _MP.update(oracle=oracle, aln=aln, spec=spec, p=p)
    shards = np.array_split(np.asarray(chains_done), cores)
    sels = [np.concatenate([np.arange(tx.ex_off[c], tx.ex_off[c + 1]) for c in sh]) for sh in shards]
    ctx = mp.get_context("fork")  # before the GPU is initialised; children only run the C oracle
    with ctx.Pool(cores) as pool:
        t0 = time.perf_counter()
        pool.map(_mp_worker, sels)
        wall = time.perf_counter() - t0
    frac = len(chains_done) / float(tx.n)
    return {"value": n_records * frac / wall, "unit": "reads/s", "cores": cores,
            "sample": "%d transcripts in %d forked processes: %.2f s wall" % (len(chains_done), cores, wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2", choices=sorted(WORKLOAD_TEXT))
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the read count (testing only)")
    ap.add_argument("--tx-scale", type=float, default=1.0)
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU baseline work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--out-dtype", default="int64", choices=["int64", "float64"])
    args = ap.parse_args()

    from plastid_amd import multigpu
    rank, local_rank, world = multigpu.env_rank()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))

    # ---------------------------------------------------------------- inputs (host)
    t0 = time.perf_counter()
    genome, tx, reads, mapping = synth.make_config(args.config, scale=args.scale, tx_scale=args.tx_scale,
                                                   seed_shift=rank)
    gen_s = time.perf_counter() - t0
    center = mapping[0] == "center"
    out_dtype = np.float64 if (center or args.out_dtype == "float64") else np.int64
    factory = synth.mapping_factory(mapping)
    rows = getattr(factory, "_numlengths", 1)
    p = tx.plan_arrays(rows=rows)

    # ---------------------------------------------------------------- CPU baseline (before the GPU is touched)
    cpu = None
    check_sel = check_arrays = None
    from oracle import oracle
    from plastid_amd.packing import concat_file_major
    aln = concat_file_major([reads])
    spec = oracle_spec(oracle, mapping)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, check_sel, check_arrays, chains_done = cpu_baseline(oracle, aln, spec, p, tx, reads.n, args.cpu_budget,
                                                                 np.random.default_rng(7))
        cpu["all_cores"] = cpu_baseline_all_cores(oracle, aln, spec, p, tx, reads.n, chains_done)
        cpu["cpu_model"] = cpu_model()
        cpu["host_cores"] = os.cpu_count()
    else:
        # parity gate only: a small seeded sample of chains
        order = np.random.default_rng(7 + rank).permutation(tx.n)[:min(tx.n, 100)]
        check_sel = np.concatenate([np.arange(tx.ex_off[c], tx.ex_off[c + 1]) for c in order])
        check_arrays, _ = oracle.count_segments(aln, spec, p["tid"][check_sel], p["start"][check_sel],
                                                p["end"][check_sel], p["strand"][check_sel])
    del aln

    # ---------------------------------------------------------------- GPU
    import torch
    import torch.distributed as dist
    # one rank per GPU over RCCL.  PC_BENCH_BACKEND=gloo is a rehearsal aid only: it lets the
    # multi-rank flow run on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("PC_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    tdev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(dev_index)
    multigpu.init(backend, device=torch.device("cuda", dev_index))  # "nccl" is RCCL on ROCm
    from plastid_amd.engine import Engine
    eng = Engine(dev_index)
    t0 = time.perf_counter()
    eng.set_alignments([reads])
    stage_s = time.perf_counter() - t0
    factory._configure(eng)
    t0 = time.perf_counter()
    plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"],
                    p["out_elems"], rows)
    plan_s = time.perf_counter() - t0

    barrier = multigpu.barrier

    for _ in range(max(args.warmup, 1)):
        plan.launch(out_dtype)
    eng.sync()

    # parity gate: bit-exact vs the oracle on the sampled chains
    t0 = time.perf_counter()
    got = plan.read()
    read_s = time.perf_counter() - t0   # D2H of every output position (t_staged of SURVEY 8d = stage + step + this)
    exp = scatter_expected(check_arrays, p, check_sel, rows, out_dtype, p["out_elems"])
    touched = np.zeros(p["out_elems"], bool)
    for s in check_sel:
        n = p["end"][s] - p["start"][s]
        idx = p["out_off"][s] + p["out_step"][s].astype(np.int64) * np.arange(n)
        for r in range(rows):
            touched[idx + r * p["row_stride"][s]] = True
    if not np.array_equal(got[touched], exp[touched]):
        raise SystemExit("PARITY FAILURE: HIP counts differ from the oracle on the sampled chains")
    n_checked = int(touched.sum())
    total_counts = plan.total()
    del got, exp, touched

    # ---------------------------------------------------------------- timed region
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.launch(out_dtype)
    eng.sync()
    torch.cuda.synchronize()
    barrier()
    elapsed = multigpu.max_over_ranks(time.perf_counter() - t0, device=tdev)

    # summary totals: the only collective (RCCL all-reduce over xGMI)
    n_records_all, counts_all, positions_all = multigpu.allreduce_int_totals(
        [int(reads.n), int(total_counts) if not center else 0, int(p["out_elems"])], device=tdev)
    if center:
        counts_all = multigpu.reduce_float_totals_ordered([float(total_counts)], device=tdev)[0]

    # ---------------------------------------------------------------- per-kernel timing (HIP events on the engine's stream)
    # The timed region above runs the product default (no events: each hipEventRecord costs a few
    # microseconds of stream time).  The same launches are repeated here with the engine's phase
    # events switched on; the dominant kernel's average duration feeds the roofline object.
    phases = {"total": 0.0, "worklist": 0.0, "hist": 0.0, "long": 0.0, "gather": 0.0, "zero": 0.0}
    m = max(3, min(args.steps, 20))
    eng.set_profiling(2)
    for _ in range(m):
        plan.launch(out_dtype)
        eng.sync()
        for k, v in eng.last_timing().items():
            phases[k] += v / m
    alg_bytes_step = eng.last_algorithmic_bytes()

    if rank == 0:
        n_extra_runs = int(len(reads.blk_start))
        if center:
            # k_center streams the candidate records and writes the float64 island histogram
            kern_alg_bytes = reads.n * 8 + n_extra_runs * 8 + plan.positions * 8
        else:
            # dominant kernel = k_hist_point (fused): streams every packed record once (8 B) + the
            # runs of gapped records (8 B each) + the segment table (24 B each) and writes every
            # output position once (8 B) -- exactly SURVEY section 8(d)'s B_alg for the step
            kern_alg_bytes = reads.n * 8 + n_extra_runs * 8 + tx.n_segments * 24 + int(p["out_elems"]) * 8
        kern_ms = phases["hist"]
        achieved = kern_alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("config") == args.config and int(tj.get("n_records", -1)) == reads.n:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        ms_per_step = elapsed / args.steps * 1e3
        value = n_records_all * args.steps / elapsed
        result = {
            "metric": "mapped_reads_per_sec",
            "value": value,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "float64" if out_dtype == np.float64 else "int64",
            "data": "synthetic",
            "config": {
                "workload": WORKLOAD_TEXT[args.config] + ("" if args.scale == 1.0 else " [scaled x%g]" % args.scale),
                "records_per_gpu": int(reads.n), "records_total": n_records_all,
                "chains": int(tx.n), "segments": int(tx.n_segments), "output_positions_per_gpu": int(p["out_elems"]),
                "island_positions": int(plan.positions), "tiles": int(plan.tiles), "rows": rows,
                "mapping": [str(x) for x in mapping], "read_seed": synth.CONFIGS[args.config][5],
                "transcript_seed": synth.CONFIGS[args.config][3],
                "positions_per_sec": positions_all * args.steps / elapsed,
                "parity": "bit-exact vs oracle on %d output positions (seeded sample of chains)" % n_checked,
                "sum_of_counts_all_ranks": counts_all,
                "host_generate_s": round(gen_s, 2), "host_stage_s": round(stage_s, 2), "host_read_outputs_s": round(read_s, 3),
                "plan_build_ms_once_per_annotation": round(plan_s * 1e3, 2),
                "algorithmic_bytes_per_step": int(alg_bytes_step),
                "staged_stream_bytes_per_record": 4,  # what the tile kernel reads; the algorithmic record is 8 B
                "step_GBps_algorithmic": alg_bytes_step / (ms_per_step * 1e-3) / 1e9,
                "kernel_ms": {k: round(v, 4) for k, v in phases.items()},
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_center" if center else "k_hist_point",
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": int(kern_alg_bytes), "avg_launch_ms": kern_ms,
            },
            "cpu_baseline": cpu,
        }
        print(json.dumps(result))
    plan.close()
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
