#!/usr/bin/env python
"""Time the REFERENCE's own Cython map factories against the oracle on a downsized C2 workload (four
mapping rules) and on the C1 shape (build container only; needs the scratch reference of
build_scratch_reference.sh).

    PYTHONPATH=/tmp/oracle:/tmp/oracle/stubs:. python tests/golden/time_reference.py

Both run on 1 core over the same reads and the same transcripts.  The reference side is
charged only for map_fn (stub reads with pre-built ``positions`` lists, reads pre-bucketed per
segment by bisect) -- pysam fetch/decode, which real plastid also pays, is excluded.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import pysam  # scratch stub
from plastid.genomics.map_factories import (CenterMapFactory, FivePrimeMapFactory, StratifiedVariableFivePrimeMapFactory,
                                            VariableFivePrimeMapFactory)
from plastid.genomics.roitools import GenomicSegment

from oracle import oracle
from plastid_amd import synth
from plastid_amd.packing import concat_file_major

VDICT = dict(synth.VARIABLE_OFFSETS)
out = {}
for shape, cfg, scale, tx_scale, rules in (
        ("C2 shape: 1 M reads x 2 000 transcripts", "C2", 0.01, 0.1,
         (("fiveprime12", lambda: FivePrimeMapFactory(12), oracle.mapping_spec("fiveprime", 12)),
          ("center0", lambda: CenterMapFactory(0), oracle.mapping_spec("center", 0)),
          ("variable", lambda: VariableFivePrimeMapFactory(VDICT), oracle.mapping_spec("variable", 0, VDICT)),
          ("stratified_25_35", lambda: StratifiedVariableFivePrimeMapFactory(VDICT, 25, 35),
           oracle.mapping_spec("stratified", 0, VDICT, 25, 35)))),
        ("C1 shape: 1 M reads x 200 chains, offset 0 (BASELINE configs[0])", "C1", 1.0, 1.0,
         (("c1_fiveprime0", lambda: FivePrimeMapFactory(0), oracle.mapping_spec("fiveprime", 0)),))):
    genome, tx, reads, _ = synth.make_config(cfg, scale=scale, tx_scale=tx_scale)
    p = tx.plan_arrays(rows=1)
    aln = concat_file_major([reads])
    ref_end = reads.ref_end()
    stub = [pysam.AlignedSegment(reads.read(i).positions, bool(reads.flags[i] & 1)) for i in range(reads.n)]
    bounds = reads.tid_bounds()
    max_span = int((ref_end - reads.pos).max())
    # reference: per segment, fetch-equivalent bucketing is done outside the timer
    buckets = []
    for s_ in range(len(p["tid"])):
        t, a, b, st = int(p["tid"][s_]), int(p["start"][s_]), int(p["end"][s_]), int(p["strand"][s_])
        lo = bounds[t] + np.searchsorted(reads.pos[bounds[t]:bounds[t + 1]], a - max_span, side="right")
        hi = bounds[t] + np.searchsorted(reads.pos[bounds[t]:bounds[t + 1]], b, side="left")
        idx = np.arange(lo, hi)
        idx = idx[(ref_end[lo:hi] > a) & (((reads.flags[lo:hi] & 1) == 1) == (st == 2))]
        buckets.append(([stub[i] for i in idx], GenomicSegment(tx.references[t], a, b, "+" if st == 1 else "-")))
    touched = sum(len(b[0]) for b in buckets)
    for name, make, spec in rules:
        fac = make()
        t0 = time.perf_counter()
        ref_arrays = [fac(rs, seg)[1] for rs, seg in buckets]
        t_ref = time.perf_counter() - t0
        t0 = time.perf_counter()
        arrays, _ = oracle.count_segments(aln, spec, p["tid"], p["start"], p["end"], p["strand"])
        t_or = time.perf_counter() - t0
        assert all(np.array_equal(a, b) for a, b in zip(ref_arrays, arrays)), name
        out[name] = {"shape": shape, "reads": int(reads.n), "segments": int(len(buckets)), "read_visits": int(touched),
                     "reference_cython_s": round(t_ref, 4), "oracle_c_s": round(t_or, 4),
                     "reference_reads_per_s_whole_job": reads.n / t_ref, "oracle_reads_per_s_whole_job": reads.n / t_or,
                     "oracle_over_reference": t_ref / t_or}
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(HERE)), "profiles", "reference_vs_oracle_cpu.json"), "w"), indent=1)
