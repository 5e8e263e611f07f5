#!/usr/bin/env python
"""Time the REFERENCE's own Cython map factories against the oracle on a downsized C2
workload (build container only; needs the scratch reference of build_scratch_reference.sh).

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
from plastid.genomics.map_factories import CenterMapFactory, FivePrimeMapFactory
from plastid.genomics.roitools import GenomicSegment

from oracle import oracle
from plastid_amd import synth
from plastid_amd.packing import concat_file_major

genome, tx, reads, _ = synth.make_config("C2", scale=0.01, tx_scale=0.1)   # 1 M reads, 2 000 transcripts
p = tx.plan_arrays(rows=1)
aln = concat_file_major([reads])
ref_end = reads.ref_end()
stub = [pysam.AlignedSegment(reads.read(i).positions, bool(reads.flags[i] & 1)) for i in range(reads.n)]
bounds = reads.tid_bounds()
max_span = int((ref_end - reads.pos).max())
out = {}
for name, fac, spec in (("fiveprime12", FivePrimeMapFactory(12), oracle.mapping_spec("fiveprime", 12)),
                        ("center0", CenterMapFactory(0), oracle.mapping_spec("center", 0))):
    # reference: per segment, fetch-equivalent bucketing is done outside the timer
    buckets = []
    for s in range(len(p["tid"])):
        t, a, b, st = int(p["tid"][s]), int(p["start"][s]), int(p["end"][s]), int(p["strand"][s])
        lo = bounds[t] + np.searchsorted(reads.pos[bounds[t]:bounds[t + 1]], a - max_span, side="right")
        hi = bounds[t] + np.searchsorted(reads.pos[bounds[t]:bounds[t + 1]], b, side="left")
        idx = np.arange(lo, hi)
        idx = idx[(ref_end[lo:hi] > a) & (((reads.flags[lo:hi] & 1) == 1) == (st == 2))]
        buckets.append(([stub[i] for i in idx], GenomicSegment(tx.references[t], a, b, "+" if st == 1 else "-")))
    t0 = time.perf_counter()
    ref_arrays = [fac(rs, seg)[1] for rs, seg in buckets]
    t_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    arrays, _ = oracle.count_segments(aln, spec, p["tid"], p["start"], p["end"], p["strand"])
    t_or = time.perf_counter() - t0
    assert all(np.array_equal(a, b) for a, b in zip(ref_arrays, arrays)), name
    touched = sum(len(b[0]) for b in buckets)
    out[name] = {"reads": int(reads.n), "segments": int(len(buckets)), "read_visits": int(touched),
                 "reference_cython_s": round(t_ref, 4), "oracle_c_s": round(t_or, 4),
                 "reference_reads_per_s_whole_job": reads.n / t_ref, "oracle_reads_per_s_whole_job": reads.n / t_or,
                 "oracle_over_reference": t_ref / t_or}
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(HERE)), "profiles", "reference_vs_oracle_cpu.json"), "w"), indent=1)
