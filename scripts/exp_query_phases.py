"""Experiment: where a single-segment query spends its time (engine phases vs the Python mirror)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plastid_amd as pa
from plastid_amd import synth
genome, tx, reads, _ = synth.make_config("C2", scale=0.1, tx_scale=0.05)
ga = pa.BAMGenomeArray(reads, mapping=pa.FivePrimeMapFactory(12))
chains = tx.chains(limit=500)
segs = [c[0] for c in chains]
ga[segs[0]]
eng = ga._engine
N = len(segs)
T = dict(create=0.0, launch=0.0, read=0.0, close=0.0)
for s in segs:
    n = s.end - s.start
    t0 = time.perf_counter()
    p = eng.plan([ga._chrom_index[s.chrom]], [s.start], [s.end], [s.c_strand], [0], [1], [n], n)
    t1 = time.perf_counter()
    p.launch(np.int64)
    t2 = time.perf_counter()
    out = p.read()
    t3 = time.perf_counter()
    p.close()
    t4 = time.perf_counter()
    T["create"] += t1 - t0; T["launch"] += t2 - t1; T["read"] += t3 - t2; T["close"] += t4 - t3
print({k: round(v / N * 1e6, 1) for k, v in T.items()}, "us per query")
t0 = time.perf_counter()
for s in segs:
    ga[s]
print("ga[segment] %.1f us" % ((time.perf_counter() - t0) / N * 1e6))
