"""Experiment: latency of single-segment / single-chain queries through the Python mirror."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plastid_amd as pa
from plastid_amd import synth
genome, tx, reads, _ = synth.make_config("C2", scale=0.1, tx_scale=0.05)
ga = pa.BAMGenomeArray(reads, mapping=pa.FivePrimeMapFactory(12))
chains = tx.chains(limit=500)
segs = [c[0] for c in chains]
for name, fn, items in (("ga[segment]", lambda s: ga[s], segs), ("chain.get_counts(ga)", lambda c: c.get_counts(ga), chains)):
    fn(items[0])
    t0 = time.perf_counter()
    for it in items:
        fn(it)
    dt = (time.perf_counter() - t0) / len(items)
    print("%-24s %.1f us per query" % (name, dt * 1e6))
t0 = time.perf_counter(); out = ga.get_counts_batch(chains); dt = time.perf_counter() - t0
print("get_counts_batch(500)    %.1f us per chain" % (dt / len(chains) * 1e6))
