"""Experiment: cProfile of the Python mirror on single-segment queries."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plastid_amd as pa
from plastid_amd import synth
genome, tx, reads, _ = synth.make_config("C2", scale=0.1, tx_scale=0.05)
ga = pa.BAMGenomeArray(reads, mapping=pa.FivePrimeMapFactory(12))
chains = tx.chains(limit=500)
segs = [c[0] for c in chains]
ga[segs[0]]
which = sys.argv[1] if len(sys.argv) > 1 else "seg"
pr = cProfile.Profile()
pr.enable()
for _ in range(4):
    if which == "seg":
        for s in segs:
            ga[s]
    else:
        for c in chains:
            c.get_counts(ga)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
