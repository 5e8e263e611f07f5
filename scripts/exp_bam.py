"""Experiment: phases of the native BAM reader (PB_TIMING=1) on a synthetic coordinate-sorted BAM."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PB_TIMING"] = "1"
from plastid_amd import bam, synth
from tests import bam_writer as bw
genome, tx, reads, _ = synth.make_config("C2", scale=float(os.environ.get("SCALE", "0.03")), tx_scale=0.05)
path = "/tmp/exp_bam_%d.bam" % os.getpid()
bw.write_bam(path, reads.references, reads.lengths, bw.packed_to_records(reads))
print("BAM: %d records, %.1f MB" % (reads.n, os.path.getsize(path) / 1e6), flush=True)
for th in (0, 0, 8, 1):
    t0 = time.perf_counter()
    p = bam.read_bam(path, threads=th)
    dt = time.perf_counter() - t0
    print("threads=%d read_bam %.3f s = %.1f M records/s" % (th, dt, p.n / dt / 1e6), flush=True)
assert np.array_equal(p.pos, reads.pos) and np.array_equal(p.alen, reads.alen) and np.array_equal(p.flags & 1, reads.flags & 1)
os.unlink(path)
