"""Experiment: compressed BAM -> staged on the GPU without the records visiting the host (Engine.add_bam), N records as an
aligner writes them; prints the decoder's and the staging laps (PC_BAM_TIMING=1 PC_STAGE_TIMING=1) of four passes on one
engine (FRESH_ENGINE=1: a new engine per pass).
usage: PC_BAM_TIMING=1 PC_STAGE_TIMING=1 python scripts/exp_bam_resident.py 2e7"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plastid_amd import synth  # noqa: E402
from plastid_amd.engine import Engine  # noqa: E402
from tests import bam_writer  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
genome, tx, reads, mapping = synth.make_config("C2", scale=n / 1e8)
tmp = tempfile.mkdtemp(prefix="pc_bamres_")
path = os.path.join(tmp, "s.bam")
nbytes = bam_writer.write_bam_realistic(path, reads, threads=16)
print("file: %d records, %.1f MB compressed, %.1f MB inflated" % (reads.n, os.path.getsize(path) / 1e6, nbytes / 1e6), flush=True)
fresh = os.environ.get("FRESH_ENGINE") == "1"     # a new engine per pass (its page-locked upload ring is then made inside the lap)
eng = None
for rep in range(4):
    if eng is None or fresh:
        eng = Engine(0)
    sys.stderr.write("== pass %d\n" % rep); sys.stderr.flush()
    t0 = time.perf_counter()
    eng.clear_alignments()
    t1 = time.perf_counter()
    eng.add_bam(path)
    eng.sync()
    t2 = time.perf_counter()
    print("pass %d: clear %.1f ms, file -> staged %.1f ms = %.3g reads/s" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, reads.n / (t2 - t0)), flush=True)
    if fresh:
        eng.close()
os.remove(path)
os.rmdir(tmp)
