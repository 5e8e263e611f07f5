"""Experiment: where pc_add_alignment_file spends its time (PC_STAGE_TIMING=1) on the C2 reads."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PC_STAGE_TIMING"] = "1"
from plastid_amd import synth
from plastid_amd.engine import Engine
cfg = os.environ.get("CONFIG", "C2")
genome, tx, reads, mapping = synth.make_config(cfg, scale=float(os.environ.get("SCALE", "1.0")), tx_scale=float(os.environ.get("TX", "1.0")))
for fresh in (True, False):
    eng = Engine(0)
    for _ in range(3):
        t0 = time.perf_counter()
        eng.clear_alignments()
        t1 = time.perf_counter()
        eng.add_alignment_file(reads)
        t2 = time.perf_counter()
        print("clear %.3f s, add_alignment_file %.3f s for %d records" % (t1 - t0, t2 - t1, reads.n), flush=True)
    eng.close()
