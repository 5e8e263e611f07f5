"""Experiment: plan creation time for a large annotation (C4: 60 k transcripts, 479 k exons)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plastid_amd import synth
from plastid_amd.engine import Engine
genome, tx, reads, mapping = synth.make_config("C4", scale=0.01, tx_scale=1.0)
eng = Engine(0); eng.set_alignments([reads]); synth.mapping_factory(mapping)._configure(eng)
p = tx.plan_arrays(rows=1)
time.sleep(0.5)
for _ in range(4):
    t0 = time.perf_counter()
    plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
    dt = time.perf_counter() - t0
    out = plan.count(np.int64)
    print("plan build %.1f ms (%d segments, %d tiles), counted %d" % (dt * 1e3, len(p["tid"]), plan.tiles, int(out.sum())), flush=True)
    plan.close()
