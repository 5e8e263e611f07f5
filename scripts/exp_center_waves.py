"""Experiment: how long every wave of the center kernel runs (PC_CENTER_DEBUG) on C3 -- is the launch the
tail of one hot chunk, or throughput?  (how the scripts/ubench/notes experiments were read)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from plastid_amd import synth
from plastid_amd.engine import Engine
genome, tx, reads, mapping = synth.make_config("C3", scale=float(os.environ.get("SCALE", "1.0")))
factory = synth.mapping_factory(mapping)
eng = Engine(0)
eng.add_alignment_file(reads)
factory._configure(eng)
p = tx.plan_arrays(rows=1)
plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
plan.launch(np.float64); eng.sync()
import time
eng.set_profiling(2)
for _ in range(3):
    plan.launch(np.float64); eng.sync()
    print("phases ms:", {k: round(v, 4) for k, v in eng.last_timing().items()})
eng.set_profiling(0)
os.environ["PC_CENTER_DEBUG"] = "1"
eng.reload_knobs()   # the knobs are read at pc_create / pc_reload_knobs
plan.launch(np.float64); eng.sync()
