"""Experiment: plan creation time right after staging vs later (background free of the staging scratch,
first page-locked allocation)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plastid_amd import synth
from plastid_amd.engine import Engine
genome, tx, reads, mapping = synth.make_config("C2")
p = tx.plan_arrays(rows=1)
def build(eng):
    t0 = time.perf_counter()
    plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
    dt = (time.perf_counter() - t0) * 1e3
    plan.close()
    return dt
for wait in (0.0, 1.0):
    eng = Engine(0)
    t0 = time.perf_counter(); eng.set_alignments([reads]); ts = time.perf_counter() - t0
    synth.mapping_factory(mapping)._configure(eng)
    time.sleep(wait)
    print("stage %.3f s, wait %.1f s, plan builds (ms):" % (ts, wait), [round(build(eng), 1) for _ in range(4)], flush=True)
    eng.close()
