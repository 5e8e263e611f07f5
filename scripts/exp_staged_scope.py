"""Experiment: where the staged scope of a config goes -- staging laps (PC_STAGE_TIMING=1 prints them), plan build,
one count, read-back.  usage: PC_STAGE_TIMING=1 CONFIG=C4 python scripts/exp_staged_scope.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plastid_amd import synth
from plastid_amd.engine import Engine
cfg = os.environ.get("CONFIG", "C4")
genome, tx, reads, mapping = synth.make_config(cfg, scale=float(os.environ.get("SCALE", "1.0")))
factory = synth.mapping_factory(mapping)
rows = getattr(factory, "_numlengths", 1)
p = tx.plan_arrays(rows=rows)
print("%s: %d records, %d runs, %d segments, %d outputs" % (cfg, reads.n, len(reads.blk_start), tx.n_segments, int(p["out_elems"])), flush=True)
for rep in range(int(os.environ.get("REPS", "2"))):
    eng = Engine(0)
    t0 = time.perf_counter()
    eng.set_alignments([reads])
    t1 = time.perf_counter()
    factory._configure(eng)
    plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], rows)
    t2 = time.perf_counter()
    plan.launch(np.int64)
    eng.sync()
    t3 = time.perf_counter()
    out = plan.read()
    t4 = time.perf_counter()
    print("rep %d: stage %.3f s  plan %.1f ms  first count %.2f ms  read-back %.3f s  -> staged %.3g reads/s" % (
        rep, t1 - t0, (t2 - t1) * 1e3, (t3 - t2) * 1e3, t4 - t3, reads.n / (t4 - t0 - (t2 - t1))), flush=True)
    del out
    plan.close()
    eng.close()
