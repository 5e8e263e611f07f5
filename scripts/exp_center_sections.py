"""Experiment: time of the center-rule count of C3 with the library named by PLASTID_AMD_LIB -- a build variant with
-DPC_CENTER_SKIP=<mask> leaves sections of k_center out (1 replay steps, 2 the stream loop, 4 epilogue, 8 the LDS copy of
the by-length table); the results are then wrong, only the time counts.
usage: PLASTID_AMD_LIB=build_variants/cskip1.so python scripts/exp_center_sections.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plastid_amd import synth
from plastid_amd.engine import Engine
genome, tx, reads, mapping = synth.make_config("C3", scale=float(os.environ.get("SCALE", "1.0")))
p = tx.plan_arrays(rows=1)
eng = Engine(0)
eng.set_alignments([reads])
synth.mapping_factory(mapping)._configure(eng)
plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
for _ in range(3):
    plan.launch(np.float64)
eng.sync()
res = []
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    n = int(os.environ.get("STEPS", "30"))
    t0 = time.perf_counter()
    for _ in range(n):
        plan.launch(np.float64)
    eng.sync()
    res.append((time.perf_counter() - t0) / n * 1e3)
print("C3 %-28s %s ms per count" % (os.path.basename(os.environ.get("PLASTID_AMD_LIB", "product")), " ".join("%.4f" % x for x in res)), flush=True)
