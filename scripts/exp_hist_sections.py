"""Experiment: time of the point-rule count of one config with the library named by PLASTID_AMD_LIB (a build variant,
e.g. -DPC_HIST_SKIP=<mask> which leaves sections of k_hist_point out -- results are then wrong, only the time counts).
usage: CONFIG=C4 PLASTID_AMD_LIB=build_variants/skip1.so python scripts/exp_hist_sections.py"""
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
eng = Engine(0)
eng.set_alignments([reads])
factory._configure(eng)
plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], rows)
dt = np.float64 if mapping[0] == "center" else np.int64
for _ in range(3):
    plan.launch(dt)
eng.sync()
n = int(os.environ.get("STEPS", "20"))
t0 = time.perf_counter()
for _ in range(n):
    plan.launch(dt)
eng.sync()
ms = (time.perf_counter() - t0) / n * 1e3
eng.set_profiling(2)
ph = {}
for _ in range(5):
    plan.launch(dt); eng.sync()
    for k, v in eng.last_timing().items():
        ph[k] = ph.get(k, 0.0) + v / 5
print("%s %-28s %.4f ms per count  hist %.4f" % (cfg, os.path.basename(os.environ.get("PLASTID_AMD_LIB", "product")), ms, ph["hist"]), flush=True)
