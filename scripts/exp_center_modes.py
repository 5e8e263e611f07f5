"""Experiment: the center-rule count of C3 under several settings of the engine's knobs, in ONE process on ONE box
(boxes of the pool differ by +-10 %): records staged once, the knobs re-read between the legs (PC_CENTER_MODE,
PC_CENTER_PWAVES, ...).  Every leg is checked against the first one bit for bit.
usage: python scripts/exp_center_modes.py "PC_CENTER_MODE=0" "PC_CENTER_MODE=2" "PC_CENTER_MODE=2 PC_CENTER_PWAVES=7" ..."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plastid_amd import synth
from plastid_amd.engine import Engine
legs = sys.argv[1:] or ["PC_CENTER_MODE=0", "PC_CENTER_MODE=2"]
cfg = os.environ.get("CONFIG", "C3")
genome, tx, reads, mapping = synth.make_config(cfg, scale=float(os.environ.get("SCALE", "1.0")))
rows = 11 if mapping[0] == "stratified" else 1
p = tx.plan_arrays(rows=rows)
eng = Engine(0)
eng.set_alignments([reads])
synth.mapping_factory(mapping)._configure(eng)
plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], rows)
dt = np.float64 if mapping[0] == "center" else np.int64
ref = None
touched = set()
for rnd in range(int(os.environ.get("ROUNDS", "2"))):
    for leg in legs:
        for k in touched:
            os.environ.pop(k, None)
        for kv in leg.split():
            k, v = kv.split("=")
            os.environ[k] = v
            touched.add(k)
        eng.reload_knobs()
        t0 = time.perf_counter()
        plan.launch(dt); eng.sync()
        first = (time.perf_counter() - t0) * 1e3
        for _ in range(3):
            plan.launch(dt); eng.sync()
        n = int(os.environ.get("STEPS", "30"))
        t0 = time.perf_counter()
        for _ in range(n):
            plan.launch(dt)
        eng.sync()
        ms = (time.perf_counter() - t0) / n * 1e3
        out = plan.read().copy()
        if ref is None:
            ref = out
        same = np.array_equal(out.view(np.uint64), ref.view(np.uint64))
        print("%s round %d %-44s first %.4f ms, steady %.4f ms per count, %s" % (cfg, rnd, leg, first, ms, "bit-identical" if same else "DIFFERENT"), flush=True)
