"""Experiment: phase timings of one BASELINE config under engine knobs (GPU box).
usage: CONFIG=C4 SCALE=0.04 TX=0.5 KNOBS="PC_TILE_G=2048;PC_NO_SMALL=1|PC_TILE_G=1024" python scripts/exp_config.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plastid_amd import synth
from plastid_amd.engine import Engine

cfg = os.environ.get("CONFIG", "C4")
genome, tx, reads, mapping = synth.make_config(cfg, scale=float(os.environ.get("SCALE", "0.04")),
                                               tx_scale=float(os.environ.get("TX", "0.5")))
fac = synth.mapping_factory(mapping)
rows = getattr(fac, "_numlengths", 1)
p = tx.plan_arrays(rows=rows)
print(cfg, "reads", reads.n, "chains", tx.n, "segments", tx.n_segments, "outputs", p["out_elems"], flush=True)
dtype = np.float64 if mapping[0] == "center" else np.int64
for knobs in os.environ.get("KNOBS", "").split("|"):
    for kv in filter(None, knobs.split(";")):
        k, v = kv.split("=")
        os.environ[k] = v
    eng = Engine(0)
    eng.set_alignments([reads])
    fac._configure(eng)
    plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], rows)
    eng.set_profiling(2)
    for _ in range(2):
        plan.launch(dtype)
    eng.sync()
    acc = {}
    for _ in range(5):
        plan.launch(dtype); eng.sync()
        for k, v in eng.last_timing().items():
            acc[k] = acc.get(k, 0) + v / 5
    eng.set_profiling(0)
    plan.launch(dtype); eng.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        plan.launch(dtype)
    eng.sync()
    acc["wall0"] = (time.perf_counter() - t0) / 20 * 1e3
    print("[%s] tiles=%d" % (knobs, plan.tiles), {k: round(v, 4) for k, v in acc.items()}, flush=True)
    plan.close(); eng.close()
    for kv in filter(None, knobs.split(";")):
        os.environ.pop(kv.split("=")[0], None)
