"""Experiment: center kernel time vs expression skew (GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plastid_amd import synth
from plastid_amd.engine import Engine
n = int(float(os.environ.get("N", "20e6")))
tx = synth.make_transcripts(synth.YEAST, 20000, 2001, "yeast")
for sigma in [float(x) for x in os.environ.get("SIGMAS", "1.5,0.0").split(",")]:
    reads = synth.make_reads(synth.YEAST, tx, n, 1003, expr_sigma=sigma)
    eng = Engine(0); eng.set_alignments([reads])
    synth.mapping_factory(("center", 0))._configure(eng)
    p = tx.plan_arrays(rows=1)
    plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
    eng.set_profiling(2)
    plan.launch(np.float64); eng.sync()
    acc = {}
    for _ in range(3):
        plan.launch(np.float64); eng.sync()
        for k, v in eng.last_timing().items(): acc[k] = acc.get(k, 0) + v / 3
    print("sigma=%s n=%d" % (sigma, n), {k: round(v, 3) for k, v in acc.items()}, flush=True)
    plan.close(); eng.close()
