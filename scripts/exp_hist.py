"""Experiment: hist kernel time vs expression skew / work-item size (GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plastid_amd import synth
from plastid_amd.engine import Engine

n = int(float(os.environ.get("N", "100e6")))
tx = synth.make_transcripts(synth.YEAST, 20000, 2001, "yeast")
for sigma in [float(x) for x in os.environ.get("SIGMAS", "1.5,0.0").split(",")]:
    reads = synth.make_reads(synth.YEAST, tx, n, 1002, expr_sigma=sigma)
    eng = Engine(0)
    eng.set_profiling(2)
    eng.set_alignments([reads])
    synth.mapping_factory(("fiveprime", 12))._configure(eng)
    p = tx.plan_arrays(rows=1)
    for RG in os.environ.get("RS", "32768").split(","):
        R, _, Gs = RG.partition(":")
        os.environ["PC_WORK_R"] = R
        if Gs:
            os.environ["PC_TILE_G"] = Gs
        plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
        for _ in range(3):
            plan.launch(np.int64)
        eng.sync()
        acc = {}
        for _ in range(10):
            plan.launch(np.int64); eng.sync()
            for k, v in eng.last_timing().items():
                acc[k] = acc.get(k, 0) + v / 10
        print("sigma=%s R=%s tiles=%d" % (sigma, RG, plan.tiles), {k: round(v, 4) for k, v in acc.items()}, flush=True)
        plan.close()
    eng.close()
