"""Experiment (GPU box): k_center time vs the cut thresholds PC_CENTER_T1 / PC_CENTER_T2 on C3
(chunks with more than T1 x / T1*T2 x the mean candidate count are cut into 4 / 8 sub-chunks)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plastid_amd import synth
from plastid_amd.engine import Engine
scale = float(os.environ.get("SCALE", "1.0"))
genome, tx, reads, mapping = synth.make_config("C3", scale=scale)
eng = Engine(0); eng.set_alignments([reads])
synth.mapping_factory(mapping)._configure(eng)
p = tx.plan_arrays(rows=1)
plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
ref = None
for t1, t2 in [(8, 4), (16, 4), (32, 4), (64, 4), (128, 4), (256, 4), (64, 2), (128, 2), (32, 8), (100000, 4)]:
    os.environ["PC_CENTER_T1"] = str(t1); os.environ["PC_CENTER_T2"] = str(t2)
    eng.reload_knobs()
    eng.set_profiling(2)
    plan.launch(np.float64); eng.sync()
    acc = {}
    for _ in range(3):
        plan.launch(np.float64); eng.sync()
        for k, v in eng.last_timing().items(): acc[k] = acc.get(k, 0) + v / 3
    got = plan.read()
    if ref is None: ref = got
    print("T1=%d T2=%d" % (t1, t2), {k: round(v, 3) for k, v in acc.items()}, "same bits as first:", np.array_equal(ref.view(np.uint64), got.view(np.uint64)), flush=True)
plan.close(); eng.close()
