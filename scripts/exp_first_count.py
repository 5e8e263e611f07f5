import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from plastid_amd import synth
from plastid_amd.engine import Engine
cfg = os.environ.get("CONFIG", "C5")
genome, tx, reads, mapping = synth.make_config(cfg)
factory = synth.mapping_factory(mapping)
rows = getattr(factory, "_numlengths", 1)
p = tx.plan_arrays(rows=rows)
eng = Engine(0)
eng.set_alignments([reads])
factory._configure(eng)
def mk():
    return eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], rows)
for rep in range(3):
    for prof in (0, 2, 0, 2):
        plan = mk()
        eng.sync()
        eng.set_profiling(prof)
        t0 = time.perf_counter(); plan.launch(np.int64); eng.sync(); t1 = time.perf_counter()
        ev = eng.last_timing() if prof else None
        eng.set_profiling(0)
        t2 = time.perf_counter(); plan.launch(np.int64); eng.sync(); t3 = time.perf_counter()
        t4 = time.perf_counter(); plan.launch(np.int64); eng.sync(); t5 = time.perf_counter()
        print("profiling %d: first %.3f ms (events: %s), second %.3f, third %.3f" % (prof, (t1 - t0) * 1e3, ("%.3f / lists %.3f" % (ev["total"], ev["worklist"])) if ev else "-", (t3 - t2) * 1e3, (t5 - t4) * 1e3), flush=True)
        plan.close()
eng.close()
