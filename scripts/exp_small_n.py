"""Experiment: the single-wave class of sparse windows (PC_SMALL_N / PC_SMALL_G / PC_NO_SMALL) on C4 at
several read depths; the knobs are read per count, so one staged data set serves the whole sweep."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plastid_amd import synth
from plastid_amd.engine import Engine

cfg = os.environ.get("CONFIG", "C4")
for scale in [float(x) for x in os.environ.get("SCALES", "0.125,0.5").split(",")]:
    genome, tx, reads, mapping = synth.make_config(cfg, scale=scale, tx_scale=1.0)
    eng = Engine(0)
    eng.set_alignments([reads])
    fac = synth.mapping_factory(mapping)
    fac._configure(eng)
    rows = eng.rows
    p = tx.plan_arrays(rows=rows)
    plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], rows)
    print("%s scale %.3f: %d records, %d segments, %d tiles" % (cfg, scale, reads.n, len(p["tid"]), plan.tiles), flush=True)
    settings = [{}] + [{"PC_SMALL_N": str(n)} for n in (64, 256, 512, 1024, 4096, 8192)] + [{"PC_NO_SMALL": "1"}] + \
               [{"PC_SMALL_G": str(g)} for g in (256, 1024, 2048)] + [{"PC_DEBUG_WORK": "1"}]
    for env in settings:
        for k in ("PC_SMALL_N", "PC_NO_SMALL", "PC_SMALL_G", "PC_DEBUG_WORK"):
            os.environ.pop(k, None)
        os.environ.update(env)
        for _ in range(2):
            plan.launch(np.int64)
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            plan.launch(np.int64)
        eng.sync()
        print("   %-24s %.3f ms" % (env or "default", (time.perf_counter() - t0) / 10 * 1e3), flush=True)
    plan.close()
    # window size (read at plan creation)
    for g in (512, 1024, 4096):
        for k in ("PC_SMALL_N", "PC_NO_SMALL", "PC_SMALL_G", "PC_DEBUG_WORK"):
            os.environ.pop(k, None)
        os.environ["PC_TILE_G"] = str(g)
        plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], rows)
        for sg in (None, 256, 1024):
            if sg is None:
                os.environ.pop("PC_SMALL_G", None)
            else:
                os.environ["PC_SMALL_G"] = str(sg)
            for _ in range(2):
                plan.launch(np.int64)
            eng.sync()
            t0 = time.perf_counter()
            for _ in range(10):
                plan.launch(np.int64)
            eng.sync()
            print("   PC_TILE_G=%d tiles=%d PC_SMALL_G=%s  %.3f ms" % (g, plan.tiles, sg, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
        plan.close()
    os.environ.pop("PC_TILE_G", None); os.environ.pop("PC_SMALL_G", None)
    eng.close()
