"""Experiment: pc_plan_create of the human-scale annotation (60 k transcripts, 479 k exons) with the host builder and
with the GPU builder (csrc/plan_kernels.hip.h), for the single-row plan of C4 and the 11-row plan of C5.
PC_STAGE_TIMING=1 prints the laps of either."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plastid_amd import synth  # noqa: E402
from plastid_amd.engine import Engine  # noqa: E402

genome, tx, reads, mapping = synth.make_config("C4", scale=0.002, tx_scale=1.0)
for rows in (1, 11):
    p = tx.plan_arrays(rows=rows)
    for where in ("host", "gpu"):
        os.environ["PC_PLAN_BUILD"] = where
        eng = Engine(0)
        eng.set_alignments([reads])
        times = []
        for _ in range(6):
            t0 = time.perf_counter()
            plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], rows)
            eng.sync()
            times.append((time.perf_counter() - t0) * 1e3)
            tiles = plan.tiles
            plan.close()
        print("rows %2d  %-4s builder: %d segments, %d tiles; plan builds (ms, sync included): %s" % (
            rows, where, len(p["tid"]), tiles, " ".join("%.2f" % t for t in times)), flush=True)
        eng.close()
