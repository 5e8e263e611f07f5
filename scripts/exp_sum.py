"""Experiment: hist kernel time with per-position outputs vs fused per-chain sums (no output stream)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plastid_amd import synth
from plastid_amd.engine import Engine

n = int(float(os.environ.get("N", "100e6")))
tx = synth.make_transcripts(synth.YEAST, 20000, 2001, "yeast")
reads = synth.make_reads(synth.YEAST, tx, n, 1002, expr_sigma=float(os.environ.get("SIGMA", "1.5")))
eng = Engine(0)
eng.set_alignments([reads])
synth.mapping_factory(("fiveprime", 12))._configure(eng)
p = tx.plan_arrays(rows=1)
seg_tx = tx.ex_tx
variants = {
    "positions": (p["out_off"], p["out_step"], p["row_stride"], p["out_elems"]),
    "chain sums": (seg_tx.astype(np.int64), np.zeros(len(seg_tx), np.int8), np.ones(len(seg_tx), np.int64), tx.n),
}
for name, (off, step, stride, nout) in variants.items():
    plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], off, step, stride, nout, 1)
    eng.set_profiling(2)
    for _ in range(3):
        plan.launch(np.int64)
    eng.sync()
    acc = {}
    for _ in range(10):
        plan.launch(np.int64); eng.sync()
        for k, v in eng.last_timing().items():
            acc[k] = acc.get(k, 0) + v / 10
    import time
    for lvl in (2, 1, 0):
        eng.set_profiling(lvl)
        plan.launch(np.int64); eng.sync()
        t0 = time.perf_counter()
        for _ in range(200):
            plan.launch(np.int64)
        eng.sync()
        acc["wall%d" % lvl] = (time.perf_counter() - t0) / 200 * 1e3
    print(name, "tiles=%d" % plan.tiles, {k: round(v, 4) for k, v in acc.items()}, flush=True)
    plan.close()
eng.close()
