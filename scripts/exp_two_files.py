"""Experiment: the same records as one file vs dealt alternately into two files (C2)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plastid_amd import synth
from plastid_amd.engine import Engine
from plastid_amd.packing import PackedAlignments
genome, tx, reads, mapping = synth.make_config("C2", scale=float(os.environ.get("SCALE", "1.0")))
multi = np.nonzero(reads.nblk >= 2)[0]
rec_of_run = np.repeat(multi, reads.nblk[multi])
halves = []
for k in (0, 1):
    sel = np.arange(k, reads.n, 2)
    runs = np.nonzero((rec_of_run & 1) == k)[0]
    halves.append(PackedAlignments(reads.tid[sel], reads.pos[sel], reads.alen[sel], reads.flags[sel], reads.nblk[sel],
                                   reads.blk_start[runs], reads.blk_len[runs], references=reads.references,
                                   lengths=reads.lengths, validate=False))
p = tx.plan_arrays(rows=1)
eng = Engine(0)
ref = None
for name, files in (("one file", [reads]), ("two files", halves)):
    eng.set_alignments(files)
    synth.mapping_factory(mapping)._configure(eng)
    plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
    out = plan.count(np.int64)
    if ref is None:
        ref = out
    assert np.array_equal(out, ref)
    eng.set_profiling(2)
    plan.launch(np.int64); eng.sync()
    ph = eng.last_timing()
    eng.set_profiling(0)
    for _ in range(3):
        plan.launch(np.int64)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        plan.launch(np.int64)
    eng.sync()
    print("%-10s %.3f ms per count" % (name, (time.perf_counter() - t0) / 20 * 1e3), {k: round(v, 3) for k, v in ph.items()}, flush=True)
    plan.close()
