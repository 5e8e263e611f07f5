"""Experiment (GPU box): what ONE wave of k_center costs per stream entry when it runs alone -- a plan of a single
64-position segment over the deepest pile-up of C3 (no cutting: PC_CENTER_FLOOR is raised), timed with the phase events
and with PC_CENTER_DEBUG.  usage: [SCALE=1.0] python scripts/exp_center_lone.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ["PC_CENTER_FLOOR"] = "2000000000"
os.environ["PC_CENTER_DEBUG"] = "1"
from plastid_amd import synth
from plastid_amd.engine import Engine
genome, tx, reads, mapping = synth.make_config("C3", scale=float(os.environ.get("SCALE", "1.0")))
fwd = (reads.flags & 1) == 0
key = reads.tid[fwd].astype(np.int64) * (1 << 32) + (reads.pos[fwd] // 64) * 64
u, c = np.unique(key, return_counts=True)
k = u[np.argmax(c)]
tid, start = int(k >> 32), int(k & 0xffffffff)
print("deepest 64-window: tid %d start %d, %d forward reads start in it" % (tid, start, c.max()))
eng = Engine(0)
eng.add_alignment_file(reads)
synth.mapping_factory(mapping)._configure(eng)
for width in (64, 16, 8):
    a = np.array
    plan = eng.plan(a([tid], np.int32), a([start], np.int64), a([start + width], np.int64), a([1], np.uint8), a([0], np.int64), a([1], np.int8),
                    a([width], np.int64), width, 1)
    plan.launch(np.float64); eng.sync()
    print("width", width)
    plan.launch(np.float64); eng.sync()
    plan.close()
