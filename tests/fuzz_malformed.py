"""Malformed staging inputs through the C ABI (GPU box; test infrastructure, not collected):
    python tests/fuzz_malformed.py [cases]
Every case corrupts one field of a valid packed file.  pc_add_alignment_file must either reject it
(ValueError / EngineError) or accept a still-consistent file, in which case a count must run."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plastid_amd as pa
from plastid_amd import synth
from plastid_amd.engine import Engine
from plastid_amd.exceptions import EngineError

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
genome, tx, reads, _ = synth.make_config("C4", scale=0.00005, tx_scale=0.001)
p = tx.plan_arrays(rows=1)
rng = np.random.default_rng(9)
rejected = accepted = 0
for it in range(n_cases):
    a = {k: getattr(reads, k).copy() for k in ("tid", "pos", "alen", "flags", "nblk", "blk_start", "blk_len")}
    field = str(rng.choice(list(a)))
    arr = a[field]
    if len(arr):
        mode = int(rng.integers(0, 4))
        i = int(rng.integers(0, len(arr)))
        if mode == 0:
            arr[i] = rng.integers(np.iinfo(arr.dtype).min, np.iinfo(arr.dtype).max, dtype=np.int64).astype(arr.dtype)
        elif mode == 1:
            arr[i:] = arr[i:][::-1].copy()
        elif mode == 2:
            a[field] = arr[:i].copy()
        else:
            arr[i] = 0
    eng = Engine(0)
    try:
        bad = pa.PackedAlignments(a["tid"][:len(a["pos"])] if len(a["tid"]) > len(a["pos"]) else a["tid"], a["pos"], a["alen"],
                                  a["flags"], a["nblk"], a["blk_start"], a["blk_len"], references=reads.references,
                                  lengths=reads.lengths, validate=False)
        eng.set_alignments([bad])
    except (ValueError, EngineError, AssertionError, IndexError):
        rejected += 1
        eng.close()
        continue
    accepted += 1
    pa.FivePrimeMapFactory(3)._configure(eng)
    plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
    plan.count(np.int64)
    pa.CenterMapFactory(1)._configure(eng)
    plan.count(np.float64)
    plan.close()
    eng.close()
print("rejected", rejected, "accepted and counted", accepted)
