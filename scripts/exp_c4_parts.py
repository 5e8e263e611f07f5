"""Experiment: C4-like sparse workload, with and without the spliced / gapped reads."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plastid_amd import synth
from plastid_amd.engine import Engine

cfg = os.environ.get("CONFIG", "C4")
genome, tx, reads, mapping = synth.make_config(cfg, scale=float(os.environ.get("SCALE", "0.04")), tx_scale=float(os.environ.get("TX", "0.5")))
fac = synth.mapping_factory(mapping)
rows = getattr(fac, "_numlengths", 1)
p = tx.plan_arrays(rows=rows)
span = reads.ref_end() - reads.pos
print("records", reads.n, "gapped", int((reads.nblk >= 2).sum()), "span>1024", int((span > 1024).sum()), "max span", int(span.max()))
variants = {"all": reads, "ungapped only": reads.subset(np.nonzero(reads.nblk < 2)[0]) if False else None}
keep = np.nonzero(reads.nblk < 2)[0]
from plastid_amd.packing import PackedAlignments
ung = PackedAlignments(reads.tid[keep], reads.pos[keep], reads.alen[keep], reads.flags[keep], reads.nblk[keep],
                       references=reads.references, lengths=reads.lengths, validate=False)
short = np.nonzero(span <= 1024)[0]
for name, rd in (("all", reads), ("ungapped only", ung)):
    eng = Engine(0)
    eng.set_alignments([rd])
    fac._configure(eng)
    plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], rows)
    eng.set_profiling(2)
    for _ in range(2):
        plan.launch(np.int64)
    eng.sync()
    acc = {}
    for _ in range(5):
        plan.launch(np.int64); eng.sync()
        for k, v in eng.last_timing().items():
            acc[k] = acc.get(k, 0) + v / 5
    print(name, rd.n, "tiles=%d" % plan.tiles, {k: round(v, 4) for k, v in acc.items()}, flush=True)
    plan.close(); eng.close()
