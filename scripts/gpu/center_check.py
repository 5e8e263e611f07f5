"""Debug aid: center rule on a small synthetic workload vs the oracle; where do they differ?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle
from plastid_amd import synth
from plastid_amd.engine import Engine
from plastid_amd.packing import concat_file_major

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.002
genome, tx, reads, _ = synth.make_config("C3", scale=scale, tx_scale=0.02)
eng = Engine(0)
eng.set_alignments([reads])
aln = concat_file_major([reads])
for nib in (0, 5):
    mapping = ("center", nib)
    synth.mapping_factory(mapping)._configure(eng)
    p = tx.plan_arrays(rows=1)
    plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
    got = plan.count(np.float64)
    spec = oracle.mapping_spec("center", nib)
    arrays, _ = oracle.count_segments(aln, spec, p["tid"], p["start"], p["end"], p["strand"])
    exp = np.zeros(p["out_elems"], np.float64)
    coord = np.zeros(p["out_elems"], np.int64)
    for s, arr in enumerate(arrays):
        n = arr.shape[-1]
        idx = p["out_off"][s] + p["out_step"][s].astype(np.int64) * np.arange(n)
        exp[idx] = arr
        coord[idx] = p["start"][s] + np.arange(n)
    bad = np.nonzero(got.view(np.uint64) != exp.view(np.uint64))[0]
    print("nibble %d: %d reads, %d outputs, %d differ; sum got %.6f exp %.6f" % (nib, reads.n, len(exp), len(bad), got.sum(), exp.sum()))
    if len(bad):
        print("  coord mod 16 histogram of differing positions:", np.bincount(coord[bad] % 16, minlength=16))
        rel = (got[bad] - exp[bad])
        print("  got-exp: min %.3g max %.3g mean %.3g ; |rel| median %.3g" % (rel.min(), rel.max(), rel.mean(), np.median(np.abs(rel) / np.maximum(exp[bad], 1e-300))))
        for b in bad[:12]:
            print("   out %d coord %d got %.17g exp %.17g" % (b, coord[b], got[b], exp[b]))
    plan.close()
eng.close()
