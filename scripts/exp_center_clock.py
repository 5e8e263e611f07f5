"""Experiment (build with -DPC_EXP_CENTER_CLOCK): per-wave cycle counts of k_center."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plastid_amd import build
build.build_library(force=True, extra_flags=["-DPC_EXP_CENTER_CLOCK"])
from plastid_amd import synth
from plastid_amd.engine import Engine
genome, tx, reads, mapping = synth.make_config("C3", scale=float(os.environ.get("SCALE", "0.2")), tx_scale=1.0)
eng = Engine(0); eng.set_alignments([reads]); synth.mapping_factory(mapping)._configure(eng)
# one segment per (contig, strand) so that output index == island position
names = reads.references
p = tx.plan_arrays(rows=1)
plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
out = plan.count(np.float64)
cyc = out[(out > 1000) & (out < 9e8)]      # lane-0 values (cycle counts); real center counts stay far below 1000 here
print("kernel-clock experiment"); print("waves seen", len(cyc), "sum cycles %.3g" % cyc.sum(), "max %.3g" % cyc.max(), "p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f" % tuple(np.percentile(cyc, [50, 90, 99, 99.9])))
eng.set_profiling(2); plan.launch(np.float64); eng.sync(); print(eng.last_timing())
idx = np.nonzero(out > 1000)[0]
idx = idx[out[idx] < 9e8]
order = idx[np.argsort(out[idx])[-12:]]
for i in order:
    print("cycles %.0f  batches(long*1e6+near) %.0f  code %.0f" % (out[i], out[i + 1] - 1e9, out[i + 2] - 2e9))
