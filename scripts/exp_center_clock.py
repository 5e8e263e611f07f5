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
# one '+' segment per contig, laid out left to right: output index == genome position, so the lanes of a
# wave sit next to each other in the result
lens = np.asarray(reads.lengths, np.int64)
off = np.zeros(len(lens) + 1, np.int64); np.cumsum(lens, out=off[1:])
plan = eng.plan(np.arange(len(lens), dtype=np.int32), np.zeros(len(lens), np.int64), lens, np.full(len(lens), 1, np.uint8),
                off[:-1], np.ones(len(lens), np.int8), lens, int(off[-1]), 1)
out = plan.count(np.float64)
w = np.nonzero(out[6:] >= 7e15)[0]; w = w[(w - np.searchsorted(off, w, side='right') * 0) >= 0]                       # lane 6 marks a reporting wave; lanes 0..5 sit before it
tot, nb, wait, filt, rep, ent = (out[w + k] for k in range(6))
code = out[w + 6] - 7e15
for name, sel in (("all reporting waves", code >= 0), ("whole chunks (code 0)", code == 0), ("cut chunks (code > 0)", code > 0)):
    if not sel.any():
        continue
    T, B = tot[sel].sum(), nb[sel].sum()
    print("%-24s waves %d  cycles %.3g (max %.3g)  batches %.3g  entries %.3g" % (name, sel.sum(), T, tot[sel].max(), B, ent[sel].sum()))
    print("    share of wave cycles: wait for records %.1f%%, filter+compact %.1f%%, replay %.1f%%, rest (start-up, index look-ups, loop) %.1f%%" % (
        100 * wait[sel].sum() / T, 100 * filt[sel].sum() / T, 100 * rep[sel].sum() / T, 100 * (T - wait[sel].sum() - filt[sel].sum() - rep[sel].sum()) / T))
    print("    per batch: wait %.0f, filter %.0f, replay %.0f cycles; %.1f entries per batch, %.0f replay cycles per entry" % (
        wait[sel].sum() / B, filt[sel].sum() / B, rep[sel].sum() / B, ent[sel].sum() / B, rep[sel].sum() / max(ent[sel].sum(), 1)))
eng.set_profiling(2); plan.launch(np.float64); eng.sync(); print(eng.last_timing())
