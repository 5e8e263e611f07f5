"""Experiment: compressed BAM -> counts, host decoder against the GPU decoder, on the realistic sample of bench.py
(records as an aligner writes them: ~119 bytes each) and on the skeleton file.  Prints the phase times."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plastid_amd import synth  # noqa: E402
from plastid_amd.bam import read_bam, read_bam_gpu  # noqa: E402
from plastid_amd.engine import Engine  # noqa: E402
from tests import bam_writer  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3000000
realistic = (sys.argv[2] if len(sys.argv) > 2 else "realistic") == "realistic"
genome, tx, reads, mapping = synth.make_config("C2", scale=n / 1e8)
factory = synth.mapping_factory(mapping)
p = tx.plan_arrays(rows=1)
tmp = tempfile.mkdtemp(prefix="pc_bamexp_")
path = os.path.join(tmp, "s.bam")
writer = bam_writer.write_bam_realistic if realistic else bam_writer.write_bam_packed
nbytes = writer(path, reads, threads=16)
print("file: %d records, %.1f MB compressed, %.1f MB inflated (%.0f B/record)" % (reads.n, os.path.getsize(path) / 1e6, nbytes / 1e6, nbytes / reads.n))
eng = Engine(0)
factory._configure(eng)
read_bam(path)
for name in ("host", "gpu", "host", "gpu", "gpu"):
    timing = {}
    t0 = time.perf_counter()
    packed = read_bam(path) if name == "host" else read_bam_gpu(path, eng, timing=timing)
    t1 = time.perf_counter()
    eng.set_alignments([packed])
    t2 = time.perf_counter()
    plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
    got = plan.count(np.int64)
    t3 = time.perf_counter()
    plan.close()
    ok = all(np.array_equal(getattr(packed, k), getattr(reads, k)) for k in ("tid", "pos", "alen", "flags", "nblk", "blk_start", "blk_len"))
    print("%-4s decode %.1f ms  stage %.1f ms  plan+count+read %.1f ms  total %.1f ms = %.3g reads/s  columns ok: %s  %s" % (
        name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3, reads.n / (t3 - t0), ok,
        {k: (round(v, 2) if isinstance(v, float) else v) for k, v in timing.items()}))
eng.close()
os.remove(path)
os.rmdir(tmp)
