import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from plastid_amd import synth
from plastid_amd.engine import Engine
cfg = os.environ.get("CONFIG", "C5")
genome, tx, reads, mapping = synth.make_config(cfg)
print(cfg, reads.n, "records", flush=True)
eng = Engine(0)
eng.set_alignments([reads]); eng.clear_alignments()     # warm-up: ring, pool
for threads in ("6", "10", "16", "6", "10", "16"):
    os.environ["PC_STAGE_UPLOAD_THREADS"] = threads
    eng.reload_knobs() if hasattr(eng, "reload_knobs") else None
    t0 = time.perf_counter()
    eng.set_alignments([reads])
    eng.sync()
    t1 = time.perf_counter()
    print("upload threads %s: stage %.3f s" % (threads, t1 - t0), flush=True)
    eng.clear_alignments()
eng.close()
