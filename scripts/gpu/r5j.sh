export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r5j
timeout 600 python - > gpurun_out/r5j/resident.log 2>&1 <<'PY'
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
from plastid_amd import synth
from plastid_amd.engine import Engine
from tests import bam_writer
for n, realistic in ((2e7, True), (1e8, False)):
    genome, tx, reads, mapping = synth.make_config("C2", scale=n / 1e8)
    p = tx.plan_arrays(rows=1)
    path = os.path.join(tempfile.mkdtemp(), "s.bam")
    (bam_writer.write_bam_realistic if realistic else bam_writer.write_bam_packed)(path, reads, threads=16)
    eng = Engine(0)
    synth.mapping_factory(mapping)._configure(eng)
    eng.add_bam(path)
    for _ in range(5):
        t0 = time.perf_counter(); eng.clear_alignments(); eng.add_bam(path); t1 = time.perf_counter()
        plan = eng.plan(p["tid"], p["start"], p["end"], p["strand"], p["out_off"], p["out_step"], p["row_stride"], p["out_elems"], 1)
        got = plan.count(np.int64); t2 = time.perf_counter(); plan.close()
        print("%s %d records: file -> staged %.1f ms, plan+count+read %.1f ms, total %.1f ms = %.3g reads/s" % ("realistic" if realistic else "skeleton", reads.n, (t1-t0)*1e3, (t2-t1)*1e3, (t2-t0)*1e3, reads.n/(t2-t0)), flush=True)
    eng.close(); os.remove(path)
PY
cat gpurun_out/r5j/resident.log | tail -12
