export TMPDIR=/tmp PC_SYNTH_CACHE=/tmp/pc_synth_cache
mkdir -p gpurun_out/r5h
PC_BAM_TIMING=1 PC_STAGE_TIMING=1 timeout 600 python - > gpurun_out/r5h/resident.log 2>&1 <<'PY'
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
from plastid_amd import synth
from plastid_amd.engine import Engine
from tests import bam_writer
genome, tx, reads, mapping = synth.make_config("C2", scale=1.0)
p = tx.plan_arrays(rows=1)
path = os.path.join(tempfile.mkdtemp(), "s.bam")
bam_writer.write_bam_packed(path, reads, threads=16)
eng = Engine(0)
synth.mapping_factory(mapping)._configure(eng)
eng.add_bam(path)
for k in range(8):
    t0 = time.perf_counter(); eng.clear_alignments(); t1 = time.perf_counter(); eng.add_bam(path); t2 = time.perf_counter()
    print("PASS %d: clear %.1f ms, file -> staged %.1f ms" % (k, (t1-t0)*1e3, (t2-t1)*1e3), flush=True)
eng.close()
PY
grep -B14 "PASS" gpurun_out/r5h/resident.log | grep "PASS\|[0-9][0-9][0-9]\.[0-9]* ms" | head -60
