"""Experiment: the inflate lap of the GPU BAM decoder with the library named by PLASTID_AMD_LIB -- a build variant with
-DPC_BGZF_SKIP=<mask> leaves sections of k_bgzf_inflate out (the output is then wrong and the CRC check fails: the lap is
printed before it, PC_BAM_TIMING=1).  usage: PLASTID_AMD_LIB=build_variants/bskip1.so python scripts/exp_bam_sections.py 2e7"""
import ctypes
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plastid_amd import synth  # noqa: E402
from tests import bam_writer  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
path = os.path.join(tempfile.gettempdir(), "pc_bamsec_%d.bam" % n)
if not os.path.exists(path):
    genome, tx, reads, mapping = synth.make_config("C2", scale=n / 1e8)
    bam_writer.write_bam_realistic(path, reads, threads=16)
lib = os.environ.get("PLASTID_AMD_LIB") or os.path.join(ROOT, "plastid_amd", "libplastid_counts.so")
L = ctypes.CDLL(lib)
L.pc_last_error.restype = ctypes.c_char_p
e = ctypes.c_void_p()
assert L.pc_create(0, ctypes.byref(e)) == 0
os.environ["PC_BAM_TIMING"] = "1"
print("== %s (%d records, %.0f MB)" % (os.path.basename(lib), n, os.path.getsize(path) / 1e6), file=sys.stderr, flush=True)
for rep in range(3):
    h = ctypes.c_void_p()
    rc = L.pc_bam_open_path(e, path.encode(), ctypes.byref(h))
    print("   pass %d rc %d %s" % (rep, rc, (L.pc_last_error() or b"").decode()[:60] if rc else ""), file=sys.stderr, flush=True)
    if rc == 0:
        L.pc_bam_close(h)
