#!/bin/bash
# The host BAM decoder (plastid_amd/csrc/bam_stager.cpp) under AddressSanitizer + UBSan on the CPU: the htslib fixture,
# the odd / damaged auxiliary fields, region reads and a file cut off at every 997th byte.  (GPU sanitizers are not
# available on the pool; this is the CPU build only.)  usage: bash scripts/asan_bam_stager.sh
set -e
cd "$(dirname "$0")/.."
OUT=/tmp/libplastid_bam_asan.so
g++ -O1 -g -std=c++17 -fPIC -shared -pthread -fsanitize=address,undefined -fno-omit-frame-pointer plastid_amd/csrc/bam_stager.cpp -o $OUT -lz -ldl
LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
python - <<'PY'
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import plastid_amd.bam as B
B.BAM_LIB = "/tmp/libplastid_bam_asan.so"
import plastid_amd.build as bld
bld.BAM_LIB = B.BAM_LIB
from tests import bam_writer
from plastid_amd.bam import read_bam
tmp = tempfile.mkdtemp()
recs, want = bam_writer.odd_aux_records()
p = os.path.join(tmp, "aux.bam")
bam_writer.write_bam(p, ["c"], [100000], recs)
assert read_bam(p).nh.tolist() == want
import numpy as np
hts = np.load(os.path.join("tests", "golden", "hts_fixture.npz"))
fix = os.path.join(tmp, "htslib.bam")
raw = hts["bam"].tobytes()
open(fix, "wb").write(raw)
open(fix + ".bai", "wb").write(hts["bai"].tobytes())
for threads in (1, 4):
    a = read_bam(fix, threads=threads)
print("fixture:", a.n, "records,", int((a.nh > 0).sum()), "with NH")
print("region read:", read_bam(fix, regions=[(a.references[0], 0, 50000), (a.references[-1], 100, 2000)]).n)
bad = 0
step = max(1, len(raw) // 400)
for cut in range(step, len(raw), step):
    q = os.path.join(tmp, "cut.bam")
    open(q, "wb").write(raw[:cut])
    try:
        read_bam(q)
    except (ValueError, IOError, RuntimeError):
        bad += 1
print("files cut off at %d places: %d rejected, none crashed" % (len(range(step, len(raw), step)), bad))
rng = np.random.default_rng(7)
flips = 0
for k in range(300):                       # single damaged bytes (most are caught by the member CRC; none may crash)
    b = bytearray(raw)
    at = int(rng.integers(0, len(b)))
    b[at] ^= 1 << int(rng.integers(0, 8))
    q = os.path.join(tmp, "flip.bam")
    open(q, "wb").write(bytes(b))
    try:
        read_bam(q)
    except (ValueError, IOError, RuntimeError):
        flips += 1
print("300 files with one flipped bit: %d rejected, none crashed" % flips)
print("sanitizers: clean")
PY
