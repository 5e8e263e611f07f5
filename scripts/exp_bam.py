"""Experiment: phases of the native BAM reader (PB_TIMING=1) on a synthetic coordinate-sorted BAM
(SCALE x the C2 reads), for several thread counts."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PB_TIMING"] = "1"
from plastid_amd import bam, synth
from tests import bam_writer as bw
genome, tx, reads, _ = synth.make_config("C2", scale=float(os.environ.get("SCALE", "0.05")), tx_scale=0.05)
path = "/tmp/exp_bam_%d.bam" % os.getpid()
t0 = time.perf_counter()
bw.write_bam_packed(path, reads, threads=min(16, os.cpu_count() or 1))
print("BAM: %d records, %.1f MB, written in %.1f s" % (reads.n, os.path.getsize(path) / 1e6, time.perf_counter() - t0), flush=True)
for th in [int(x) for x in os.environ.get("THREADS", "0,0,128,64,32,16").split(",")]:
    t0 = time.perf_counter()
    p = bam.read_bam(path, threads=th)
    dt = time.perf_counter() - t0
    print("threads=%d read_bam %.3f s = %.1f M records/s" % (th, dt, p.n / dt / 1e6), flush=True)
assert all(np.array_equal(getattr(p, k), getattr(reads, k)) for k in ("tid", "pos", "alen", "flags", "nblk", "blk_start", "blk_len"))

# where the time outside the decoder's own laps goes (the ctypes calls of read_bam, one by one)
import ctypes
L = bam._load()
bw.write_bam_packed(path, reads, threads=min(16, os.cpu_count() or 1))
for rep in range(3):
    t = [time.perf_counter()]
    h = L.pb_open(os.fsencode(path)); t.append(time.perf_counter())
    L.pb_load(h, int(os.environ.get("T2", "128"))); t.append(time.perf_counter())
    counts = np.zeros(4, np.int64); L.pb_counts(h, counts.ctypes.data_as(ctypes.c_void_p))
    n, nrun = int(counts[0]), int(counts[1])
    arrs = [np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.uint16), np.empty(n, np.uint8), np.empty(n, np.uint8),
            np.empty(nrun, np.int32), np.empty(nrun, np.int32)]; t.append(time.perf_counter())
    L.pb_fill(h, *[a.ctypes.data_as(ctypes.c_void_p) for a in arrs]); t.append(time.perf_counter())
    L.pb_close(h); t.append(time.perf_counter())
    print("open %.1f load %.1f alloc %.1f fill %.1f close %.1f ms" % tuple(1e3 * (b - a) for a, b in zip(t, t[1:])), flush=True)

# the same calls made from read_bam itself, timed through wrappers
bw.write_bam_packed(path, reads, threads=min(16, os.cpu_count() or 1))
spent = {}
def timed(name, fn):
    def w(*a):
        t0 = time.perf_counter()
        r = fn(*a)
        spent[name] = spent.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w
for nm in ("pb_open", "pb_load", "pb_counts", "pb_fill", "pb_close", "pb_nref", "pb_ref_name", "pb_ref_length"):
    setattr(L, nm, timed(nm, getattr(L, nm)))
for rep in range(3):
    spent.clear()
    t0 = time.perf_counter()
    p = bam.read_bam(path, threads=128)
    dt = time.perf_counter() - t0
    print("read_bam %.1f ms: %s" % (1e3 * dt, ", ".join("%s %.1f" % (k, 1e3 * v) for k, v in spent.items())), flush=True)
os.unlink(path)
