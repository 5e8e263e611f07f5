#!/usr/bin/env python3
"""Condense a scripts/profile.sh output directory into the files kept under profiles/<round>/:
kernel_stats.csv (rocprofv3 --stats), pmc_<pass>_per_kernel.csv (kernel, counter, launches, mean per
launch), rocprofv3_summary.txt, and profiles/traffic.json (HBM bytes per launch of the tile kernel:
FETCH_SIZE x2 + WRITE_SIZE, in KB as rocprofv3 reports them -- MI355X_MICROARCH.md, HBM section).
usage: python scripts/collect_profile.py gpurun_out/prof_<tag> profiles/r01 [bench_line.json]"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, "kernel_stats.csv"))
if os.path.exists(os.path.join(src, "summary.txt")):
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, "rocprofv3_summary.txt"))
means = {}
for sub in ("sq1", "sq2", "fetch", "write", "tcc"):
    files = glob.glob(os.path.join(src, "pmc_" + sub, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    d = defaultdict(list)
    for row in csv.DictReader(open(files[0])):
        d[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
    with open(os.path.join(dst, "pmc_%s_per_kernel.csv" % sub), "w") as fh:
        fh.write("kernel,counter,launches,mean_per_launch\n")
        for (k, c), v in sorted(d.items()):
            fh.write('"%s",%s,%d,%.6g\n' % (k, c, len(v), sum(v) / len(v)))
            means[(k, c)] = sum(v) / len(v)
hist = [k for (k, c) in means if "k_hist_point" in k and c == "FETCH_SIZE"]
if hist:
    k = hist[0]
    fetch_kb, write_kb = means[(k, "FETCH_SIZE")], means.get((k, "WRITE_SIZE"), 0.0)
    tpath = os.path.join(os.path.dirname(os.path.abspath(dst)), "traffic.json")
    t = json.load(open(tpath)) if os.path.exists(tpath) else {}
    t.update({"FETCH_SIZE_KB_per_launch": round(fetch_kb, 1), "WRITE_SIZE_KB_per_launch": round(write_kb, 1),
              "hbm_bytes_per_launch": int(round((2.0 * fetch_kb + write_kb) * 1024))})
    json.dump(t, open(tpath, "w"), indent=1)
    print("traffic: FETCH %.1f KB x2 + WRITE %.1f KB = %d bytes per launch" % (fetch_kb, write_kb, t["hbm_bytes_per_launch"]))
if len(sys.argv) > 3:
    line = [l for l in open(sys.argv[3]).read().splitlines() if l.startswith("{")][-1]
    json.dump(json.loads(line), open(os.path.join(dst, "bench_line.json"), "w"), indent=1)
