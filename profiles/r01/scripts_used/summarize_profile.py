#!/usr/bin/env python3
"""Summarise rocprofv3 output dirs written by scripts/profile.sh: per-kernel durations
from the kernel trace, per-kernel PMC averages from the counter passes."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def short(name):
    name = name.split("(")[0]
    for k in ("k_hist_point", "k_tile_ranges", "k_long_point", "k_center_weigh", "k_center_order", "k_center", "k_gather", "k_total",
              "k_unmappable", "k_mapped"):
        if k in name:
            return k + ("<%s>" % name.split("<")[1].split(">")[0] if "<" in name and k == "k_gather" else "")
    return name[:60]


print("== kernel trace (durations, ns)")
for f in find("trace/**/*kernel_trace.csv"):
    d = defaultdict(list)
    for row in csv.DictReader(open(f)):
        d[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    tot = sum(sum(v) for v in d.values())
    print("%-40s %8s %12s %12s %12s %7s" % ("kernel", "calls", "avg_ns", "min_ns", "max_ns", "pct"))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print("%-40s %8d %12.0f %12d %12d %6.1f%%" % (k, len(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / tot))
for f in find("trace/**/*kernel_stats.csv"):
    print("== rocprofv3 --stats:", os.path.relpath(f, out))
    print(open(f).read()[:3000])

for sub in ("pmc_sq1", "pmc_sq2", "pmc_sq3", "pmc_fetch", "pmc_write", "pmc_tcc"):
    for f in find(sub + "/**/*counter_collection.csv"):
        d = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            d[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== %s (per-launch average)" % sub)
        for k, cs in sorted(d.items()):
            print("  %-36s " % k + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
