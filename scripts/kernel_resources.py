#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel in the built library (from the code object's notes).
usage: python scripts/kernel_resources.py [substring]"""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.environ.get("PLASTID_AMD_LIB") or os.path.join(ROOT, "plastid_amd", "libplastid_counts.so")
work = tempfile.mkdtemp(prefix="kobj_")
shutil.copy(lib, os.path.join(work, "l.so"))
subprocess.call(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", "l.so"], cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
obj = [f for f in os.listdir(work) if "gfx950" in f][0]
txt = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", os.path.join(work, obj)]).decode()
want = sys.argv[1] if len(sys.argv) > 1 else ""
for b in txt.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b)
    if name and want in name.group(1):
        g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
        print("%-60s vgpr %3s sgpr %3s lds %6s scratch %s" % (name.group(1)[:60], g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
shutil.rmtree(work)
