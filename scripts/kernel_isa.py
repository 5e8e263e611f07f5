#!/usr/bin/env python3
"""Disassembly of one kernel of the built library, with a static count per instruction class.
usage: python scripts/kernel_isa.py <substring of the demangled name> [--text] [--lib path]
(static counts: what the code object holds, not what a wave executes -- loops count once)"""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


def disassemble(lib):
    work = tempfile.mkdtemp(prefix="kisa_")
    try:
        shutil.copy(lib, os.path.join(work, "l.so"))
        subprocess.call([LLVM + "llvm-objdump", "--offloading", "l.so"], cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        obj = [f for f in os.listdir(work) if "gfx950" in f][0]
        return subprocess.check_output([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", os.path.join(work, obj)]).decode()
    finally:
        shutil.rmtree(work)


def kernels(text):
    out, name, body = {}, None, []
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if name:
                out[name] = body
            name, body = m.group(1), []
        elif name and line.strip():
            body.append(line.strip())
    if name:
        out[name] = body
    return out


def classify(ins):
    op = ins.split()[0]
    if op.startswith(("s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_setprio", "s_sleep", "s_code_end")):
        return "sync"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_memtime", "s_memrealtime", "s_dcache")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("ds_",)):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = os.environ.get("PLASTID_AMD_LIB") or os.path.join(ROOT, "plastid_amd", "libplastid_counts.so")
    if "--lib" in sys.argv:
        lib = sys.argv[sys.argv.index("--lib") + 1]
        args = [a for a in args if a != lib]
    want = args[0] if args else ""
    ks = kernels(disassemble(lib))
    names = list(ks)
    dem = subprocess.run([shutil.which("c++filt") or "c++filt"], input="\n".join(names).encode(), stdout=subprocess.PIPE).stdout.decode().splitlines()
    for mangled, d in zip(names, dem):
        if want not in d and want not in mangled:
            continue
        body = [l.split("//")[0].strip() for l in ks[mangled]]
        body = [l for l in body if l and not l.endswith(":")]
        cnt = {}
        for l in body:
            c = classify(l)
            cnt[c] = cnt.get(c, 0) + 1
        print("%s\n   %d instructions: %s" % (d[:200], len(body), "  ".join("%s %d" % kv for kv in sorted(cnt.items()))))
        if "--text" in sys.argv:
            print("\n".join("      " + l for l in ks[mangled]))


if __name__ == "__main__":
    main()
