"""Build the HIP shared library in-tree (gfx950 only)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "plastid_counts.hip")
def _headers():
    """Every header the library is compiled from: all of ``csrc/*.h`` plus the public header."""
    csrc = os.path.join(HERE, "csrc")
    return sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")) + [
        os.path.join(ROOT, "include", "plastid_counts.h")]


HDRS = _headers()
LIB = os.path.join(HERE, "libplastid_counts.so")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # float64 center counts must be the reference's left-to-right IEEE sums
    "-fno-fast-math", "-ffp-contract=off",
    "-Wall", "-Wno-unused-result",
]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; cannot build the counting engine")


BAM_SRC = os.path.join(HERE, "csrc", "bam_stager.cpp")
BAM_LIB = os.path.join(HERE, "libplastid_bam.so")


def build_bam_library(force=False, verbose=False):
    """Compile the native BAM -> packed-array stager (host C++, zlib; no GPU code)."""
    if not force and os.path.exists(BAM_LIB) and os.path.getmtime(BAM_LIB) >= os.path.getmtime(BAM_SRC):
        return BAM_LIB
    cxx = shutil.which("g++") or shutil.which("c++") or find_hipcc()
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", BAM_SRC, "-o", BAM_LIB, "-lz", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return BAM_LIB


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in [SRC] + _headers())


def build_library(force=False, verbose=False, extra_flags=(), out=None):
    """Compile ``csrc/plastid_counts.hip`` -> ``plastid_amd/libplastid_counts.so``.

    Experiment variants (``extra_flags``) must name their own output file (``out``) and be loaded
    through ``PLASTID_AMD_LIB``: the product library is only ever built with the default flags."""
    if extra_flags and out is None:
        raise ValueError("a build with extra flags needs its own output path (out=...); "
                         "it must not replace the product library")
    target = out or LIB
    if out is None and not force and not needs_build():
        return LIB
    cmd = [find_hipcc()] + HIPCC_FLAGS + list(extra_flags) + [
        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"), SRC, "-o", target]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return target


if __name__ == "__main__":
    import sys
    print(build_library(force="--force" in sys.argv, verbose=True))
    print(build_bam_library(force="--force" in sys.argv, verbose=True))
