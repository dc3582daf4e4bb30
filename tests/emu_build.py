"""Build (if stale) and load the host-side kernel-test library: the product's
onssen_hip.hip compiled with g++ against the mock HIP runtime in tests/emu."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "onssen_amd", "csrc", "onssen_hip.hip")
HDRS = [os.path.join(ROOT, "tests", "emu", "include", "hip", "hip_runtime.h"),
        os.path.join(ROOT, "include", "onssen_hip.h")]
OUT = os.path.join(ROOT, "tests", "emu", "libonssen_emu.so")


def build_emu():
    # ONSSEN_EMU_CXXFLAGS="-D..." builds an experiment variant of the same source into its own file
    global OUT
    extra = os.environ.get("ONSSEN_EMU_CXXFLAGS", "").split()
    if extra:
        import hashlib
        OUT = os.path.join(ROOT, "tests", "emu", "libonssen_emu_%s.so" % hashlib.md5(" ".join(extra).encode()).hexdigest()[:8])
    csrc = os.path.dirname(SRC)
    parts = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".inc"))]   # one TU, several files
    newest = max(os.path.getmtime(f) for f in parts + HDRS)
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < newest:
        tmp = OUT + ".tmp%d" % os.getpid()      # build beside it, then rename: a process that has the old file mapped keeps its inode
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O2", "-I",
                               os.path.join(ROOT, "tests", "emu", "include"), "-pthread", "-shared", "-fPIC",
                               SRC, "-o", tmp] + extra)
        os.replace(tmp, OUT)
    return OUT


def load_emu():
    from onssen_amd._abi import Lib
    return Lib(build_emu())
