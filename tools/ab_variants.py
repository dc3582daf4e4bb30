#!/usr/bin/env python
"""Same-box A/B of library variants.  Boxes differ by +-4 % (round 1: the same binary gave 2.38 and 2.56 ms for the
headline step on two boxes), so a comparison is only worth anything when both binaries run inside ONE gpurun call.

  build (here, no GPU):   python tools/ab_variants.py build base "" poll0 "-DONSSEN_SOME_SWITCH=1" ...
                          -> build_variants/libonssen_hip_<name>.so, one per (name, extra hipcc flags) pair; a variant can
                             also be produced by hand (git stash / build / copy) under the same file name
  run (on the GPU box):   gpurun -- 'python tools/ab_variants.py run base poll0 -- bench.py --no-cpu-baseline --steps 40'
                          -> every variant twice, interleaved, printing ms_per_step and the recurrence us per time step;
                             the in-tree library is restored afterwards
build_variants/ is git-ignored but travels to the GPU box with the snapshot."""
import json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "onssen_amd", "libonssen_hip.so")
VDIR = os.path.join(ROOT, "build_variants")
SRC = os.path.join(ROOT, "onssen_amd", "csrc", "onssen_hip.hip")


def vpath(name):
    return os.path.join(VDIR, f"libonssen_hip_{name}.so")


def build(pairs):
    os.makedirs(VDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    for name, flags in pairs:
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-shared", "-fPIC",
               SRC, "-o", vpath(name)] + flags.split()
        print(" ".join(cmd))
        subprocess.check_call(cmd)


def run(names, cmd):
    keep = LIB + ".ab_keep"
    shutil.copy2(LIB, keep)
    try:
        for rep in range(2):
            for name in names:
                shutil.copy2(vpath(name), LIB)
                out = subprocess.run([sys.executable] + cmd, cwd=ROOT, capture_output=True, text=True, timeout=600).stdout.strip().splitlines()
                try:
                    r = json.loads(out[-1])
                    extra = r.get("roofline", {}).get("us_per_time_step")
                    gem = (r.get("roofline", {}).get("other_kernels") or {}).get("ms_by_call") or {}
                    print(f"{name:16s} ms_per_step {r['ms_per_step']:.4f}" + (f"  recurrence us/step {extra:.3f}" if extra else "")
                          + ("  gemm ms " + " ".join(f"{k}={v:.4f}" for k, v in gem.items() if v) if gem else ""), flush=True)
                except Exception:
                    print(f"{name:16s} FAILED: {out[-3:]}", flush=True)
    finally:
        shutil.move(keep, LIB)


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "build":
        args = sys.argv[2:]
        build(list(zip(args[0::2], args[1::2])))
    elif len(sys.argv) >= 4 and sys.argv[1] == "run" and "--" in sys.argv:
        i = sys.argv.index("--")
        run(sys.argv[2:i], sys.argv[i + 1:])
    else:
        sys.exit(__doc__)
