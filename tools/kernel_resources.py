#!/usr/bin/env python
"""Compile the HIP library with -Rpass-analysis=kernel-resource-usage (into a temporary file: the product .so is not touched)
and print one line per kernel whose mangled name contains a pattern.
  python tools/kernel_resources.py lstm_xcd_kernelILi5 [extra hipcc flags...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-shared", "-fPIC",
       os.path.join(ROOT, "onssen_amd/csrc/onssen_hip.hip"), "-o", os.path.join(tempfile.mkdtemp(prefix="onssen_res_"), "lib.so"),
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark:\s+(?:Function Name|Name): (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
    elif "error" in line:
        print(line)
for k, v in rows.items():
    if pat in k:
        print(f"{k[:70]:70s} VGPR {v.get('VGPRs')} AGPR {v.get('AGPRs')} SGPR {v.get('TotalSGPRs')} spillS {v.get('SGPRs Spill')} spillV {v.get('VGPRs Spill')} "
              f"scratch {v.get('ScratchSize')} occ {v.get('Occupancy')} LDS {v.get('LDS Size')}")
