#!/usr/bin/env python
"""Profiling aid (GPU box): onssen_linear_x3p timing for one shape; ONSSEN_X3_ABLATE / ONSSEN_X3_GN are read at first call."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd.hip import get_lib
from onssen_amd._abi import Lib
lib = Lib(os.environ["ONSSEN_PROBE_LIB"]) if os.environ.get("ONSSEN_PROBE_LIB") else get_lib(); dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
M, K, N = 12800, int(os.environ.get("K", 1200)), 4800
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
KB = (K + 31) // 32
a_img = torch.empty(M, KB, 2, 32, device=dev, dtype=torch.int16)
w_img = torch.empty(N, KB, 2, 32, device=dev, dtype=torch.int16)
lib.x3_image(W.data_ptr(), K, 0, 1, N, K, w_img.data_ptr(), st)
lib.x3_image(A.data_ptr(), K, 0, 1, M, K, a_img.data_ptr(), st)
out = torch.empty(M, N, device=dev)
run = lambda: lib.linear_x3p(a_img.data_ptr(), M, K, w_img.data_ptr(), b.data_ptr(), N, 0, 0, 1e-12, out.data_ptr(), 1, N, 0, st)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"x3p K={K} lib={os.path.basename(os.environ.get('ONSSEN_PROBE_LIB','default'))} gn={os.environ.get('ONSSEN_X3_GN','4')}: {ms:.3f} ms  {2*M*K*N/ms/1e9:.0f} TF effective")

if "clk" in os.environ.get("ONSSEN_PROBE_LIB", ""):
    c = out[0, :2].cpu().tolist()
    print(f"   workgroup 0 main loop: {c[0]:.0f} shader clocks in {c[1] / 100:.1f} us -> {c[0] / c[1] * 100:.0f} MHz; {c[0] / ((K + 31) // 32):.0f} clocks per k-step")
