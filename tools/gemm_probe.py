#!/usr/bin/env python
"""Profiling aid (GPU box): split-bf16 GEMM (layer-1 input projection shape) timing; ONSSEN_X3_ABLATE /
ONSSEN_X3_WM are read by the library at first call, so run one process per variant."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd.hip import get_lib
lib = get_lib(); dev = torch.device("cuda:0")
M, K, N = 12800, 1200, 4800
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
ld = (K + 31) // 32 * 32
planes = torch.empty(2, N, ld, device=dev, dtype=torch.int16)
st = torch.cuda.current_stream().cuda_stream
lib.linear_pack_bf16x3(W.data_ptr(), N, K, K, ld, planes.data_ptr(), st)
out = torch.empty(M, N, device=dev)
def run():
    lib.linear_bf16x3(A.data_ptr(), K, 0, 1, M, K, planes.data_ptr(), ld, b.data_ptr(), N, 0, 0, 0.0, None, out.data_ptr(), N, 0, st)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"ablate={os.environ.get('ONSSEN_X3_ABLATE','0')} wm={os.environ.get('ONSSEN_X3_WM','2')}: {ms:.3f} ms  {2*M*K*N/ms/1e9:.0f} TF effective")
