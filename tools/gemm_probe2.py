#!/usr/bin/env python
"""Profiling aid (GPU box): split-bf16 GEMM over pre-split operands (onssen_linear_x3p) vs the on-the-fly form."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd.hip import get_lib
lib = get_lib(); dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, K, N, mode, group, name) in ((12800, 1200, 4800, 0, 0, "input_proj_l1"), (12800, 129, 4800, 0, 0, "input_proj_l0"),
                                     (12800, 1200, 2580, 1, 20, "fc_dc_l2norm"), (25600, 1200, 4800, 0, 0, "input_proj_l1 B=64"),
                                     (25600, 1200, 258, 2, 0, "mask head B=64")):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    KB = (K + 31) // 32; ld = KB * 32
    planes = torch.empty(2, N, ld, device=dev, dtype=torch.int16)
    lib.linear_pack_bf16x3(W.data_ptr(), N, K, K, ld, planes.data_ptr(), st)
    a_img = torch.empty(M, KB, 2, 32, device=dev, dtype=torch.int16)
    w_img = torch.empty(N, KB, 2, 32, device=dev, dtype=torch.int16)
    lib.x3_image(W.data_ptr(), K, 0, 1, N, K, w_img.data_ptr(), st)
    out = torch.empty(M, N, device=dev); out2 = torch.empty(M, N, device=dev)
    t_img = timeit(lambda: lib.x3_image(A.data_ptr(), K, 0, 1, M, K, a_img.data_ptr(), st))
    t_old = timeit(lambda: lib.linear_bf16x3(A.data_ptr(), K, 0, 1, M, K, planes.data_ptr(), ld, b.data_ptr(), N, mode, group, 1e-12, None, out.data_ptr(), N, 0, st))
    t_new = timeit(lambda: lib.linear_x3p(a_img.data_ptr(), M, K, w_img.data_ptr(), b.data_ptr(), N, mode, group, 1e-12, out2.data_ptr(), 1, N, 0, st))
    d = (out - out2).abs().max().item()
    print(f"{name:22s} M={M} K={K} N={N}: on-the-fly {t_old:.3f} ms ({2*M*K*N/t_old/1e9:.0f} TF) | pre-split {t_new:.3f} ms ({2*M*K*N/t_new/1e9:.0f} TF) "
          f"| image pass {t_img:.3f} ms | max diff {d:.2e}")
