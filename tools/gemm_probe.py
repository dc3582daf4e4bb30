#!/usr/bin/env python
"""GPU box: time onssen_linear_x3p (the x3-image GEMM) over K / M / N sweeps -- the slope in K is the cost of a k-step,
the intercept is prologue + epilogue (C stores).  SHAPES="(M,N,K) ..." overrides the default sweep."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import _abi
from onssen_amd.hip import get_lib
dev = torch.device("cuda:0"); lib = get_lib()
shapes = eval(os.environ.get("SHAPES", "[(12800,4800,k) for k in (32,128,600,1200,2400,4800)] + [(12800,2580,1200),(25600,4800,1200),(6400,4800,1200),(12800,4800,129)]"))
mode = int(os.environ.get("MODE", "0"))
for (M, N, K) in shapes:
    KB = (K + 31) // 32
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.zeros(N, device=dev)
    ai = torch.empty(M, KB, 2, 32, dtype=torch.int16, device=dev); wi = torch.empty(N, KB, 2, 32, dtype=torch.int16, device=dev)
    out = torch.empty(M, N, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    lib.x3_image(x.data_ptr(), K, 0, 1, M, K, ai.data_ptr(), st)
    lib.x3_image(w.data_ptr(), K, 0, 1, N, K, wi.data_ptr(), st)
    run = lambda: lib.linear_x3p(ai.data_ptr(), M, K, wi.data_ptr(), b.data_ptr(), N, mode, 20 if mode == 1 else 0, 1e-12, out.data_ptr(), 1, N, 0, st)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    ref = x[:64].double() @ w.double().T
    err = (out[:64].double() - ref).abs().max().item() if mode == 0 else float("nan")
    print(f"M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF algorithmic ({6.0*M*N*KB*32/ms/1e9:7.1f} TF bf16 issued)  err {err:.2e}")
