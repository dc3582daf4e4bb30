#!/usr/bin/env python
"""Calibration (GPU box): cost of a dependent launch boundary on this box, eager and under hipGraph replay."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd.hip import get_lib
lib = get_lib()
dev = torch.device("cuda:0")
buf = torch.zeros(64, device=dev)
N = 400
for wgs in (1, 150, 256, 1024):
    def chain():
        lib.dll.onssen_debug_launch_chain(buf.data_ptr(), N, wgs, torch.cuda.current_stream().cuda_stream)
    chain(); torch.cuda.synchronize()
    t0 = time.perf_counter(); chain(); torch.cuda.synchronize(); te = (time.perf_counter() - t0) / N * 1e6
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"probe chain: {wgs:5d} WGs x 256 thr: eager {te:.2f} us/launch (host wall), graph replay {e0.elapsed_time(e1) / 5 / N * 1e3:.2f} us/launch")
print("hip runtime:", torch.version.hip)
