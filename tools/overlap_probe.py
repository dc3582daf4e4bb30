#!/usr/bin/env python
"""Profiling aid (GPU box): do independent recurrence chains on different streams overlap inside one
hipGraph?  Times one layer (T=400) for a batch of 32 as 1 chain, 2 chains of 16 and 4 chains of 8."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn  # noqa: E402
from onssen_amd.hip import get_lib  # noqa: E402

T, F, H, ug = 400, 129, 600, 8
dev = torch.device("cuda:0")
lib = get_lib()
model = onn.deep_clustering(F, H, 2, 20).to(dev).eval()
pk = model._packed.get(ug)
Hp, NP = pk.Hp, pk.NP


def make_chain(B, ablate=0):
    y = torch.empty(T, B, 2, Hp, device=dev)
    ws = torch.zeros(lib.blstm_workspace_bytes(B, T, 2 * Hp, H, 1, ug), dtype=torch.uint8, device=dev)
    yin = torch.randn(T, B, 2 * Hp, device=dev).tanh_()

    def run():
        lib.blstm_forward(yin.data_ptr(), 2 * Hp, B * 2 * Hp, B, T, 2 * Hp, H, 1, ug, [pk.wih[1].data_ptr()],
                          [pk.whh[1].data_ptr()], [pk.bias[1].data_ptr()], y.data_ptr(), ws.data_ptr(), ws.numel(),
                          ablate << 8, torch.cuda.current_stream().cuda_stream)
    return run


def timed(chains, reps=5):
    side = [torch.cuda.Stream() for _ in chains[1:]]

    def body():
        main = torch.cuda.current_stream()
        for s in side:
            s.wait_stream(main)
        chains[0]()
        for s, c in zip(side, chains[1:]):
            with torch.cuda.stream(s):
                c()
        for s in side:
            main.wait_stream(s)
    body(); torch.cuda.synchronize()
    cs = torch.cuda.Stream(); cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        body()
    torch.cuda.current_stream().wait_stream(cs)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for ab in (0, 7, 15):
    t1 = timed([make_chain(32, ab)])
    t2 = timed([make_chain(16, ab), make_chain(16, ab)])
    t4 = timed([make_chain(8, ab) for _ in range(4)])
    t16 = timed([make_chain(16, ab)])
    print(f"ablate={ab:2d}: layer(GEMM+400 steps) ms: 1x32={t1:.3f}  2x16={t2:.3f}  4x8={t4:.3f}  (1x16 alone={t16:.3f})")
