#!/usr/bin/env python
"""Profiling aid (GPU box): in-kernel timestamps of the recurrence step kernel (workgroup 0)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn
from onssen_amd.hip import get_lib
B, T, F, H, ug = 32, 400, 129, 600, 8
dev = torch.device("cuda:0"); lib = get_lib()
model = onn.deep_clustering(F, H, 2, 20).to(dev).eval()
pk = model._packed.get(ug); Hp = pk.Hp
y = torch.empty(T, B, 2, Hp, device=dev)
nb = lib.blstm_workspace_bytes(B, T, 2 * Hp, H, 1, ug)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
yin = torch.randn(T, B, 2 * Hp, device=dev).tanh_()
for x3, ab in ((2, 0), (2, 1), (2, 2), (2, 3), (2, 11), (0, 0)):
    whh = pk.whh_x3 if x3 else pk.whh
    wih = pk.wih_x3 if x3 else pk.wih
    def layer():
        lib.blstm_forward(yin.data_ptr(), 2 * Hp, B * 2 * Hp, B, T, 2 * Hp, H, 1, ug, [wih[1].data_ptr()],
                          [whh[1].data_ptr()], [pk.bias[1].data_ptr()], y.data_ptr(), ws.data_ptr(), ws.numel(),
                          ((32 | ab) << 8) | x3, torch.cuda.current_stream().cuda_stream)
    layer(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): layer()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): layer()
    g.replay(); torch.cuda.synchronize()
    d = ws[nb - 65536:].cpu().numpy().view(np.int64)[:T * 8].reshape(T, 8)[50:350].astype(np.float64)
    wall = (d[:, 6] - d[:, 0]) * 10e-3           # 100 MHz counter -> us
    gap = (d[1:, 0] - d[:-1, 6]) * 10e-3
    period = (d[1:, 0] - d[:-1, 0]) * 10e-3
    cyc = d[:, 1:6] - d[:, 1:2]
    print(f"x3={x3} ablate={ab}: step period {period.mean():.2f} us | in-kernel (WG0 thread0) entry->end {wall.mean():.2f} us | "
          f"end->next entry {gap.mean():.2f} us | cycles since entry: loads issued {cyc[:,1].mean():.0f}, "
          f"mfma done {cyc[:,2].mean():.0f}, after barrier {cyc[:,3].mean():.0f}, end {cyc[:,4].mean():.0f}")
