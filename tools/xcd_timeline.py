#!/usr/bin/env python
"""Profiling aid (GPU box): in-kernel timestamps of the XCD-local persistent recurrence (workgroup 0)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn
from onssen_amd.hip import get_lib
B, T, F, H = int(os.environ.get("B", 32)), 400, 129, 600
ug = 4 * -(-H // 128)
dev = torch.device("cuda:0"); lib = get_lib()
model = onn.deep_clustering(F, H, 2, 20).to(dev).eval()
pk = model._packed.get(ug); Hp = pk.Hp
y = torch.empty(T, B, 2, Hp, device=dev)
nb = lib.blstm_workspace_bytes(B, T, 2 * Hp, H, 1, ug)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
yin = torch.randn(T, B, 2 * Hp, device=dev).tanh_()
def layer(dbgflag):
    lib.blstm_forward(yin.data_ptr(), 2 * Hp, B * 2 * Hp, B, T, 2 * Hp, H, 1, ug, [pk.wih_img[1].data_ptr()],
                      [pk.whh_x3[1].data_ptr()], [pk.bias[1].data_ptr()], y.data_ptr(), ws.data_ptr(), ws.numel(),
                      (dbgflag << 8) | 2 | 4, torch.cuda.current_stream().cuda_stream)
AB = int(os.environ.get('AB', 0))
for _ in range(3): layer(32 | AB)
torch.cuda.synchronize()
d = ws[nb - 65536:].cpu().numpy().view(np.int64)[:T * 8].reshape(T, 8)[50:350].astype(np.float64)
per = (d[1:, 0] - d[:-1, 0]).mean()
seg = [(d[:, i + 1] - d[:, i]).mean() for i in range(5)]
st = ws[:2048].cpu().numpy().view(np.uint32)
print(f"   seg1 detail: poll-done -> h loads issued {(d[:,6]-d[:,1]).mean():.0f} | G prefetch issued {(d[:,7]-d[:,6]).mean():.0f} | -> MFMA+LDS write done {(d[:,2]-d[:,7]).mean():.0f}")
print(f"ablate={AB} B={B} ug={ug}: cycles/step {per:.0f} (~{per/2.3e3:.2f} us) | G-load+poll {seg[0]:.0f} | h load+MFMA {seg[1]:.0f} | reduce barrier {seg[2]:.0f} | "
      f"epilogue+stores {seg[3]:.0f} | drain+barrier {seg[4]:.0f} | abort={st[280]} safe={st[281]}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): layer(AB)
e1.record(); torch.cuda.synchronize()
print(f"ablate={AB}: layer (GEMM ~0.6 ms + persistent recurrence), no stamps: {e0.elapsed_time(e1)/5:.3f} ms")
