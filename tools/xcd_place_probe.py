#!/usr/bin/env python
"""Profiling aid (GPU box): which XCD did every workgroup of the persistent recurrence launch land on?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn
from onssen_amd.hip import get_lib
dev = torch.device("cuda:0"); lib = get_lib()
T, F = 100, 129
for (H, B) in ((600, 32), (600, 48), (600, 64), (300, 64)):
    ug = 4 * -(-H // 128)
    model = onn.deep_clustering(F, H, 2, 20).to(dev).eval()
    pk = model._packed.get(ug); Hp = pk.Hp; NU = Hp // ug
    y = torch.empty(T, B, 2, Hp, device=dev)
    nb = lib.blstm_workspace_bytes(B, T, 2 * Hp, H, 1, ug)
    ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
    yin = torch.randn(T, B, 2 * Hp, device=dev).tanh_()
    for it in range(3):
        lib.blstm_forward(yin.data_ptr(), 2 * Hp, B * 2 * Hp, B, T, 2 * Hp, H, 1, ug, [pk.wih_img[1].data_ptr()],
                          [pk.whh_x3[1].data_ptr()], [pk.bias[1].data_ptr()], y.data_ptr(), ws.data_ptr(), ws.numel(),
                          (32 << 8) | 2 | 4, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        d = ws[nb - 65536:].cpu().numpy().view(np.int64)[4096:4096 + 8 * NU]
        xcc = (d & 15).reshape(NU, 8)
        st = ws[:2048].cpu().numpy().view(np.uint32)
        print(f"H={H} B={B} NU={NU} launch {it}: safe={st[281]} abort={st[280]}")
        # per column (= group) the set of XCDs its members sit on; per row the dispatch pattern
        print("   group -> XCDs:", [sorted(set(xcc[:, g].tolist())) for g in range(8)])
        print("   first rows :", xcc[:4].tolist())
        bad = [m for m in range(NU) if (np.roll(xcc[m], -0) - xcc[0]).any()]
        print("   rows whose pattern differs from row 0:", bad[:20])
        if bad:
            print("   e.g. row", bad[0], xcc[bad[0]].tolist())
