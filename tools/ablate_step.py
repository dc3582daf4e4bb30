#!/usr/bin/env python
"""Profiling aid (GPU box): time the recurrence of one layer (T=400 dependent launches, hipGraph replay)
for several unit-group sizes and with parts of the kernel switched off (ABI flag bits 8..11)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import _abi, nn as onn  # noqa: E402
from onssen_amd.hip import get_lib  # noqa: E402
from onssen_amd.nn._core import _stream  # noqa: E402

B, T, F, H = int(os.environ.get("B", 32)), 400, 129, 600
dev = torch.device("cuda:0")
lib = get_lib()
model = onn.deep_clustering(F, H, 2, 20).to(dev).eval()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for ug in (8, 12, 16):
    pk = model._packed.get(ug)
    Hp, NP = pk.Hp, pk.NP
    y = torch.empty(T, B, 2, Hp, device=dev)
    ws = torch.zeros(lib.blstm_workspace_bytes(B, T, 2 * Hp, H, 1, ug), dtype=torch.uint8, device=dev)
    yin = torch.randn(T, B, 2 * Hp, device=dev).tanh_()
    gbuf = ws[_abi.BLSTM_WS_HEADER:]

    def gemm():
        lib.linear(yin.data_ptr(), B * 2 * Hp, 2 * Hp, B, T * B, 2 * Hp, pk.wih[1].data_ptr(), 2 * Hp,
                   pk.bias[1].data_ptr(), 2 * NP, 0, 0, 0.0, None, gbuf.data_ptr(), B * 2 * NP, 2 * NP, _stream())
    tg = timed(gemm)
    row = []
    for ab in (0, 16, 3, 4, 7, 15, 0x10000):
        def layer():
            lib.blstm_forward(yin.data_ptr(), 2 * Hp, B * 2 * Hp, B, T, 2 * Hp, H, 1, ug, [pk.wih[1].data_ptr()],
                              [pk.whh[1].data_ptr()], [pk.bias[1].data_ptr()], y.data_ptr(), ws.data_ptr(), ws.numel(),
                              (ab << 8) if ab < 0x10000 else 1, _stream())
        row.append((timed(layer) - tg) / T * 1e6)
    x3 = []
    for ab in (0, 3, 4):
        def layer3():
            lib.blstm_forward(yin.data_ptr(), 2 * Hp, B * 2 * Hp, B, T, 2 * Hp, H, 1, ug, [pk.wih_x3[1].data_ptr()],
                              [pk.whh_x3[1].data_ptr()], [pk.bias[1].data_ptr()], y.data_ptr(), ws.data_ptr(),
                              ws.numel(), (ab << 8) | 2, _stream())
        x3.append((timed(layer3) - tg) / T * 1e6)
    print(f"   split-bf16 recurrence ug={ug}: us/step full={x3[0]:.2f} no_loads={x3[1]:.2f} no_mfma={x3[2]:.2f}")
    print(f"B={B} ug={ug:2d} WGs={2 * (Hp // ug) * ((B + 31) // 32 if B > 16 else 1)}: us/step full={row[0]:.2f} libm_act={row[1]:.2f} "
          f"no_loads={row[2]:.2f} no_mfma={row[3]:.2f} no_loads_mfma={row[4]:.2f} +no_G={row[5]:.2f} split_rows={row[6]:.2f} (gemm {tg * 1e3:.2f} ms)")
