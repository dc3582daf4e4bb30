#!/usr/bin/env python
"""Profiling aid (GPU box, a build with -DONSSEN_XCD_PROFILE=0x83729 -DONSSEN_XCD_PROFILE_STN=32): the step timeline of the FIRST layer's persistent
recurrence with its input projection fused into the launch (ONSSEN_BLSTM_FUSE_IN0 | FUSE_TAIL) against the same layer fed
from G -- where do the ~270 clocks per step go that the fused form costs at B <= 32?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn, _abi
from onssen_amd.hip import get_lib
B, T, F, H = int(os.environ.get("B", 32)), 400, 129, 600
ug = 20
dev = torch.device("cuda:0"); lib = get_lib()
model = onn.deep_clustering(F, H, 2, 20).to(dev).eval()
pk = model._packed.get(ug); Hp = pk.Hp
y = torch.empty(T, B, 2, Hp, device=dev)
nb = lib.blstm_workspace_bytes(B, T, F, H, 1, ug)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
x = torch.randn(B, T, F, device=dev)
def layer(fused, dbgflag):
    flags = _abi.BLSTM_BF16X3 | _abi.BLSTM_XCD | (dbgflag << 8)
    wih, bias = pk.wih_img[0], pk.bias[0]
    if fused:
        flags |= _abi.BLSTM_FUSE_IN0 | _abi.BLSTM_FUSE_TAIL
        wih, bias = pk.wih_frag0, pk.bias0_tail
    lib.blstm_forward(x.data_ptr(), x.stride(0), x.stride(1), B, T, F, H, 1, ug, [wih.data_ptr()], [pk.whh_x3[0].data_ptr()],
                      [bias.data_ptr()], y.data_ptr(), ws.data_ptr(), ws.numel(), flags, torch.cuda.current_stream().cuda_stream)
for fused in (0, 1):
    for _ in range(3): layer(fused, 32)
    torch.cuda.synchronize()
    d = ws[nb - 65536:].cpu().numpy().view(np.int64)[:32 * 24].reshape(32, 24)[2:30].astype(np.float64)
    per = (d[1:, 0] - d[:-1, 0]).mean()
    m = lambda a, b: (d[:, a] - d[:, b]).mean()
    nx = lambda a: (d[1:, a] - d[:-1, 3]).mean()
    print(f"fused={fused} B={B}: cycles/step {per:.0f} | relative to the barrier: wave 0 hand-off store issued {m(19,3):.0f} | wave 1 done {m(13,3):.0f} | "
          f"last wave starts waiting for the done counter {m(12,3):.0f} | last wave requests chunks {m(9,3):.0f} | wave 0 requests chunks {m(5,3):.0f} | "
          f"next step: last wave has all chunks {nx(10):.0f} | wave 0 starts {nx(0):.0f} has all chunks {nx(8):.0f} | next barrier {nx(3):.0f}")
    if os.environ.get("CELL") == "1":      # build with -DONSSEN_XCD_PROFILE=0xF0009: the cell update of wave 0 in pieces
        print(f"   cell update of wave 0: barrier -> sums in registers {m(16,3):.0f} | gates + cell + h {m(17,16):.0f} | split + quad gather {m(18,17):.0f} | "
              f"hand-off store issued {m(19,18):.0f} | step period {per:.0f}")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): layer(fused, 0)
    e1.record(); torch.cuda.synchronize()
    print(f"fused={fused}: layer call (split + GEMM + recurrence, or recurrence alone when fused), no stamps: {e0.elapsed_time(e1)/5:.3f} ms")
