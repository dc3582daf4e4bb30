#!/usr/bin/env python
"""Profiling aid (GPU box): in-kernel timestamps of the XCD-local persistent recurrence (workgroup 0, steps 64..127,
collected in the LDS and written out at the end of the launch: [step][16]; wave 0 -> slots 0..7, polling wave -> 8, 9)."""
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
STN = int(os.environ.get('STN', 64))      # stamped steps of the build (-DONSSEN_XCD_PROFILE_STN, default 64)
d = ws[nb - 65536:].cpu().numpy().view(np.int64)[:STN * 24].reshape(STN, 24)[2:STN - 2].astype(np.float64)
per = (d[1:, 0] - d[:-1, 0]).mean()
m = lambda a, b: (d[:, a] - d[:, b]).mean()
st = ws[:2048].cpu().numpy().view(np.uint32)
print(f"ablate={AB} B={B} ug={ug} waves={os.environ.get('ONSSEN_XCD_WAVES', 8)}: cycles/step {per:.0f} | wave 0: step start->all chunks complete {m(8,0):.0f} | "
      f"MFMA {m(7,8):.0f} | partials written {m(2,7):.0f} | barrier {m(3,2):.0f} | G prefetch + cell update + hand-off stores issued {m(4,3):.0f} | "
      f"output stores + pause {m(6,4):.0f} | next chunks requested {m(5,6):.0f} | to next step {(d[1:, 0] - d[:-1, 5]).mean():.0f} | abort={st[280]} safe={st[281]} nonfinite={st[282]}"
      f" || cell update of wave 0: barrier -> sums in registers {m(16,3):.0f} | gates + cell + h {m(17,16):.0f} | split + quad gather {m(18,17):.0f} | hand-off store issued {m(19,18):.0f} | done counter {m(4,19):.0f}"
      f" || passes per step: wave 0 {d[:,15].mean():.2f}, last wave {d[:,14].mean():.2f}"
      f" || last wave: chunks requested {(d[:-1, 9] - d[:-1, 3]).mean():.0f} after the barrier | all chunks complete {(d[1:, 10] - d[:-1, 9]).mean():.0f} | MFMA done {m(11,10):.0f} | wave 0's MFMA done {m(7,11):.0f} later")
# sparse view (works with a build that stamps only slots 0,3,5,8,9,10,12,13,19: -DONSSEN_XCD_PROFILE=0x83729): everything relative
# to wave 0 leaving the barrier
nx = lambda a: (d[1:, a] - d[:-1, 3]).mean()
print(f"relative to the barrier (wave 0 leaves it at 0): wave 0 hand-off store issued {m(19,3):.0f} | wave 1 done {m(13,3):.0f} | last wave starts "
      f"waiting for the done counter {m(12,3):.0f} | last wave requests chunks {m(9,3):.0f} | wave 0 requests chunks {m(5,3):.0f} | next step: "
      f"last wave has all chunks {nx(10):.0f} | wave 0 starts {nx(0):.0f} has all chunks {nx(8):.0f} | next barrier {nx(3):.0f} | wave 2 done {m(1,3):.0f} | "
      f"passes per step: wave 0 {d[:,15].mean():.2f}, last wave {d[:,14].mean():.2f} | store acknowledged (experiment builds) {m(20,19):.0f} after issue")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): layer(AB)
e1.record(); torch.cuda.synchronize()
print(f"ablate={AB}: layer (GEMM + persistent recurrence), no stamps: {e0.elapsed_time(e1)/5:.3f} ms")
