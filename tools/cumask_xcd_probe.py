#!/usr/bin/env python
"""Profiling aid (GPU box): where do the persistent recurrence's workgroups land on a CU-masked stream, and do two
masked streams run two recurrences concurrently?"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn
from onssen_amd.hip import get_lib
lib = get_lib(); dev = torch.device("cuda:0")
cands = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l]
hip = ctypes.CDLL(cands[0])
create = hip.hipExtStreamCreateWithCUMask
create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (w * 32 + b) in bits) for w in range(8)])
    s = ctypes.c_void_p(); assert create(ctypes.byref(s), 8, words) == 0
    return torch.cuda.ExternalStream(s.value)
T, F, H, B = 400, 129, 600, 32
ug = 20
model = onn.deep_clustering(F, H, 2, 20).to(dev).eval()
pk = model._packed.get(ug); Hp = pk.Hp; NU = Hp // ug
nb = lib.blstm_workspace_bytes(B, T, 2 * Hp, H, 1, ug)
def mkbuf():
    return dict(ws=torch.zeros(nb, dtype=torch.uint8, device=dev), yin=torch.randn(T, B, 2 * Hp, device=dev).tanh_())
bufs = [mkbuf(), mkbuf()]
torch.cuda.synchronize()
def layer(i, s, dbg=0):
    b = bufs[i]
    lib.blstm_forward(b["yin"].data_ptr(), 2 * Hp, B * 2 * Hp, B, T, 2 * Hp, H, 1, ug, [pk.wih_img[1].data_ptr()],
                      [pk.whh_x3[1].data_ptr()], [pk.bias[1].data_ptr()], None, b["ws"].data_ptr(), nb, (dbg << 8) | 2 | 4, s.cuda_stream)
def place(i):
    d = bufs[i]["ws"][nb - 65536:].cpu().numpy().view(np.int64)[4096:4096 + 8 * NU]
    xcc = (d & 15).reshape(NU, 8)
    st = bufs[i]["ws"][:2048].cpu().numpy().view(np.uint32)
    return [sorted(set(xcc[:, g].tolist())) for g in range(8)], int(st[280]), int(st[281])
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
d0, d1 = torch.cuda.Stream(), torch.cuda.Stream()
layer(0, d0, 32); torch.cuda.synchronize(); print("plain stream: group -> XCDs", place(0))
print(f"plain streams: one layer {timeit(lambda: layer(0, d0)):.3f} ms, two layers on two streams {timeit(lambda: (layer(0, d0), layer(1, d1))):.3f} ms")
for name, sets in (("contiguous halves", (set(range(128)), set(range(128, 256)))),
                   ("interleaved bit%8<4", ({b for b in range(256) if b % 8 < 4}, {b for b in range(256) if b % 8 >= 4}))):
    s0, s1 = masked_stream(sets[0]), masked_stream(sets[1])
    layer(0, s0, 32); layer(1, s1, 32); torch.cuda.synchronize()
    print(name, ": mask 0 group -> XCDs", place(0), "| mask 1", place(1))
    print(f"   one layer on mask 0 {timeit(lambda: layer(0, s0)):.3f} ms, two layers on the two masks {timeit(lambda: (layer(0, s0), layer(1, s1))):.3f} ms, status {place(0)[1:]}, {place(1)[1:]}")
