#!/usr/bin/env python
"""Profiling aid (GPU box): do two streams created with hipExtStreamCreateWithCUMask run kernels concurrently on
disjoint parts of the chip?  Tries two bit layouts for 'half of the chip'."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd.hip import get_lib
lib = get_lib(); dev = torch.device("cuda:0")
import glob
cands = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l]
hip = ctypes.CDLL(cands[0]) if cands else ctypes.CDLL("libamdhip64.so")   # the HIP runtime torch loaded
print("HIP runtime:", cands[0] if cands else "libamdhip64.so")
create = hip.hipExtStreamCreateWithCUMask
create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
create.restype = ctypes.c_int
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (w * 32 + b) in bits) for w in range(8)])
    s = ctypes.c_void_p()
    rc = create(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
M, K, N = 12800, 1200, 4800
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; bb = torch.randn(N, device=dev)
KB = (K + 31) // 32
a_img = torch.empty(M, KB, 2, 32, device=dev, dtype=torch.int16); w_img = torch.empty(N, KB, 2, 32, device=dev, dtype=torch.int16)
st = torch.cuda.current_stream().cuda_stream
lib.x3_image(A.data_ptr(), K, 0, 1, M, K, a_img.data_ptr(), st); lib.x3_image(W.data_ptr(), K, 0, 1, N, K, w_img.data_ptr(), st)
outs = [torch.empty(M, N, device=dev) for _ in range(2)]
torch.cuda.synchronize()
def gemm(i, s):
    lib.linear_x3p(a_img.data_ptr(), M, K, w_img.data_ptr(), bb.data_ptr(), N, 0, 0, 0.0, outs[i].data_ptr(), 1, N, 0, s.cuda_stream)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
d0 = torch.cuda.Stream(); d1 = torch.cuda.Stream()
print(f"plain streams: one GEMM {timeit(lambda: gemm(0, d0)):.3f} ms, two on two streams {timeit(lambda: (gemm(0, d0), gemm(1, d1))):.3f} ms")
for name, sets in (("contiguous halves (bits 0-127 / 128-255)", (set(range(128)), set(range(128, 256)))),
                   ("interleaved (bit % 8 < 4 / >= 4)", ({b for b in range(256) if b % 8 < 4}, {b for b in range(256) if b % 8 >= 4})),
                   ("even / odd bits", ({b for b in range(256) if b % 2 == 0}, {b for b in range(256) if b % 2 == 1}))):
    s0, s1 = masked_stream(sets[0]), masked_stream(sets[1])
    t1 = timeit(lambda: gemm(0, s0)); t2 = timeit(lambda: (gemm(0, s0), gemm(1, s1)))
    print(f"{name}: one GEMM on mask 0 {t1:.3f} ms, two GEMMs on the two masks {t2:.3f} ms")
