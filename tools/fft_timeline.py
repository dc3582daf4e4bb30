#!/usr/bin/env python
"""GPU box, profile build (-DONSSEN_FFT_PROFILE, loaded through ONSSEN_HIP_LIB): wave clocks at the phases of
stft_logmag_kernel<256> at the headline shape -- where do a wave's cycles go, and when do the waves start and end?"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd.features import stft_logmag
from onssen_amd.hip import get_lib

lib = get_lib()
dev = torch.device("cuda:0")
B, T = 32, 400
wav = torch.randn(B, (T - 1) * 64, device=dev) * 0.1
for _ in range(3):
    stft_logmag(wav, 256, 64)
torch.cuda.synchronize()
buf = np.zeros(4096 * 16, dtype=np.int64)
fn = lib.dll.onssen_debug_fft_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf.ctypes.data, buf.size) == 0
s = buf.reshape(4096, 16)
live = s[:, 0] != 0
s = s[live]
print("waves stamped:", len(s))
t0 = s[:, 0].min()
names = ["start", "constants built", "first fetch issued"] + [f"pair {i // 3}: {('LDS written', 'FFT done', 'outputs issued')[i % 3]}" for i in range(12)]
last = np.where(s != 0, s, 0).max(1)
print(f"wave start  (rel. to first): mean {np.mean(s[:, 0] - t0):9.0f}  p50 {np.median(s[:, 0] - t0):9.0f}  max {np.max(s[:, 0] - t0):9.0f}")
print(f"wave end    (rel. to first): mean {np.mean(last - t0):9.0f}  p50 {np.median(last - t0):9.0f}  max {np.max(last - t0):9.0f}")
for k in range(1, 15):
    ok = (s[:, k] != 0) & (s[:, k - 1] != 0)
    if ok.sum() == 0:
        continue
    d = s[ok, k] - s[ok, k - 1]
    print(f"{names[k]:28s} +{d.mean():8.0f} ticks (p10 {np.percentile(d, 10):7.0f}, p90 {np.percentile(d, 90):7.0f}; {ok.sum()} waves)")
