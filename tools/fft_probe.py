#!/usr/bin/env python
"""Profiling aid (GPU box): STFT + log-magnitude and mask-apply + iSTFT kernel times at the bench shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd.features import mask_istft, stft_logmag
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for (B, n, nfft, hop) in ((32, 25536, 256, 64), (64, 25536, 256, 64), (32, 16000, 512, 128)):
    wav = torch.randn(B, n, device=dev) * 0.1
    lm, ri = stft_logmag(wav, nfft, hop)
    T, F = lm.shape[1], lm.shape[2]
    masks = torch.rand(B, T, F, 2, device=dev)
    t1 = timeit(lambda: stft_logmag(wav, nfft, hop)); t2 = timeit(lambda: mask_istft(ri, masks, hop, n))
    by1 = (wav.numel() + lm.numel() + ri.numel()) * 4; by2 = (ri.numel() + masks.numel() + 2 * wav.numel()) * 4
    print(f"B={B} n={n} n_fft={nfft}: STFT+logmag {t1:.1f} us ({by1 / t1 / 1e6:.2f} TB/s algorithmic) | mask+iSTFT {t2:.1f} us ({by2 / t2 / 1e6:.2f} TB/s)")
