#!/usr/bin/env python
"""GPU box: the two HBM-bound kernels of the step by themselves, at the headline shape
(32 utterances x 400 frames, STFT 256 / 64, two speakers) and at config 5's (512 / 128, 32 x 1 s at 16 kHz).
Each kernel is captured 8 times back to back in one hipGraph and replayed between events; the algorithmic bytes are
SURVEY 8(d)'s (samples in, log-magnitude + (re, im) out; (re, im) + two masks in, two waveforms out).
With a library built with -DONSSEN_DEBUG_KNOBS the launch shapes can be varied from the environment
(ONSSEN_STFT_PPW = pairs of frames per wave, ONSSEN_ISTFT_FB = frames per workgroup of the inverse transform)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd.features import stft_logmag, mask_istft


def timed(fn, reps=10, inner=8):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (reps * inner) * 1e-3)
    return best


dev = torch.device("cuda:0")
torch.manual_seed(0)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("ONSSEN_STFT_") or k.startswith("ONSSEN_ISTFT_"))
for name, B, T, n_fft, hop in (("headline 256/64", 32, 400, 256, 64), ("cfg5 512/128", 32, 126, 512, 128), ("B=1 256/64", 1, 1000, 256, 64)):
    n = (T - 1) * hop
    F = n_fft // 2 + 1
    wav = torch.randn(B, n, device=dev) * 0.1
    lm, ri = stft_logmag(wav, n_fft, hop)
    assert lm.shape == (B, T, F), lm.shape
    masks = torch.rand(B, T, F, 2, device=dev)
    t_s = timed(lambda: stft_logmag(wav, n_fft, hop))
    t_i = timed(lambda: mask_istft(ri, masks, hop, n))
    by_s = B * n * 4 + B * T * F * 12
    by_i = B * T * F * 16 + B * 2 * n * 4
    print(f"{name:16s} {tag:28s} stft {t_s * 1e6:7.2f} us {by_s / t_s / 1e12:5.2f} TB/s | mask_istft {t_i * 1e6:7.2f} us {by_i / t_i / 1e12:5.2f} TB/s", flush=True)
