#!/usr/bin/env python
"""Profiling aid (GPU box): end-to-end separate_dc (waveform -> waveforms, K-means on the device) vs the bench step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn
from onssen_amd.separation import separate_dc, dc_masks
from onssen_amd.features import stft_logmag, mask_istft
from onssen_amd.synthetic import make_state_dict, synth_batch
dev = torch.device("cuda:0")
F, H, L, D, B, NS = 129, 600, 2, 20, 32, 25536
sd = make_state_dict("deep_clustering", F, H, L, D, 2, seed=0)
m = onn.deep_clustering(F, H, L, D); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.to(dev).eval()
wav = torch.from_numpy(np.concatenate([synth_batch(1, 8, NS, 8000)] * 4)).to(dev)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    t_all = timeit(lambda: separate_dc(m, wav))
    lm, ri = stft_logmag(wav)
    emb, = m([lm])
    t_km = timeit(lambda: dc_masks(emb, lm))
    print(f"separate_dc (STFT -> BLSTM -> embedding -> threshold + 2-means -> masks -> iSTFT), B=32 eager: {t_all:.3f} ms  "
          f"({B * 3.2 / t_all * 1e3:.0f} x RT) | K-means + masks alone {t_km:.3f} ms")
