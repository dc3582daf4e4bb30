#!/usr/bin/env python
"""Profiling aid (GPU box; run under rocprofv3 --kernel-trace --stats): the 'next'-row kernels at BASELINE sizes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn
from onssen_amd.evaluate import batch_SDR_torch
from onssen_amd.features import stft_logmag, training_labels, mask_istft
from onssen_amd.loss import loss_dc, loss_chimera_psa
from onssen_amd.separation import dc_masks
dev = torch.device("cuda:0")
B, T, F, D, NS = 32, 400, 129, 20, 25536
wav3 = torch.randn(B * 3, NS, device=dev) * 0.1
with torch.no_grad():
    for _ in range(3):
        lm, ri = stft_logmag(wav3)
        lm, ri = lm.view(B, 3, T, F), ri.view(B, 3, T, F, 2)
        out = training_labels(ri[:, 0].contiguous(), ri[:, 1].contiguous(), ri[:, 2].contiguous(), lm[:, 0].contiguous(), 40.0, with_cos=True)
        one_hot, mm, m1, m2, c1, c2 = out
        emb = torch.nn.functional.normalize(torch.randn(B, T, F, D, device=dev), dim=-1)
        masks = torch.rand(B, T, F, 2, device=dev)
        loss_dc([emb], [one_hot, mm])
        loss_chimera_psa([emb, masks[..., 0], masks[..., 1]], [one_hot, mm, m1, m2, c1, c2])
        km = dc_masks(emb, lm[:, 0].contiguous())
        sig = mask_istft(ri[:, 0].contiguous(), km, 64, NS)
        batch_SDR_torch(sig, torch.randn_like(sig))
    m = onn.enhance(F, 600, 2).to(dev).eval()
    m([lm[:, 0].contiguous(), mm])
torch.cuda.synchronize()
print("done")
