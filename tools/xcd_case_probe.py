#!/usr/bin/env python
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn
from onssen_amd.synthetic import make_state_dict
from oracle import torch_cpu as TC
dev = torch.device("cuda:0")
for (H, L, B, T) in ((300, 3, 33, 21), (300, 3, 48, 21), (600, 4, 64, 50), (128, 2, 64, 9)):
    sd = make_state_dict("deep_clustering", 129, H, L, 20, 2, seed=3)
    m = onn.deep_clustering(129, H, L, 20)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.to(dev).eval()
    x = torch.randn(B, T, 129)
    ref = TC.deep_clustering_forward(sd, x.numpy()).numpy()
    for r in range(4):
        with torch.no_grad():
            out = m([x.to(dev)])[0].cpu().numpy()
        torch.cuda.synchronize()
        st = [buf[1120:1128].cpu().view(torch.int32).tolist() for buf in m._ws.cache.values()]
        gens = [buf[1152:1184].cpu().view(torch.int32).tolist() for buf in m._ws.cache.values()]
        err = np.abs(out - ref)
        bad_b = sorted(set(np.argwhere(err > 1e-3)[:, 0].tolist()))
        print(f"H={H} L={L} B={B} T={T} run {r}: max err {err.max():.3e} status(abort,safe) {st} gens {gens} bad batch rows {bad_b[:12]}")
