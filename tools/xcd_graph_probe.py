#!/usr/bin/env python
"""Debug aid (GPU box): XCD-local recurrence, eager vs hipGraph replay."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn
from onssen_amd.synthetic import make_state_dict
dev = torch.device("cuda:0")
for (H, L, B, T) in ((64, 2, 4, 40), (600, 2, 32, 400)):
    sd = make_state_dict("deep_clustering", 129, H, L, 20, 2, seed=2)
    m = onn.deep_clustering(129, H, L, 20)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.to(dev).eval()
    x = torch.randn(B, T, 129, device=dev)
    with torch.no_grad():
        a = m([x])[0].clone(); b = m([x])[0].clone()
        print(f"H={H} B={B} T={T}: eager equal {torch.equal(a, b)}")
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s): m([x])
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): out = m([x])[0]
        for r in range(4):
            out.zero_(); g.replay(); torch.cuda.synchronize()
            st = [buf[1120:1128].cpu().view(torch.int32).tolist() for buf in m._ws.cache.values()]
            d = (out - a).abs().max().item()
            print(f"   replay {r}: max|out-a| {d:.3e} nan {torch.isnan(out).any().item()} status(abort,safe) {st}")
        x2 = torch.randn(B, T, 129, device=dev); x.copy_(x2)
        ref = m([x])[0].clone()
        g.replay(); torch.cuda.synchronize()
        print(f"   new input: replay vs eager max diff {(out - ref).abs().max().item():.3e}")
