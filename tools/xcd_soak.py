#!/usr/bin/env python
"""Soak test (GPU box): thousands of hipGraph replays of the XCD-local recurrence with changing inputs; every replay
must be bit-equal to an eager run of the same input, the abort word must stay 0."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn
from onssen_amd.synthetic import make_state_dict
dev = torch.device("cuda:0")
N = int(os.environ.get("N", 1500))
for kind, H, L, B, T in (("deep_clustering", 600, 2, 32, 400), ("chimera", 600, 4, 64, 400), ("deep_clustering", 300, 3, 33, 50),
                         ("deep_clustering", 768, 2, 32, 400)):          # (the last: 24-unit members, every CU of the XCDs)
    sd = make_state_dict(kind, 129, H, L, 20, 2, seed=2)
    m = getattr(onn, kind)(129, H, L, 20)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.to(dev).eval()
    xs = [torch.randn(B, T, 129, device=dev) for _ in range(4)]
    x = xs[0].clone()
    with torch.no_grad():
        refs = []
        for xi in xs:
            x.copy_(xi); refs.append(m([x])[0].clone())
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s): m([x])
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): out = m([x])[0]
    bad = 0; t0 = time.time()
    for r in range(N):
        i = r % 4
        x.copy_(xs[i]); g.replay()
        if r % 50 == 49 or r == N - 1:           # compare a sample of replays (a compare per replay would serialise everything)
            torch.cuda.synchronize()
            if not torch.equal(out, refs[i]): bad += 1
    torch.cuda.synchronize()
    st = [buf[1120:1132].cpu().view(torch.int32).tolist() for buf in m._ws.cache.values()]
    print(f"{kind} H={H} L={L} B={B} T={T}: {N} replays in {time.time() - t0:.1f} s, mismatching samples {bad}, status(abort,safe,nonfinite) {st}")
