#!/usr/bin/env python
"""GPU box: a 2 x BLSTM-H deep-clustering forward (32 x 400 frames) on the persistent recurrence against the launch-per-step form,
for layer widths either side of 640 (round 4: 640 < H <= 768 on 24-unit members)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn

dev = torch.device("cuda:0")
out = {}
only = os.environ.get("WIDE_ONLY")        # e.g. WIDE_ONLY=768 under rocprofv3: that width on the persistent form only
for H in ((int(only),) if only else (600, 640, 704, 768)):
    torch.manual_seed(H)
    m = onn.deep_clustering(129, H, 2, 20).to(dev).eval()
    x = torch.randn(32, 400, 129, device=dev)
    for xcd in (("1",) if only else ("1", "0")):
        os.environ["ONSSEN_XCD"] = xcd
        with torch.no_grad():
            for _ in range(3):
                m([x])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                m([x])
            e1.record()
        torch.cuda.synchronize()
        out[f"H{H}_xcd{xcd}_ms"] = round(e0.elapsed_time(e1) / 10, 4)
os.environ["ONSSEN_XCD"] = "1"
print(json.dumps(out))
