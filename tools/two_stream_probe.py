#!/usr/bin/env python
"""GPU box: two models on two streams of one process, with and without the stream-ordered serialisation of persistent launches
(nn/_core._XcdSerial): aborted launches, wall time."""
import json, os, sys, time, warnings
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn
from onssen_amd.nn import _core
from onssen_amd.nn._core import XcdAborted, _XcdPolicy, _XcdSerial, _XcdStatus

dev = torch.device("cuda:0")
torch.manual_seed(0)
ma = onn.deep_clustering(129, 600, 2, 20).to(dev).eval()
mb = onn.deep_clustering(129, 300, 3, 20).to(dev).eval()
xa, xb = torch.randn(8, 400, 129, device=dev), torch.randn(20, 200, 129, device=dev)
with torch.no_grad():
    ma([xa]); mb([xb])
torch.cuda.synchronize(); _XcdStatus.flush()
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
out = {}
orig_before = _XcdSerial.before
for mode in ("serialised", "unserialised"):
    _XcdSerial.before = orig_before if mode == "serialised" else classmethod(lambda cls, device: None)
    a0 = _XcdPolicy.aborts
    t0 = time.perf_counter()
    n_err = 0
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(20):
            for m, x, s in ((ma, xa, sa), (mb, xb, sb)):
                try:
                    with torch.cuda.stream(s):
                        m([x])
                except XcdAborted:
                    n_err += 1
    torch.cuda.synchronize()
    try:
        _XcdStatus.flush()
    except XcdAborted:
        n_err += 1
    out[mode] = {"aborted_launches": _XcdPolicy.aborts - a0, "raised": n_err, "wall_ms": round(1e3 * (time.perf_counter() - t0), 1)}
    _XcdPolicy.skip = 0; _XcdPolicy.streak = 0
print(json.dumps(out))
