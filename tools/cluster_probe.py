#!/usr/bin/env python
"""GPU box: the deep-clustering back end by itself (onssen_dc_cluster_f32, B = 32 chunks of the headline workload): time per
call (persistent Lloyd launch vs launch per iteration), active-bin fraction, Lloyd iterations actually run.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from onssen_amd.features import stft_logmag
from onssen_amd.hip import get_lib
from onssen_amd.separation import dc_masks, _CLUSTER_WS

dev = torch.device("cuda:0")
lib = get_lib()
B = int(os.environ.get("B", 32))
with torch.no_grad():
    wl = bench.build_workload("dc_l2", B, dev)
    logmag, ri = stft_logmag(wl["wav"], 256, 64)
    emb, = wl["model"]([logmag])
    torch.cuda.synchronize()
    for form in ("0", "1"):
        os.environ["ONSSEN_DC_PERSISTENT"] = form
        for _ in range(3):
            m = dc_masks(emb, logmag)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            m = dc_masks(emb, logmag)
        e1.record()
        torch.cuda.synchronize()
        print(f"persistent={form}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per call (eager, B={B})")
    ws = next(iter(_CLUSTER_WS.values()))
    T, F, D = emb.shape[1:]
    so = int(lib.dll.onssen_dc_cluster_status_offset(B, D))
    iw = ws[so - B * 72 * 4:so].view(torch.int32).view(B, 72).cpu().numpy()
    print("active fraction per utterance:", np.round(iw[:, 64] / (T * F), 3).tolist())
    print("Lloyd generation word (iterations | 0x10000 converged):", [hex(int(v)) for v in iw[:, 66]])
    # the persistent launch by the number of Lloyd iterations: intercept = start-up + loading the rows + pass 0, slope = one pass
    os.environ["ONSSEN_DC_PERSISTENT"] = "1"
    for iters in (0, 1, 5, 10, 20):
        for _ in range(3):
            dc_masks(emb, logmag, iters=iters, tol=0.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dc_masks(emb, logmag, iters=iters, tol=0.0)
        e1.record()
        torch.cuda.synchronize()
        print(f"iters={iters:2d}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per call")
