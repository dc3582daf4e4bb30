#!/usr/bin/env python
"""GPU box: how many co-tenant workgroups fit beside the persistent recurrences?  

A recurrence exchange group needs 30 of its XCD's 32 CUs resident at the same time; data-parallel training runs RCCL's
channel kernels (one workgroup per channel) UNDER the backward recurrence.  This probe stands in for RCCL with
``onssen_debug_cotenant_spin``: k workgroups x 256 threads that only hold their CUs, launched on a side stream BEFORE the
step so that the recurrences must fit around them, and records
  (a) the time of a training step (DC 2 x BLSTM-600, 16 chunks x 400 frames) and of the headline inference step
      (32 chunks) while a 20 ms co-tenant of k workgroups is resident: a recurrence that cannot become resident waits for
      the co-tenant to finish, so the step time jumps to >= 20 ms at the k that exceeds the budget;
  (b) whether a co-tenant that outlives the bounded wait (0.6 s > ~0.2 s) makes a launch abort, and that the step is
      recovered (dist.train_step re-runs it) -- the abort path end to end.
Output: one table, committed as profiles/r03_cotenant_probe.txt.  The budget it shows is what dist.RCCL_MAX_CHANNELS enforces."""
import os, sys, time, warnings
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn, dist as odist
from onssen_amd.hip import get_lib
from onssen_amd.loss import loss_dc
from onssen_amd.nn import _core

dev = torch.device("cuda:0")
lib = get_lib()
torch.manual_seed(0)
P = _core._XcdPolicy
side = torch.cuda.Stream()

mt = onn.deep_clustering(129, 600, 2, 20, dropout=0.0).to(dev).train()
opt = torch.optim.SGD(mt.parameters(), lr=1e-4)
xt = torch.randn(16, 400, 129, device=dev)
lab = torch.randint(0, 2, (16, 400, 129), device=dev)
label = [torch.stack([lab, 1 - lab], -1).double(), torch.rand(16, 400, 129, device=dev) + 0.1]
mi = onn.deep_clustering(129, 600, 2, 20).to(dev).eval()
xi = torch.randn(32, 400, 129, device=dev)


HEAVY = int(os.environ.get("HEAVY", "1"))      # 1: >= 112 VGPRs + 32 KB LDS per workgroup (needs a CU of its own); 0: shares CUs


def spin(k, ms):
    if k > 0:
        lib.check(lib.dll.onssen_debug_cotenant_spin(k, 256, int(ms * 100_000), HEAVY, side.cuda_stream), "spin")


def timed(fn, k, ms):
    torch.cuda.synchronize()
    a0 = P.aborts
    spin(k, ms)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), P.aborts - a0


def train():
    odist.train_step(mt, opt, loss_dc, [xt], label)


def infer():
    with torch.no_grad():
        mi([xi])
    _core._XcdStatus.flush()


for _ in range(3):
    train(); infer()
print(f"# co-tenant: k workgroups x 256 threads ({'>= 112 VGPRs + 32 KB LDS each' if HEAVY else '2 VGPRs, no LDS'}) spinning for `ms` on a side stream, launched BEFORE the step")
print("# k   train_step_ms(20ms co-tenant)  aborts   infer_step_ms(20ms co-tenant)  aborts")
rows = []
for k in (0, 8, 16, 17, 24, 32, 48, 64, 96, 128, 256):
    t_ms, t_ab = min((timed(train, k, 20.0) for _ in range(2)), key=lambda r: r[0])
    i_ms, i_ab = min((timed(infer, k, 20.0) for _ in range(2)), key=lambda r: r[0])
    print(f"{k:4d}   {t_ms:8.2f}                     {t_ab}        {i_ms:8.2f}                     {i_ab}", flush=True)
    rows.append((k, t_ms, i_ms))
base_t, base_i = rows[0][1], rows[0][2]
fit = max(k for k, t, i in rows if t < base_t * 1.5 and i < base_i * 1.5 + 0.5)
print(f"# largest probed k that does not delay either step: {fit}  (dist.RCCL_MAX_CHANNELS = {odist.RCCL_MAX_CHANNELS})")
print("# a co-tenant that outlives the bounded wait (600 ms; default limit ~0.2 s): abort + recovery")
print("# k   train_step_ms  aborts  recovered_total  persistent_again_next_step")
for k in (24, 64, 256):
    r0 = P.recovered
    t_ms, t_ab = timed(train, k, 600.0)
    torch.cuda.synchronize()
    time.sleep(0.7)                       # let the co-tenant finish
    n_p = P.persistent_launches
    train(); torch.cuda.synchronize()
    print(f"{k:4d}   {t_ms:9.1f}     {t_ab}       {P.recovered - r0}                {P.persistent_launches > n_p and P.skip == 0}", flush=True)
print("# policy state at the end:", dict(aborts=P.aborts, recovered=P.recovered, skip=P.skip, streak=P.streak,
                                        persistent=P.persistent_launches, fallback=P.fallback_launches))
