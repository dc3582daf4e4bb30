#!/usr/bin/env python
"""GPU box: the evaluation loop the reference actually runs (onssen/utils/test.py:29-41 -- whole utterances, batch 1):
tester_dc.eval() / tester_chimera.eval() over the synthetic evaluation loader, one utterance per forward like upstream
(batch=1) and K utterances of different lengths per forward (batch=K, round 4: same SDRs bit for bit).  Reports wall time
per utterance and audio seconds per wall second."""
import json, os, sys, time
import torch
os.environ.setdefault("ONSSEN_SYNTHETIC_DATA", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn
from onssen_amd.data import wsj0_2mix_dataloader
from onssen_amd.evaluate import tester_chimera, tester_dc
from onssen_amd.utils import AttrDict


def load(name):
    with open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", name)) as f:
        return AttrDict(json.load(f))


for cfg, cls, tcls in (("config_dc.json", "deep_clustering", tester_dc), ("config_chimera_psa.json", "chimera", tester_chimera)):
    args = load(cfg)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)                           # (random-init weights: the SI-SDR printed is only a fingerprint of the run)
    args.model = getattr(onn, cls)(**args["model_options"]).to(dev)
    args.checkpoint_path = None
    args.test_loader = list(wsj0_2mix_dataloader(args.model_name, args.feature_options, "tt", dev)) * 4    # resident: the loop itself is what is timed
    t = tcls(args)
    secs = sum(float(lab[2].shape[-1]) / 8000.0 * lab[2].shape[0] for _, lab in args.test_loader)
    n = sum(1 for _ in args.test_loader)
    for batch, kw in ((1, {}), (8, {}), (16, {}), (32, {}), (8, dict(pipeline=False)), (16, dict(pipeline=False)), (16, dict(bucket=4)),
                      (16, dict(bucket=4, pipeline=False))):
        if kw.get("pipeline") is False and tcls is not tester_dc:
            continue                               # (round 6c: tester_dc pipelines batch = 2 .. 16 over consecutive forwards by default)
        if tcls is not tester_dc:
            kw = {k: v for k, v in kw.items() if k != "pipeline"}
        t.eval(batch=batch, **kw)                  # warm-up: weight packing, workspaces
        torch.cuda.synchronize()
        dt = 1e9
        for _ in range(3):                         # best of three: the loop is short, a host hiccup is a large fraction of it
            t0 = time.perf_counter()
            sdr = t.eval(batch=batch, **kw)
            torch.cuda.synchronize()
            dt = min(dt, time.perf_counter() - t0)
        print(f"{cls:16s} batch {batch:2d} {str(kw):36s}: {n} utterances, {secs:.1f} s of audio: eval() {dt * 1e3:.1f} ms = {dt / n * 1e3:.3f} ms per utterance = "
              f"{secs / dt:.0f} x real time (SI-SDR {sdr:.4f})", flush=True)
