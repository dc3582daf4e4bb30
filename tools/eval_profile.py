#!/usr/bin/env python
"""GPU box: cProfile of tester_dc.eval() over resident utterances -- where does the host time of the evaluation loop go?"""
import cProfile, io, json, os, pstats, sys
import torch
os.environ.setdefault("ONSSEN_SYNTHETIC_DATA", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn
from onssen_amd.data import wsj0_2mix_dataloader
from onssen_amd.evaluate import tester_dc
from onssen_amd.utils import AttrDict

with open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "config_dc.json")) as f:
    args = AttrDict(json.load(f))
dev = torch.device("cuda:0")
args.model = onn.deep_clustering(**args["model_options"]).to(dev)
args.checkpoint_path = None
args.test_loader = list(wsj0_2mix_dataloader(args.model_name, args.feature_options, "tt", dev)) * 4
t = tester_dc(args)
K = int(os.environ.get("K", "16"))
t.eval(batch=K); t.eval(batch=K)
torch.cuda.synchronize()
# device time of the same loop: events around it (the host queues ahead)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import time
t0 = time.perf_counter(); e0.record(); t.eval(batch=K); e1.record(); torch.cuda.synchronize()
print(f"eval(batch={K}): wall {1e3 * (time.perf_counter() - t0):.2f} ms, events {e0.elapsed_time(e1):.2f} ms for {len(args.test_loader)} utterances")
pr = cProfile.Profile()
pr.enable()
t.eval(batch=K)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
