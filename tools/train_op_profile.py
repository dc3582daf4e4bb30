#!/usr/bin/env python
"""GPU box: torch.profiler over a few training steps (tools/train_step_bench.py's step) -- device time per ATen op with the Python
line that issued it: where the time OUTSIDE the package's kernels goes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn
from onssen_amd.data.synthetic_wsj0_2mix import wsj0_2mix_dataloader
from onssen_amd.dist import train_step
from onssen_amd.loss import loss_dc
from onssen_amd.utils import build_optimizer
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
fo = dict(batch_size=16, frame_length=400, sampling_rate=8000, window_size=256, hop_size=64, db_threshold=40)
torch.manual_seed(0)
model = onn.deep_clustering(129, 600, 3, 20, dropout=0.3).to(dev).train()
opt = build_optimizer(model.parameters(), {"name": "adam", "lr": 1e-3})
batches = [b for _, b in zip(range(3), wsj0_2mix_dataloader("dc", fo, "tr", device=str(dev)))]
for i in range(3): train_step(model, opt, loss_dc, *batches[i % 3])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(5): train_step(model, opt, loss_dc, *batches[i % 3])
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cuda_time_total", row_limit=45, max_src_column_width=90, max_name_column_width=40))
