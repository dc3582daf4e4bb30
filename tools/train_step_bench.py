#!/usr/bin/env python
"""BASELINE config 4: one deep-clustering TRAINING step -- forward, loss_dc, backward, RCCL
all-reduce of the gradients, clip, Adam -- on synthetic wsj0-2mix batches whose features and labels come from the HIP
front end (STFT + label kernels).  The BLSTM stack runs on the HIP training path (saved-state forward + backward
recurrence, onssen_amd/nn/_train.py; ONSSEN_TRAIN_HIP=0 selects the stock ATen LSTM for comparison); BatchNorm, the
embedding head and loss_dc run on the package's HIP kernels behind autograd Functions, clipping and Adam on ATen.  One
JSON line like bench.py.

    python tools/train_step_bench.py [--steps K --warmup W]            # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_step_bench.py --gpus N
"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--hidden", type=int, default=600, help="BLSTM width (640 < H <= 768: persistent forward + launch-per-step backward)")
    ap.add_argument("--dropout", type=float, default=0.3, help="nn.LSTM inter-layer dropout (0 makes the HIP and ATen paths comparable step for step)")
    args = ap.parse_args()
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from onssen_amd import nn as onn
    from onssen_amd.data.synthetic_wsj0_2mix import wsj0_2mix_dataloader
    from onssen_amd.dist import train_step
    from onssen_amd.loss import loss_dc
    fo = dict(batch_size=16, frame_length=400, sampling_rate=8000, window_size=256, hop_size=64, db_threshold=40)
    torch.manual_seed(0)
    model = onn.deep_clustering(129, args.hidden, args.layers, 20, dropout=args.dropout).to(dev).train()
    from onssen_amd.utils import build_optimizer
    opt = build_optimizer(model.parameters(), {"name": "adam", "lr": 1e-3})      # egs/*/config.json: adam, 1e-3
    loader = wsj0_2mix_dataloader("dc", fo, "tr", device=str(dev))
    batches = []
    for i, b in enumerate(loader):
        batches.append(b)
        if i >= 3:
            break
    def step(i):
        inp, lab = batches[i % len(batches)]
        return train_step(model, opt, loss_dc, inp, lab, world=world)
    import contextlib
    aten = contextlib.nullcontext()
    if os.environ.get("ONSSEN_TRAIN_HIP", "1") != "1":      # the comparison rows: stock ATen / MIOpen LSTM patched in by this tool
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from aten_lstm_reference import aten_lstm_reference
        aten = aten_lstm_reference()
    aten.__enter__()
    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=dev)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    if rank == 0:
        frames = world * 16 * 400 * args.steps
        print(json.dumps({"metric": "training real_time_factor (forward + loss_dc + backward + all-reduce + clip + Adam)",
                          "value": frames * 64 / 8000 / dt, "unit": "audio-seconds trained per wall-second, whole job",
                          "frames_per_s": frames / dt, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "dtype": ("f32 as split-bf16 (bf16x3 MFMA, fp32 accumulate) in the HIP BLSTM forward/backward recurrences, the weight/input-gradient GEMMs and the heads"
                                    if os.environ.get("ONSSEN_TRAIN_HIP", "1") == "1" else "f32 (ATen / MIOpen autograd path)"),
                          "blstm_path": "hip" if os.environ.get("ONSSEN_TRAIN_HIP", "1") == "1" else "aten",
                          "data": "synthetic", "last_loss": loss,
                          "config": {"workload": f"deep_clustering {args.layers}xBLSTM-{args.hidden} training, 16 x 400-frame chunks per GPU, features + labels from the HIP front end",
                                     "parallelism": f"data parallel x{world}, RCCL all-reduce of per-layer gradient buckets before clipping"}}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
