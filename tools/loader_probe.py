#!/usr/bin/env python
"""GPU box: can the wsj0-2mix FILE loader feed the training step?  (round 5)

A corpus of synthetic utterances (4-7 s, 16-bit PCM at 8 kHz, wsj0-2mix layout) is written to tmpfs; the loader's batches of
16 x 400-frame chunks (features + labels on the device, the reference's "dc" yield contract) are timed by themselves, then
feeding ``dist.train_step`` of deep_clustering 3xBLSTM-600 (the as-shipped recipe).  Target: batches/s >= 2x the training
step rate (>= 280 at a 7.1 ms step), and a fed step that costs what a step on resident batches costs.

    python tools/loader_probe.py [--utterances 320 --epochs 3]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=320)
    ap.add_argument("--epochs", type=int, default=3)
    args = ap.parse_args()
    from onssen_amd import nn as onn, options
    from onssen_amd.data import Wsj02mixFiles, write_wav
    from onssen_amd.dist import train_step
    from onssen_amd.loss import loss_dc
    from onssen_amd.synthetic import synth_mixture
    from onssen_amd.utils import build_optimizer
    dev = torch.device("cuda:0")
    root = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        for sub in ("mix", "s1", "s2"):
            os.makedirs(os.path.join(root, "wav8k", "min", "tr", sub))
        rng = np.random.default_rng(0)
        base = [synth_mixture(300 + i, 8000 * 7, 8000, return_sources=True) for i in range(16)]      # 16 distinct utterances, cut to many lengths
        secs = 0.0
        for i in range(args.utterances):
            n = int(rng.integers(4 * 8000, 7 * 8000))
            trip = base[i % 16]
            for sub, sig in zip(("mix", "s1", "s2"), trip):
                write_wav(os.path.join(root, "wav8k", "min", "tr", sub, f"u{i:04d}.wav"), sig[:n], 8000)
            secs += n / 8000.0
        fo = dict(batch_size=16, frame_length=400, sampling_rate=8000, window_size=256, hop_size=64, db_threshold=40, data_path=root)
        out = {"corpus": f"{args.utterances} utterances, {secs:.0f} s of audio, 16-bit PCM on tmpfs", "batch": "16 x 400 frames"}

        def loader_only(workers, prefetch):
            options.configure(loader_workers=workers, loader_prefetch=prefetch)
            dl = Wsj02mixFiles("dc", fo, "tr", device=dev, seed=0)
            for _ in dl:                                    # first epoch: header cache, pinned buffers, kernels warm
                pass
            torch.cuda.synchronize()
            t0, nb = time.perf_counter(), 0
            for _ in range(args.epochs):
                for inp, lab in dl:
                    nb += 1
            torch.cuda.synchronize()
            return nb / (time.perf_counter() - t0)
        out["batches_per_s"] = {f"workers={w},prefetch={p}": loader_only(w, p) for w, p in ((0, 0), (1, 3), (4, 3), (8, 3))}

        # host side alone (the producer's work: header cache + batch reads + crops), no device work
        options.configure(loader_workers=4, loader_prefetch=3)
        dl = Wsj02mixFiles("dc", fo, "tr", device=dev, seed=0)
        list(dl.host_batches())
        t0, nb = time.perf_counter(), 0
        for _ in range(args.epochs):
            for _ in dl.host_batches():
                nb += 1
        out["host_side_batches_per_s"] = nb / (time.perf_counter() - t0)

        # feeding the training step
        torch.manual_seed(0)
        model = onn.deep_clustering(129, 600, 3, 20, dropout=0.3).to(dev).train()
        opt = build_optimizer(model.parameters(), {"name": "adam", "lr": 1e-3})
        dl = Wsj02mixFiles("dc", fo, "tr", device=dev, seed=0)
        resident = [b for b in dl]
        for b in resident[:3]:
            train_step(model, opt, loss_dc, *b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.epochs):
            for b in resident:
                train_step(model, opt, loss_dc, *b)
        torch.cuda.synchronize()
        ms_res = (time.perf_counter() - t0) / (args.epochs * len(resident)) * 1e3
        t0, nb = time.perf_counter(), 0
        for _ in range(args.epochs):
            for b in dl:
                train_step(model, opt, loss_dc, *b)
                nb += 1
        torch.cuda.synchronize()
        ms_fed = (time.perf_counter() - t0) / nb * 1e3
        out["train_step_ms"] = {"resident_batches": ms_res, "fed_by_the_file_loader": ms_fed, "loader_cost_ms": ms_fed - ms_res}
        best = max(out["batches_per_s"].values())
        out["loader_vs_step_rate"] = best / (1e3 / ms_res)
        print(json.dumps(out))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
