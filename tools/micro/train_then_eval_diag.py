#!/usr/bin/env python
"""GPU box diagnostic (round 5): the first eval-mode forward of a model AFTER training gave an untrained-looking result unless an
eval-mode forward had also run BEFORE training.  Trains twice from the same seed (flow A: train first; flow B: one eval forward
first), then compares parameters / buffers of the two flows, the HIP eval embedding against the ATen-CPU oracle on the live
state_dict, and the effect of model.repack()."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import trained_probe as P
from onssen_amd import nn as onn
from onssen_amd.features import stft_logmag
from onssen_amd.synthetic import synth_mixture
from oracle import torch_cpu as TC

dev = torch.device("cuda:0")
steps = int(os.environ.get("STEPS", 300))
wav = torch.from_numpy(synth_mixture(900_001, 64 * 250)[None]).to(dev)


def flow(eval_first):
    torch.manual_seed(0)
    m = onn.deep_clustering(129, 600, 2, 20, dropout=0.3).to(dev)
    with torch.no_grad():
        logmag, _ = stft_logmag(wav, 256, 64)
        if eval_first:
            m.eval()
            m([logmag])
    curve, _ = P.train(m, steps, dev, log=lambda s: None)
    m.eval()
    with torch.no_grad():
        e1, = m([logmag])
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        ref = TC.deep_clustering_forward(sd, logmag.cpu().numpy()).numpy()
        d1 = float(np.abs(e1.cpu().numpy() - ref).max())
        m.repack()
        e2, = m([logmag])
        d2 = float(np.abs(e2.cpu().numpy() - ref).max())
    print(f"eval_first={eval_first}: loss {curve[0][1]:.2f} -> {curve[-1][1]:.2f} | HIP vs ATen-CPU on the live state_dict: max|d| {d1:.3e}; after repack(): {d2:.3e} | "
          f"bn running_mean norm {float(m.bn.running_mean.norm()):.4f} var mean {float(m.bn.running_var.mean()):.4f} batches {int(m.bn.num_batches_tracked)} | "
          f"versions: w_ih0 {m.rnn.weight_ih_l0._version} fc {m.fc_dc.weight._version} rm {m.bn.running_mean._version}")
    return m, e1


mA, eA = flow(False)
mB, eB = flow(True)
for (n, a), (_, b) in zip(mA.state_dict().items(), mB.state_dict().items()):
    d = float((a.double() - b.double()).abs().max())
    if d > 1e-6:
        print(f"  differs between the flows: {n} max|d| {d:.3e}")
print("embedding A vs B max|d|", float((eA - eB).abs().max()))
