#!/usr/bin/env python
"""Soak of the HIP training path: N optimizer steps of DC 3xBLSTM-600 (16 x 400 frames, dropout 0.3), then the
same forward/backward twice from identical state.  Checks: no persistent launch aborted or fell back to the
placement-independent protocol, losses finite, gradients bitwise identical between the two repeats."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    from onssen_amd import nn as onn
    from onssen_amd.loss import loss_dc
    from onssen_amd.nn._core import _XcdStatus
    torch.manual_seed(0)
    m = onn.deep_clustering(129, 600, 3, 20, dropout=0.3).to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    B, T, F = 16, 400, 129
    x = torch.randn(4, B, T, F, device=dev)
    lab = torch.nn.functional.one_hot(torch.randint(0, 2, (4, B, T, F), device=dev), 2).float()
    wt = torch.rand(4, B, T, F, device=dev)
    losses = []
    for i in range(args.steps):
        opt.zero_grad()
        loss = torch.mean(loss_dc(m([x[i % 4]]), [lab[i % 4], wt[i % 4]]))
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
        opt.step()
        if i % 50 == 0 or i == args.steps - 1:
            losses.append(float(loss.item()))
    torch.cuda.synchronize()
    _XcdStatus.poll(wait=True)           # raises on an aborted launch
    grads = []
    for rep in range(2):
        torch.manual_seed(123)           # same dropout masks
        opt.zero_grad()
        torch.mean(loss_dc(m([x[0]]), [lab[0], wt[0]])).backward()
        grads.append([p.grad.clone() for p in m.parameters()])
    torch.cuda.synchronize()
    _XcdStatus.poll(wait=True)
    same = all(torch.equal(a, b) for a, b in zip(*grads))
    finite = all(l == l and abs(l) < 1e30 for l in losses)
    print(json.dumps({"steps": args.steps, "losses_every_50": losses, "finite": finite, "repeat_bitwise_equal": same,
                      "placement_independent_protocol_seen": _XcdStatus.safe_protocol_seen}))
    assert finite and same


if __name__ == "__main__":
    main()
