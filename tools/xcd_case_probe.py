#!/usr/bin/env python
"""Debug aid (GPU box): one BLSTM layer through the XCD-local persistent kernel vs the launch-per-step kernel; prints where
the outputs differ."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd import nn as onn, _abi
from onssen_amd.hip import get_lib
dev = torch.device("cuda:0"); lib = get_lib()
CASES = eval(os.environ.get("CASES", "((32, 2, 16, 129), (64, 16, 50, 129), (600, 32, 40, 129))"))
ABL = int(os.environ.get("ABL", "0"))
for (H, B, T, F) in CASES:
    ug = 4 * -(-H // 128)
    torch.manual_seed(0)
    model = onn.deep_clustering(F, H, 1, 20).to(dev).eval()
    pk = model._packed.get(ug); Hp = pk.Hp
    x = torch.randn(B, T, F, device=dev)
    outs = []
    for xcd in (0, 1):
        y = torch.full((T, B, 2, Hp), float("nan"), device=dev)
        nb = lib.blstm_workspace_bytes(B, T, F, H, 1, ug)
        ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
        flags = _abi.BLSTM_BF16X3 | (_abi.BLSTM_XCD if xcd else 0) | ((ABL << 8) if xcd else 0)
        wih = pk.wih_img if xcd else pk.wih_x3
        lib.blstm_forward(x.data_ptr(), x.stride(0), x.stride(1), B, T, F, H, 1, ug, [wih[0].data_ptr()], [pk.whh_x3[0].data_ptr()],
                          [pk.bias[0].data_ptr()], y.data_ptr(), ws.data_ptr(), ws.numel(), flags, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append(y.cpu().numpy())
        st = ws[1120:1136].cpu().view(torch.int32).tolist()
    d = np.abs(outs[1] - outs[0]); d[np.isnan(d)] = 9.0
    bad = np.argwhere(d > 1e-4)
    print(f"H={H} B={B} T={T}: max diff {d.max():.3e} status {st} bad count {len(bad)} of {d.size}")
    if len(bad):
        for ax, name in enumerate(("t", "b", "dir", "unit")):
            print("   ", name, sorted(set(bad[:, ax].tolist()))[:40])
        t0 = bad[:, 0].min()
        print("    first bad t:", t0, "entries at that t:", bad[bad[:, 0] == t0][:10].tolist())
        i = tuple(bad[0]); print("    e.g.", i, outs[0][i], outs[1][i])
    if len(bad) and os.environ.get("DIAG"):
        # which part of h_0 did the bad elements of t = 1 (direction 0) see?  LSTM step restated on the host
        sd = {k: v.detach().cpu().double() for k, v in model.state_dict().items()}
        Wih, Whh = sd["rnn.weight_ih_l0"], sd["rnn.weight_hh_l0"]
        bias = sd["rnn.bias_ih_l0"] + sd["rnn.bias_hh_l0"]
        xs = x.cpu().double()
        def step(xt, h, c):
            g = xt @ Wih.T + h @ Whh.T + bias
            i, f, gg, o = g.chunk(4, -1)
            c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            return torch.sigmoid(o) * torch.tanh(c2), c2
        z = torch.zeros(B, H, dtype=torch.double)
        h0, c0 = step(xs[:, 0], z, z)
        variants = {"full": h0, "none": z}
        for lo in range(0, H, 32):
            hv = torch.zeros_like(h0); hv[:, lo:lo + 32] = h0[:, lo:lo + 32]
            variants[f"only k {lo}..{lo+31}"] = hv
            hv = h0.clone(); hv[:, lo:lo + 32] = 0
            variants[f"without k {lo}..{lo+31}"] = hv
        got = torch.from_numpy(outs[1][1, :, 0, :H]).double()
        badrows = sorted(set(bad[(bad[:, 0] == 1) & (bad[:, 2] == 0)][:, 1].tolist()))
        for name, hv in variants.items():
            h1, _ = step(xs[:, 1], hv, c0)
            err = (h1 - got).abs()
            print(f"    hypothesis '{name}': max err on bad rows {err[badrows].max():.2e}, on all rows {err.max():.2e}")
        print("    bad rows at t=1 dir 0:", badrows)
