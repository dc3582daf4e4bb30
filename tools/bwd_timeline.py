#!/usr/bin/env python
"""Per-step timeline of the persistent backward recurrence (lstm_xcd_bwd_kernel), workgroup 0 / thread 0:
a library built with -DONSSEN_XCD_PROFILE=1 (tools/ab_variants.py build prof "-DONSSEN_XCD_PROFILE=1") leaves 8 clock64 stamps
for each of the steps 64..127 in the tail of an oversized workspace when ONSSEN_BWD_DBG=1.
    ONSSEN_HIP_LIB=build_variants/libonssen_hip_prof.so python tools/bwd_timeline.py [B T H]"""
import os, sys
os.environ["ONSSEN_BWD_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    B, T, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (16, 400, 600)
    from onssen_amd import _abi
    from onssen_amd.hip import get_lib
    from onssen_amd.nn._core import BLSTMParams, PackedBLSTM
    lib = get_lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    rnn = BLSTMParams(2 * H, H, 1, 0.0).to(dev)
    ug = 4 * -(-H // 128)
    pk = PackedBLSTM(rnn).get(ug)
    Hp, NP = pk.Hp, pk.NP
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(B, T, 2 * H, device=dev)
    ws = torch.zeros(lib.blstm_workspace_bytes(B, T, 2 * H, H, 1, ug), dtype=torch.uint8, device=dev)
    y = torch.empty(T, B, 2, Hp, device=dev)
    gates = torch.empty(T, B, 2, NP, device=dev)
    cs = torch.empty(T, B, 2, Hp, device=dev)
    lib.lstm_train_forward(x.data_ptr(), T * 2 * H, 2 * H, B, T, 2 * H, H, ug, pk.wih_img[0].data_ptr(), pk.whh_x3[0].data_ptr(),
                           pk.bias[0].data_ptr(), y.data_ptr(), gates.data_ptr(), cs.data_ptr(), ws.data_ptr(), ws.numel(), st)
    form = _abi.LSTM_BWD_XCD
    nb = lib.lstm_train_backward_workspace_bytes(B, H, ug, form)
    wsb = torch.zeros(nb + T * 64, dtype=torch.uint8, device=dev)
    whh = pk.whh_bwd(form)[0]
    dy = torch.randn(T, B, 2, Hp, device=dev)
    for rep in range(3):
        g2 = gates.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.lstm_train_backward(B, T, H, ug, whh.data_ptr(), dy.data_ptr(), g2.data_ptr(), cs.data_ptr(), wsb.data_ptr(), wsb.numel(), form, st, None)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    d = wsb[nb:nb + T * 64].cpu().numpy().view(np.int64).reshape(T, 8)
    st_ = d[66:126]
    names = ["poll the partial sums (tagged data)", "sum + DPP + cell arithmetic + fragments to LDS", "barrier", "MFMA",
             "tagged partial stores issued", "dP store", "first request for the next step's sums", "-> next step"]
    nxt = d[67:127, 0]
    edges = [st_[:, 1] - st_[:, 0], st_[:, 2] - st_[:, 1], st_[:, 3] - st_[:, 2], st_[:, 4] - st_[:, 3], st_[:, 5] - st_[:, 4],
             st_[:, 7] - st_[:, 5], st_[:, 6] - st_[:, 7], nxt - st_[:, 6]]
    period = np.mean(nxt - st_[:, 0])
    print(f"B={B} T={T} H={H}: launch {ms * 1e3:.0f} us = {ms * 1e3 / T:.2f} us per step; clock64 period per step {period:.0f} ticks "
          f"({period / (ms * 1e3 / T):.0f} ticks/us)")
    for n, e in zip(names, edges):
        print(f"  {n:55s} {np.mean(e):8.0f} ticks  {100 * np.mean(e) / period:5.1f} %")
    print("status words (abort, safe protocol):", wsb[1120:1128].cpu().numpy().view(np.int32))


if __name__ == "__main__":
    main()
