#!/usr/bin/env python
"""Round 6c probe (GPU box): do the pipelined step's three independent GEMMs overlap when issued on separate streams?

After the pair launch of step k the step holds three GEMMs that do not depend on each other: layer 1's input projection of batch k
(750 tiles, MFMA-bound, 0.30 ms), fc_dc of batch k-1 on the active bins (450 tiles = 1.76 rounds of 256 CUs -> 2 rounds, 0.20 ms) and --
if the front end of batch k+1 were moved behind the pair launch -- layer 0's input projection of batch k+1 (750 tiles of 5 k-steps, bound
by its 245 MB of C stores, 0.08 ms).  One stream runs them back to back: every launch pays its own last partial round.  Prints the time of
each alone, of the three back to back on one stream, and of the combinations on separate streams (all streams released by one event)."""
import itertools
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from onssen_amd import _abi
    from onssen_amd.separation import DCPipeline
    dev = torch.device("cuda", 0)
    wl = bench.build_workload("dc_l2", 32, dev)
    model, wav, hop, n, nfft = wl["model"], wl["wav"], wl["HOP"], wl["N"], wl["NFFT"]
    B = wav.shape[0]
    pipe = DCPipeline(model, B, n, nfft, hop, graph=False)
    lib, T, F, D, H = pipe.lib, pipe.T, pipe.F, pipe.D, pipe.H
    for _ in range(3):
        pipe.push(wav, check=False)
    torch.cuda.synchronize()
    pk = model._packed.get(pipe.ug)
    hd = model._head.get(pk.Hp)
    Hp, NP = lib.lstm_geometry(H, pipe.ug)[:2]
    KBx = -(-F // 32)
    img_x = torch.zeros(T * B * KBx * 64, dtype=torch.int16, device=dev)
    lib.x3_image(pipe.logmag[0].data_ptr(), F, T * F, B, T * B, F, img_x.data_ptr(), torch.cuda.current_stream().cuda_stream)
    KB1 = -(-2 * Hp // 32)
    img0 = (torch.randn(T * B, KB1 * 64, device=dev) * 0.3).to(torch.bfloat16).view(torch.int16).contiguous()
    G0 = torch.empty(T * B, 2 * NP, device=dev)
    G1 = torch.empty(T * B, 2 * NP, device=dev)
    cw = pipe.cws[0]
    img1 = pipe.ws.data_ptr() + pipe.img_off

    def g_l0(st):
        lib.linear_x3p(img_x.data_ptr(), T * B, F, pk.wih_img[0].data_ptr(), pk.bias[0].data_ptr(), 2 * NP, _abi.EPI_BIAS, 0, 0.0,
                       G0.data_ptr(), B, B * 2 * NP, 2 * NP, st)

    def g_l1(st):
        lib.linear_x3p(img0.data_ptr(), T * B, 2 * Hp, pk.wih_img[1].data_ptr(), pk.bias[1].data_ptr(), 2 * NP, _abi.EPI_BIAS, 0, 0.0,
                       G1.data_ptr(), B, B * 2 * NP, 2 * NP, st)

    def g_head(st):
        lib.linear_x3p_compact(img1, T * B, 2 * pk.Hp, hd.img.data_ptr(), hd.b.data_ptr(), hd.N, D, 1e-12, cw.data_ptr() + pipe.dest_off,
                               T * F, F, cw.data_ptr() + pipe.comp_off, B, T * F * D, False, st)

    fns = {"l0": g_l0, "l1": g_l1, "head": g_head}
    main_s = torch.cuda.current_stream()
    streams = [torch.cuda.Stream() for _ in range(3)]
    res = {}
    reps = 20

    def timed_serial(names):
        for nme in names:
            fns[nme](main_s.cuda_stream)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main_s)
        for _ in range(reps):
            for nme in names:
                fns[nme](main_s.cuda_stream)
        b.record(main_s)
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    def timed_parallel(names):
        ts = []
        for r in range(reps + 3):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(main_s)
            for s, nme in zip(streams, names):
                s.wait_event(a)
            for s, nme in zip(streams, names):
                fns[nme](s.cuda_stream)
            for s, nme in zip(streams, names):
                main_s.wait_stream(s)
            b.record(main_s)
            torch.cuda.synchronize()
            if r >= 3:
                ts.append(a.elapsed_time(b))
        ts.sort()
        return ts[len(ts) // 2]

    for nme in fns:
        res[f"alone/{nme}"] = timed_serial([nme])
    for k in (2, 3):
        for names in itertools.permutations(fns, k):
            res["serial/" + "+".join(names)] = timed_serial(list(names)) if names == tuple(sorted(names)) else None
            res["parallel/" + "+".join(names)] = timed_parallel(list(names))
    res = {k: (round(v, 4) if v is not None else None) for k, v in res.items() if v is not None}
    for k, v in res.items():
        print(k, v, flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/r06c_gemm_overlap_probe.json", "w"), indent=1)


if __name__ == "__main__":
    main()
