#!/usr/bin/env python
"""Round 6c probe (GPU box): what fits on the 16 CUs the pair launch of the pipelined step leaves free?

The pair launch (layer 1 of batch k-1 || layer 0 of batch k, 240 workgroups that fill their CUs' registers and 137 KB of LDS) leaves two
CUs per XCD idle for ~0.73 ms.  The step's MFMA-free ends -- STFT, target map, feature split (+ the layer-0 projection) of the NEXT batch,
mask-apply + iSTFT of the batch BEFORE -- are independent of it.  The round-6 probe put full-width side kernels in front of a recurrence
and lost (they take the whole chip first).  Here the side stream's kernels are issued BEHIND the pair launch (eager launches, host order),
so that the dispatcher can only place their workgroups where the recurrence is not.

Prints: pair alone; pair with each side set beside it (pair's own duration from events on its stream, and the time until the side stream
is done too)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from onssen_amd import _abi
    from onssen_amd.nn._core import _XcdPolicy, _XcdStatus
    from onssen_amd.separation import DCPipeline
    dev = torch.device("cuda", 0)
    wl = bench.build_workload("dc_l2", 32, dev)
    model, wav, hop, n, nfft = wl["model"], wl["wav"], wl["HOP"], wl["N"], wl["NFFT"]
    B = wav.shape[0]
    pipe = DCPipeline(model, B, n, nfft, hop, graph=False)
    lib, T, F, D, H = pipe.lib, pipe.T, pipe.F, pipe.D, pipe.H
    with torch.no_grad():
        pipe.push(wav, check=False)
        pipe.push(wav, check=False)
        pipe.push(wav, check=False)
    torch.cuda.synchronize()
    pk = model._packed.get(pipe.ug)
    Hp, NP = lib.lstm_geometry(H, pipe.ug)[:2]
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    # side buffers (the next batch's): features, target map, image, G
    logmag2 = torch.empty(B, T, F, device=dev)
    ri2 = torch.empty(B, T, F, 2, device=dev)
    cws2 = pipe.cws[0].clone()
    KBx = -(-F // 32)
    img2 = torch.zeros(T * B * KBx * 64, dtype=torch.int16, device=dev)
    G2 = torch.empty(T * B, 2 * NP, device=dev)
    out2 = torch.zeros(B, 2, n, device=dev)
    m = pipe.masks

    def pair(st):
        lib.blstm_pipe2_forward(pipe.logmag[0].data_ptr(), T * F, F, B, T, F, H, pipe.ug, [t.data_ptr() for t in pk.wih_img],
                                [t.data_ptr() for t in pk.whh_x3], [t.data_ptr() for t in pk.bias], pipe.ws.data_ptr(), pipe.wnb,
                                pipe.flags | _abi.BLSTM_G_READY, st)

    def s_stft(st):
        lib.stft_logmag(wav.data_ptr(), B, n, n, nfft, hop, 1e-7, logmag2.data_ptr(), ri2.data_ptr(), st)

    def s_index(st):
        lib.dc_index(logmag2.data_ptr(), B, T, F, D, pipe.db, cws2.data_ptr(), pipe.cnb, st)

    def s_img(st):
        lib.x3_image(logmag2.data_ptr(), F, T * F, B, T * B, F, img2.data_ptr(), st)

    def s_gemm(st):
        lib.linear_x3p(img2.data_ptr(), T * B, F, pk.wih_img[0].data_ptr(), pk.bias[0].data_ptr(), 2 * NP, _abi.EPI_BIAS, 0, 0.0,
                       G2.data_ptr(), B, B * 2 * NP, 2 * NP, st)

    def s_istft(st):
        lib.mask_istft(pipe.ri[0].data_ptr(), m.data_ptr(), m.stride(0), m.stride(3), m.stride(1), m.stride(2), B, 2, T, nfft, hop, n,
                       out2.data_ptr(), st)

    sets = {
        "none": [],
        "stft": [s_stft],
        "index": [s_index],
        "img": [s_img],
        "istft": [s_istft],
        "gemm_l0": [s_gemm],
        "front": [s_stft, s_index, s_img],
        "front+istft": [s_stft, s_index, s_img, s_istft],
        "front+gemm": [s_stft, s_index, s_img, s_gemm],
        "all": [s_stft, s_index, s_img, s_gemm, s_istft],
    }
    res = {}
    reps = 12
    for name, fns in sets.items():
        # side kernels alone (whole chip), for reference
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        alone = None
        if fns:
            for _ in range(2):
                for f in fns:
                    f(main.cuda_stream)
            e0.record(main)
            for _ in range(reps):
                for f in fns:
                    f(main.cuda_stream)
            e1.record(main)
            torch.cuda.synchronize()
            alone = e0.elapsed_time(e1) / reps
        tp, ts = [], []
        for r in range(reps + 2):
            torch.cuda.synchronize()
            a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            a.record(main)
            side.wait_event(a)
            pair(main.cuda_stream)
            b.record(main)
            for f in fns:
                f(side.cuda_stream)
            c.record(side)
            torch.cuda.synchronize()
            if r >= 2:
                tp.append(a.elapsed_time(b))
                ts.append(a.elapsed_time(c))
        tp.sort(); ts.sort()
        res[name] = {"side_alone_whole_chip_ms": alone, "pair_ms_median": tp[len(tp) // 2], "pair_ms_min": tp[0], "pair_ms_max": tp[-1],
                     "side_done_ms_median": ts[len(ts) // 2], "side_done_ms_max": ts[-1]}
        print(name, json.dumps(res[name]), flush=True)
    _XcdStatus.poll(wait=True)
    res["aborts"] = _XcdPolicy.aborts
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/r06c_side_probe.json", "w"), indent=1)


if __name__ == "__main__":
    main()
