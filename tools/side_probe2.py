#!/usr/bin/env python
"""Round 6c probe 2 (GPU box): CU-masked streams (hipExtStreamCreateWithCUMask) for the side work of the pipelined step.

tools/side_probe.py found that side kernels issued behind the resident pair launch run beside it only if their workgroups fit into what the
recurrence's CUs have left (<= 32 VGPRs); anything that needs one of the 16 FREE CUs waited for the pair launch to end.  Here the side stream is
confined to 2 CUs per XCD by a CU mask and the pair launch to the other 30 (mask bit i <-> XCC i % 8, CU slot i / 8 -- checked below by running
the pair launch, which needs 30 CUs on EVERY XCD, on the 240-bit mask), and the side kernels are issued BEFORE the pair launch as well as behind it."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def masked_stream(hip, bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask rc {rc}")
    return torch.cuda.ExternalStream(st.value)


def main():
    from onssen_amd import _abi
    from onssen_amd.nn._core import _XcdPolicy, _XcdStatus
    from onssen_amd.separation import DCPipeline
    hip = ctypes.CDLL("libamdhip64.so")
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
    Hp, NP = lib.lstm_geometry(H, pipe.ug)[:2]
    main_s = torch.cuda.current_stream()
    n_side = int(os.environ.get("SIDE_SLOTS", "2"))           # CU slots per XCD for the side stream
    s_pair = masked_stream(hip, range(0, 8 * (32 - n_side)))
    s_side = masked_stream(hip, range(8 * (32 - n_side), 256))
    s_plain = torch.cuda.Stream()
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
        "front+istft": [s_stft, s_index, s_img, s_istft],
        "all": [s_stft, s_index, s_img, s_gemm, s_istft],
    }
    res = {}
    reps = 10

    def run(name, fns, pair_stream, side_stream, side_first):
        tp, ts = [], []
        for r in range(reps + 2):
            torch.cuda.synchronize()
            a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            a.record(main_s)
            pair_stream.wait_event(a)
            side_stream.wait_event(a)
            if side_first:
                for f in fns:
                    f(side_stream.cuda_stream)
                c.record(side_stream)
            pair(pair_stream.cuda_stream)
            b.record(pair_stream)
            if not side_first:
                for f in fns:
                    f(side_stream.cuda_stream)
                c.record(side_stream)
            torch.cuda.synchronize()
            if r >= 2:
                tp.append(a.elapsed_time(b))
                ts.append(a.elapsed_time(c))
        tp.sort(); ts.sort()
        rec = {"pair_ms_median": round(tp[len(tp) // 2], 4), "pair_ms_max": round(tp[-1], 4), "side_done_ms_median": round(ts[len(ts) // 2], 4),
               "side_done_ms_max": round(ts[-1], 4)}
        print(name, json.dumps(rec), flush=True)
        return rec

    # the mask layout: the pair launch needs 30 CUs of every XCD
    res["pair_on_plain_stream"] = run("pair_on_plain_stream", [], s_plain, s_side, False)
    res["pair_on_240_mask"] = run("pair_on_240_mask", [], s_pair, s_side, False)
    _XcdStatus.poll(wait=True)
    res["aborts_after_mask_check"] = _XcdPolicy.aborts
    for name, fns in sets.items():
        if not fns:
            continue
        for mode, ps, first in (("masked_pair_side_first", s_pair, True), ("masked_pair_side_behind", s_pair, False),
                                ("plain_pair_side_first", s_plain, True), ("plain_pair_side_behind", s_plain, False)):
            res[f"{name}/{mode}"] = run(f"{name}/{mode}", fns, ps, s_side, first)
    # the side sets alone on the masked stream (their duration on 16 CUs)
    for name, fns in sets.items():
        if not fns:
            continue
        ts = []
        for r in range(6):
            torch.cuda.synchronize()
            a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s_side)
            for f in fns:
                f(s_side.cuda_stream)
            c.record(s_side)
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(c))
        res[f"{name}/alone_on_side_mask_ms"] = round(sorted(ts)[len(ts) // 2], 4)
        print(name, "alone on the side mask", res[f"{name}/alone_on_side_mask_ms"], flush=True)
    _XcdStatus.poll(wait=True)
    res["aborts"] = _XcdPolicy.aborts
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open(f"gpurun_out/r06c_side_probe2_slots{n_side}.json", "w"), indent=1)


if __name__ == "__main__":
    main()
