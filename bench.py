#!/usr/bin/env python
"""bench.py -- separation throughput of the HIP hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic 8 kHz
2-speaker mixtures that is already resident in HBM:
    K1+K2  framed STFT 256/64 + log-magnitude
    K3-K8  deep-clustering network (BASELINE configs[1]: 2 x BLSTM-600, fc_dc + per-bin L2 normalise)
    N2     threshold at max - 40 dB + 2-means on the active bins' embeddings -> binary masks, on the device (what
           egs/wsj0-2mix/deep_clustering/evaluate.py:36-41 does with sklearn on the host)
    K10    mask-apply + iSTFT overlap-add for 2 speakers
(round 4: the headline IS the real separation; the step with resident random masks instead of the clustering -- the headline of
rounds 1-3 -- is reported next to it as "resident_mask_step")
captured once in a hipGraph and replayed.  N>1: one process per GPU (launched by
torch.distributed.run), every rank separates its own batch -- utterances are independent, so
there is no data-path collective ("scaling": "weak").

Prints ONE JSON line (rank 0): metric = real-time factor (audio seconds separated per wall
second, whole job), plus frames/s, a roofline block for the dominant kernel (every leg of the step timed
by itself with HIP events: the legs sum to <= the step), "extra_configs" -- the other BASELINE.json
configurations and the batch-1 whole-utterance latency, a few graph replays each, AFTER the timed region --
and the CPU baseline (oracle restatement on the host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR, NFFT, HOP, T_FRAMES = 8000, 256, 64, 400
N_SAMPLES = 25536                                    # 1 + 25536 // 64 = 400 frames
BF16_MFMA_PEAK_TFLOPS = 2500.0                       # dense bf16 MFMA (2:1-sparse marketing figure excluded)
FP32_MFMA_PEAK_TFLOPS = 157.3                        # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
HBM_PEAK_GBS = 8000.0

CONFIGS = {
    # name: (kind, hidden, layers, batch per GPU)
    "dc_l2": ("deep_clustering", 600, 2, 32),   # BASELINE.json configs[1] -- the headline workload
    "dc_l3": ("deep_clustering", 600, 3, 16),   # as-shipped egs/wsj0-2mix/deep_clustering/config.json
    "chimera_l4": ("chimera", 600, 4, 64),      # BASELINE.json configs[2]
    "phase_l4": ("phase_net", 600, 4, 32),      # BASELINE.json configs[4]: 16 kHz STFT 512/128, 1 s chunks
}


def flops_per_frame(kind, F, H, L, D=20, C=2):
    """sum_layers 2*2*4H*(in_l+H) + heads 2*2H*N (2 FLOP per MAC, forward)."""
    fl = 0
    for l in range(L):
        fl += 2 * 2 * 4 * H * ((F if l == 0 else 2 * H) + H)
    fl += 2 * 2 * H * F * D
    if kind == "chimera":
        fl += 2 * 2 * H * F * C
    return fl


def build_workload(config, B, dev, rank=0, T=None):
    """Model (random-init weights of the named architecture), resident synthetic input and the step function of one
    configuration.  ``T`` overrides the chunk length in frames (whole-utterance latency legs)."""
    from onssen_amd import nn as onn
    from onssen_amd.features import mask_istft, stft_logmag
    from onssen_amd.synthetic import make_state_dict, synth_batch
    kind, H, L, _ = CONFIGS[config]
    sr, nfft, hop, n = 8000, 256, 64, 25536
    if kind == "phase_net":
        sr, nfft, hop, n = 16000, 512, 128, 16000
    if T is not None:
        n = hop * (T - 1)
    Tn = 1 + n // hop
    F, D = nfft // 2 + 1, 20
    sd = make_state_dict(kind, F, H, L, D, 2, seed=0)
    cls = {"deep_clustering": onn.deep_clustering, "chimera": onn.chimera, "phase_net": onn.phase_net}[kind]
    model = cls(F, H, L, D)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model = model.to(dev).eval()
    # synthetic mixtures: B DISTINCT structured utterances per rank (round 6; up to 64, tiled beyond -- the batch sweep's 128 / 256 rows).
    # Rounds 1-5 tiled 8 utterances over the batch: the network does not care, but the 2-means' pass count is per utterance and the
    # persistent Lloyd launch lasts as long as its slowest one, so 8 x 4 copies flattered the deep-clustering step by ~4 %
    base = synth_batch(1 + rank, min(B, 64), n, sr)
    wav_np = np.concatenate([base] * ((B + len(base) - 1) // len(base)))[:B]
    wav = torch.from_numpy(wav_np).to(dev)
    gen = torch.Generator(device="cpu").manual_seed(rank)
    m0 = (torch.rand(B, Tn, F, generator=gen) > 0.5).float()
    bin_masks = torch.stack([m0, 1 - m0], -1).to(dev)   # stand-in for the K-means assignment

    from onssen_amd.separation import dc_masks, dc_masks_from_features

    def step_resident_masks():       # rounds 1-3's headline: the network's embedding is computed, the masks are a resident stand-in
        logmag, ri = stft_logmag(wav, nfft, hop)
        emb, = model([logmag])
        return emb, mask_istft(ri, bin_masks, hop, n)

    def step():
        logmag, ri = stft_logmag(wav, nfft, hop)
        if kind == "phase_net":
            emb, mA, mB, pA, pB = model([logmag, ri])
            sig = mask_istft(ri, mA._base.view(B, Tn, F, 2), hop, n)
        elif kind == "chimera":
            emb, masks = model.embedding_and_masks(logmag)
            sig = mask_istft(ri, masks, hop, n)
        else:                          # deep clustering: the real separation (threshold + 2-means -> binary masks), as
            emb = dc_masks_from_features(model, logmag)      # separation.separate_dc runs it: the fc_dc GEMM stores only the
            if emb is None:                                  # active bins' embeddings, straight into the clustering's array
                e, = model([logmag])
                emb = dc_masks(e, logmag)
            sig = mask_istft(ri, emb, hop, n)
        return emb, sig
    return dict(kind=kind, H=H, L=L, B=B, F=F, D=D, SR=sr, NFFT=nfft, HOP=hop, T=Tn, N=n, model=model, wav=wav, wav_np=wav_np,
                bin_masks=bin_masks, sd=sd, step=step, step_resident_masks=step_resident_masks)


def capture(step, use_graph=True):
    """First call (packs weights, allocates workspaces), a warm-up on a side stream, then ONE hipGraph of the whole step.
    Returns (run, graph)."""
    step()
    torch.cuda.synchronize()
    if not use_graph:
        return step, None
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    return graph.replay, graph


def time_replays(run, reps, warm=2):
    for _ in range(warm):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def extra_configs(dev):
    """The other BASELINE.json configurations and the evaluation shape the reference actually runs (whole utterances, batch 1:
    onssen/utils/test.py:29-41), a few hipGraph replays each, AFTER the timed region (rank 0, N = 1).  Every entry is the same
    kind of step as the headline (waveform -> STFT -> network -> mask-apply + iSTFT, inputs resident in HBM)."""
    from onssen_amd.nn._core import _XcdStatus
    out = {}

    def leg(name, config, B, T=None, precision="bf16x3", reps=6):
        old = os.environ.get("ONSSEN_PRECISION")
        os.environ["ONSSEN_PRECISION"] = precision
        try:
            with torch.no_grad():
                wl = build_workload(config, B, dev, 0, T)
                run, _ = capture(wl["step"])
                ms = time_replays(run, reps)
            torch.cuda.synchronize()
            _XcdStatus.poll(wait=True)
            audio = B * (wl["T"] * wl["HOP"] / wl["SR"])
            out[name] = {"config": config, "chunks": B, "frames_per_chunk": wl["T"], "stft": f"{wl['NFFT']}/{wl['HOP']} @ {wl['SR'] // 1000} kHz",
                         "precision": precision, "ms_per_step": ms, "x_real_time": audio / ms * 1e3,
                         "frames_per_s": B * wl["T"] / ms * 1e3, "launch": "hipGraph replay"}
        except Exception as e:          # an extra leg must never take the headline line down with it
            out[name] = {"config": config, "error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            if old is None:
                os.environ.pop("ONSSEN_PRECISION", None)
            else:
                os.environ["ONSSEN_PRECISION"] = old
            torch.cuda.empty_cache()
    leg("cfg3_chimera_l4_b64", "chimera_l4", 64)
    leg("cfg5_phase_l4_b32", "phase_l4", 32)
    leg("dc_l3_b16_as_shipped", "dc_l3", 16)
    leg("cfg2_literal_bf16_b32", "dc_l2", 32, precision="bf16")
    leg("b1_utterance_T400", "dc_l2", 1, T=400, reps=10)
    leg("b1_utterance_T1000", "dc_l2", 1, T=1000, reps=10)
    leg("b1_utterance_T1000_dc_l3", "dc_l3", 1, T=1000, reps=10)
    try:
        out["b16_ragged_utterances"] = ragged_leg(dev)
    except Exception as e:
        out["b16_ragged_utterances"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    for k in ("b1_utterance_T400", "b1_utterance_T1000", "b1_utterance_T1000_dc_l3"):
        if "ms_per_step" in out.get(k, {}):
            out[k]["latency_ms"] = out[k]["ms_per_step"]
    try:
        out["cfg4_training_step_dc_l3_b16"] = training_leg(dev)
    except Exception as e:
        out["cfg4_training_step_dc_l3_b16"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    try:
        out["trained_weights_dc_l2_b32"] = trained_leg(dev)
    except Exception as e:
        out["trained_weights_dc_l2_b32"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    try:
        out["batch_sweep"] = batch_sweep(dev)
    except Exception as e:
        out["batch_sweep"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


import contextlib


@contextlib.contextmanager
def geometry(wl):
    """kernel_roofline reads the STFT geometry from module globals (set by main for the headline): swap in ``wl``'s."""
    global SR, NFFT, HOP, T_FRAMES, N_SAMPLES
    old = (SR, NFFT, HOP, T_FRAMES, N_SAMPLES)
    SR, NFFT, HOP, T_FRAMES, N_SAMPLES = wl["SR"], wl["NFFT"], wl["HOP"], wl["T"], wl["N"]
    try:
        yield
    finally:
        SR, NFFT, HOP, T_FRAMES, N_SAMPLES = old


def lloyd_iterations(B, T, F, D, ws=None):
    """Lloyd passes the device 2-means of the LAST separation step of this shape actually ran, per utterance, from the
    header of its workspace (csrc/labels_cluster.inc: word 66 of an utterance's 72-word record = passes | 0x10000 once it
    stopped by the tolerance rule / at the fixed point; word 64 = active bins)."""
    from onssen_amd import separation
    from onssen_amd.hip import get_lib
    lib = get_lib()
    if ws is None:
        keys = [k for k in separation._CLUSTER_WS if tuple(k[1:5]) == (B, T, F, D)]
        if not keys:
            return None
        keys.sort(key=lambda k: k not in separation._CLUSTER_PINNED)          # the captured step's buffer first
        ws = separation._CLUSTER_WS[keys[0]]
    so = int(lib.dll.onssen_dc_cluster_status_offset(B, D))
    iw = ws[so - B * 72 * 4:so].view(torch.int32).view(B, 72).cpu().numpy()
    it = (iw[:, 66] & 0xffff).astype(int)
    return {"per_utterance": it.tolist(), "min": int(it.min()), "max": int(it.max()), "mean": float(it.mean()),
            "stopped_before_the_cap": int(((iw[:, 66] >> 16) & 1).sum()), "cap": 20,
            "active_bin_fraction": float(iw[:, 64].sum()) / float(B * T * F)}


def batch_sweep(dev):
    """the roofline fraction "at the config batch and additionally as a batch sweep" -- ONE model across
    chunk counts: the headline network (dc_l2) at 8 .. 256 chunks and cfg5's phase_net at 32 .. 128, the whole step (hipGraph
    replay) and the recurrence kernel by itself.  ``knee`` = the smallest batch within 10 % of the best x-real-time: the
    chunks-per-GPU to shard cfg5 / the headline at."""
    from onssen_amd.nn._core import _XcdStatus
    out = {}
    for config, Bs in (("dc_l2", (8, 16, 32, 64, 128, 256)), ("phase_l4", (32, 64, 128))):
        rows = []
        for B in Bs:
            try:
                with torch.no_grad():
                    wl = build_workload(config, B, dev, 0)
                    run, _ = capture(wl["step"])
                    ms = time_replays(run, 5)
                    kind = wl["kind"]
                    with geometry(wl):
                        roof = kernel_roofline(wl["model"].chimera if kind == "phase_net" else wl["model"], wl["wav"], dev,
                                               "chimera" if kind == "phase_net" else kind, wl["F"], wl["H"], wl["L"], B, wl["D"],
                                               recurrence_only=True)
                torch.cuda.synchronize()
                _XcdStatus.poll(wait=True)
                audio = B * (wl["T"] * wl["HOP"] / wl["SR"])
                rows.append({"chunks": B, "ms_per_step": ms, "x_real_time": audio / ms * 1e3, "frames_per_s": B * wl["T"] / ms * 1e3,
                             "recurrence_us_per_time_step": roof["us_per_time_step"], "recurrence_frac_of_peak": roof["frac"],
                             "recurrence_launches_per_layer": roof["launches_per_step"] // wl["L"],
                             "recurrence_TFLOPs": roof["achieved"]})
                del wl, run
            except Exception as e:
                rows.append({"chunks": B, "error": f"{type(e).__name__}: {e}"[:200]})
            torch.cuda.empty_cache()
        ok = [r for r in rows if "x_real_time" in r]
        best = max((r["x_real_time"] for r in ok), default=0.0)
        knee = next((r["chunks"] for r in ok if r["x_real_time"] >= 0.9 * best), None)
        out[config] = {"rows": rows, "knee_chunks": knee, "best_x_real_time": best}
    return out


def trained_leg(dev, B=32, steps=1000):
    """The headline step on a TRAINED network (the headline itself runs random-init weights of the named architecture: there
    is no checkpoint in the image).  The clustering's cost depends on the embeddings: a random-init network gives 2-means
    nothing to find (15-17 Lloyd passes of the 20 allowed), a trained one converges in 5-6.  So: ``steps`` optimizer steps of
    dc_l2 on the synthetic voice-pair corpus with the package's own training step (~5 s), then the SAME captured step as the
    headline on B held-out synthetic mixtures; reports its time, the Lloyd passes it ran and the SI-SDR of what it separated
    (egs/wsj0-2mix/deep_clustering/evaluate.py:31-45; tools/trained_probe.py is the long form)."""
    from onssen_amd import nn as onn
    from onssen_amd.data import SyntheticVoicePairs
    from onssen_amd.dist import train_step
    from onssen_amd.evaluate import batch_SDR_torch
    from onssen_amd.features import mask_istft, stft_logmag
    from onssen_amd.loss import loss_dc
    from onssen_amd.nn._core import _XcdStatus
    from onssen_amd.separation import dc_masks, dc_masks_from_features
    from onssen_amd.synthetic import synth_mixture
    from onssen_amd.utils import build_optimizer
    F, H, L, D, n = 129, 600, 2, 20, 25536
    torch.manual_seed(0)
    model = onn.deep_clustering(F, H, L, D, dropout=0.3).to(dev).train()
    opt = build_optimizer(model.parameters(), {"name": "adam", "lr": 1e-3})
    data = SyntheticVoicePairs(dict(batch_size=16, frame_length=400, sampling_rate=8000, window_size=256, hop_size=64, db_threshold=40), device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = [train_step(model, opt, loss_dc, *next(data)) for _ in range(steps)]
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    model.eval()
    trips = [synth_mixture(910_000 + u, n, 8000, return_sources=True) for u in range(B)]      # held out: other seeds, other voices
    wav = torch.from_numpy(np.stack([t[0] for t in trips])).to(dev)
    ref = torch.from_numpy(np.stack([np.stack(t[1:]) for t in trips])).to(dev)

    def step():
        logmag, ri = stft_logmag(wav, 256, 64)
        masks = dc_masks_from_features(model, logmag)
        if masks is None:
            e, = model([logmag])
            masks = dc_masks(e, logmag)
        return mask_istft(ri, masks, 64, n)
    with torch.no_grad():
        run, _ = capture(step)
        ms = time_replays(run, 10)
        sig = step()
        sdr = batch_SDR_torch(sig, ref)
        sdr_mix = batch_SDR_torch(torch.stack([wav, wav], 1), ref)
        # the same weights and mixtures through the two-batch pipeline (what the headline line times): its step, its SI-SDR
        piped = None
        try:
            from onssen_amd.separation import DCPipeline
            pipe = DCPipeline(model, B, n, 256, 64)
            pipe.push(wav)
            sdr_p = batch_SDR_torch(pipe.push(wav), ref)
            pipe.wav[0].copy_(wav); pipe.wav[1].copy_(wav)
            ms_p = time_replays(pipe.replay, 20, warm=20)
            piped = {"ms_per_step": ms_p, "x_real_time": B * 3.2 / ms_p * 1e3, "si_sdr_db_separated_mean": float(sdr_p.mean()),
                     "lloyd_iterations": lloyd_iterations(B, 400, F, D, ws=pipe.cws[pipe.count & 1])}
        except RuntimeError as e:
            piped = {"error": f"{e}"[:200]}
    torch.cuda.synchronize()
    _XcdStatus.poll(wait=True)
    return {"workload": f"the headline step (dc_l2, {B} x 400-frame chunks, hipGraph replay) with weights from {steps} training steps on the "
                        "synthetic voice-pair corpus, on held-out synthetic mixtures",
            "training": {"steps": steps, "seconds": train_s, "ms_per_step_with_corpus": train_s / steps * 1e3,
                         "loss_first_100": float(np.mean(losses[:100])), "loss_last_100": float(np.mean(losses[-100:]))},
            "ms_per_step": ms, "x_real_time": B * 3.2 / ms * 1e3, "lloyd_iterations": lloyd_iterations(B, 400, F, D),
            "pipelined": piped,
            "si_sdr_db": {"separated_mean": float(sdr.mean()), "separated_min": float(sdr.min()), "mixture_mean": float(sdr_mix.mean())}}


def ragged_leg(dev, K=16, batches=6):
    """The reference's evaluation shape (whole utterances of different lengths, onssen/utils/test.py:29-41) K at a time:
    ``separate_dc(model, wav, lengths=...)`` -- STFT, DC 2xBLSTM-600, threshold + 2-means, mask-apply + iSTFT, every stage told
    each row's own extent -- on batches of K synthetic utterances of 3-8 s, eager launches (every batch has its own longest
    utterance: nothing to capture).  Real time is counted on the utterances' OWN durations, not on the padded batch."""
    from onssen_amd import nn as onn
    from onssen_amd.nn._core import _XcdStatus
    from onssen_amd.separation import separate_dc
    from onssen_amd.synthetic import make_state_dict, synth_batch
    F, H, L, D = 129, 600, 2, 20
    sd = make_state_dict("deep_clustering", F, H, L, D, 2, seed=0)
    model = onn.deep_clustering(F, H, L, D)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model = model.to(dev).eval()
    rng = np.random.default_rng(7)
    all_ns = [int(v) for v in rng.integers(3 * 8000, 8 * 8000, K * batches)]

    def make_sets(lengths):
        out_sets = []
        for i in range(batches):
            ns = lengths[i * K:(i + 1) * K]
            base = synth_batch(50 + i, 8, max(ns), 8000)
            wav = torch.from_numpy(np.concatenate([base, base[::-1]])[:K].copy()).to(dev)
            out_sets.append((wav, torch.tensor(ns, dtype=torch.int32, device=dev), ns))
        return out_sets
    sets = make_sets(all_ns)                                  # batches in the order the utterances come
    sets_sorted = make_sets(sorted(all_ns, reverse=True))     # the same utterances bucketed by length (tester.eval(bucket=...))
    with torch.no_grad():
        for wav, ln, _ in sets[:2]:
            separate_dc(model, wav, lengths=ln)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for wav, ln, _ in sets:
            separate_dc(model, wav, lengths=ln)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for wav, ln, _ in sets_sorted[:2]:
            separate_dc(model, wav, lengths=ln)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for wav, ln, _ in sets_sorted:
            separate_dc(model, wav, lengths=ln)
        torch.cuda.synchronize()
        dts = time.perf_counter() - t0
        one = sets[0]
        t1 = time.perf_counter()
        for b in range(K):
            separate_dc(model, one[0][b:b + 1, :one[2][b]].contiguous())
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        # round 6c: the same batches through the ragged pipeline (separation.DCRaggedPipeline: layer 1 of batch n-1 beside layer 0 of
        # batch n in one persistent launch, stacked 8-row groups) -- same bits per utterance, checked below
        piped = {}
        try:
            from onssen_amd.separation import DCRaggedPipeline
            pipe = DCRaggedPipeline(model, K, 8 * 8000)

            def through(ss, keep=False):
                res = []
                for wav, ln, _ in ss:
                    o = pipe.push(wav, ln, check=False)
                    if keep and o is not None:
                        res.append(o.clone())
                o = pipe.flush()
                if keep:
                    res.append(o.clone())
                return res
            for name, ss in (("as_they_come", sets), ("bucketed_by_length", sets_sorted)):
                through(ss[:2])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                through(ss)
                torch.cuda.synchronize()
                dtp = time.perf_counter() - t0
                got = through(ss, keep=True)
                same = all(torch.equal(g, separate_dc(model, wav, lengths=ln)) for g, (wav, ln, _) in zip(got, ss))
                piped[name] = {"ms_per_utterance": dtp / (K * batches) * 1e3, "x_real_time": sum(sum(ns) for _, _, ns in ss) / 8000.0 / dtp,
                               "bit_identical_to_separate_dc": bool(same)}
        except Exception as e:
            piped = {"error": f"{type(e).__name__}: {e}"[:300]}
        # the honest comparison for the pipeline: the SAME utterances 2K at a time through the plain call (8-row groups fill the chip
        # there too, at the price of padding 2K rows to their longest and of twice the batch in flight)
        k2 = {}
        try:
            def merged(ss):
                res = []
                for i in range(0, len(ss) - 1, 2):
                    (w0, l0, n0), (w1, l1, n1) = ss[i], ss[i + 1]
                    n = max(w0.shape[1], w1.shape[1])
                    w = torch.zeros(2 * K, n, device=dev)
                    w[:K, :w0.shape[1]] = w0
                    w[K:, :w1.shape[1]] = w1
                    res.append((w, torch.cat([l0, l1]), n0 + n1))
                return res
            for name, ss in (("as_they_come", merged(sets)), ("bucketed_by_length", merged(sets_sorted))):
                for wav, ln, _ in ss[:1]:
                    separate_dc(model, wav, lengths=ln)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for wav, ln, _ in ss:
                    separate_dc(model, wav, lengths=ln)
                torch.cuda.synchronize()
                dtk = time.perf_counter() - t0
                k2[name] = {"ms_per_utterance": dtk / (K * batches) * 1e3, "x_real_time": sum(sum(ns) for _, _, ns in ss) / 8000.0 / dtk}
        except Exception as e:
            k2 = {"error": f"{type(e).__name__}: {e}"[:300]}
    _XcdStatus.poll(wait=True)
    audio = sum(sum(ns) for _, _, ns in sets) / 8000.0
    return {"workload": f"separate_dc on {batches} ragged batches of {K} whole utterances (3-8 s each, padded to the batch's longest), "
                        "dc_l2, device 2-means, eager launches",
            "utterances": K * batches, "audio_s": audio, "ms_per_utterance": dt / (K * batches) * 1e3, "x_real_time": audio / dt,
            "one_by_one_ms_per_utterance": dt1 / K * 1e3, "one_by_one_x_real_time": sum(one[2]) / 8000.0 / dt1,
            "padding_overhead": sum(K * max(ns) for _, _, ns in sets) / sum(sum(ns) for _, _, ns in sets),
            "bucketed_by_length": {"ms_per_utterance": dts / (K * batches) * 1e3, "x_real_time": audio / dts,
                                   "padding_overhead": sum(K * max(ns) for _, _, ns in sets_sorted) / sum(sum(ns) for _, _, ns in sets_sorted)},
            "pipelined": dict(piped, what="separation.DCRaggedPipeline (onssen_blstm_pipe2_forward_ragged_f32): layer 1 of batch n-1 beside layer 0 "
                                          "of batch n in one persistent launch, stacked 8-row groups; per-batch latency two steps"),
            "plain_call_on_2K_rows": dict(k2, what=f"separate_dc on the same utterances {2 * K} at a time (8-row groups fill the chip in the plain "
                                                   "call too; more padding, twice the rows in flight)")}


def dp_training_leg(dev, rank, world, one_dev, layers=3, B=16, steps=6, warmup=3):
    """BASELINE config 4 data parallel (every rank calls this): ``dist.train_step`` -- forward -> loss_dc -> backward with the
    per-layer gradient buckets all-reduced over RCCL while the layers below still back-propagate -> clip -> Adam
    (onssen/utils/train.py:75-86, the exchange between :82 and :83) -- with B chunks per rank, identical replicas, every rank
    its own synthetic batches.  Timed twice inside the same job, each between barriers, max over ranks: with the exchange
    (world N) and, afterwards, every rank stepping alone (world 1: no collective); the difference is the all-reduce time the
    overlap did NOT hide.  Also checks that the replicas are still bit-identical after the data-parallel steps."""
    import torch.distributed as dist
    from onssen_amd import nn as onn
    from onssen_amd.data.synthetic_wsj0_2mix import SyntheticWsj02mix
    from onssen_amd.dist import train_step
    from onssen_amd.loss import loss_dc
    from onssen_amd.nn._core import _XcdStatus
    from onssen_amd.utils import build_optimizer
    fo = dict(batch_size=B, frame_length=400, sampling_rate=8000, window_size=256, hop_size=64, db_threshold=40)
    torch.manual_seed(0)                                   # identical replicas on every rank
    model = onn.deep_clustering(129, 600, layers, 20, dropout=0.3).to(dev).train()
    opt = build_optimizer(model.parameters(), {"name": "adam", "lr": 1e-3})
    batches = []
    for i, b in enumerate(SyntheticWsj02mix("dc", fo, "tr", device=str(dev), num_batches=3, seed=100_000 * rank)):
        batches.append(b)
    torch.manual_seed(1234 + rank)                         # dropout seeds differ per replica, like independent DataLoader workers
    cpu_or_dev = "cpu" if one_dev else dev

    def timed(w):
        for i in range(warmup):
            train_step(model, opt, loss_dc, *batches[i % len(batches)], world=w)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss = train_step(model, opt, loss_dc, *batches[i % len(batches)], world=w)
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        dist.barrier()
        torch.cuda.synchronize()
        span = time.perf_counter() - t0
        te = torch.tensor([mine, span, loss], device=cpu_or_dev, dtype=torch.float64)
        every = [torch.empty_like(te) for _ in range(world)]
        dist.all_gather(every, te)
        return ([1e3 * float(t[0]) / steps for t in every], 1e3 * max(float(t[1]) for t in every) / steps, [float(t[2]) for t in every])

    per_rank_dp, ms_dp, losses = timed(world)
    # replicas identical after the data-parallel steps?  (a checksum of every parameter, compared across ranks)
    with torch.no_grad():
        chk = torch.stack([p.double().sum() for p in model.parameters()]).sum().reshape(1).to(cpu_or_dev)
    allchk = [torch.empty_like(chk) for _ in range(world)]
    dist.all_gather(allchk, chk)
    identical = all(float(c) == float(allchk[0]) for c in allchk)
    reducer = getattr(model, "_onssen_reducer", None)
    buckets = [sum(p.numel() for p in ps) * 4 for ps in reducer.buckets] if reducer is not None else []
    issued = reducer.issued_in_backward if reducer is not None else 0
    per_rank_alone, ms_alone, _ = timed(1)                 # (replicas drift apart from here on: nothing reads them afterwards)
    _XcdStatus.poll(wait=True)
    grad_bytes = sum(p.numel() for p in model.parameters() if p.requires_grad) * 4
    return {"workload": f"deep_clustering {layers}xBLSTM-600 DATA-PARALLEL training step, {B} x 400-frame chunks per rank x {world} ranks, "
                        "dropout 0.3, Adam; features + labels from the HIP front end; eager launches",
            "collective": ("gloo all-reduce (ONSSEN_BENCH_ONE_DEVICE harness self-test: every rank on cuda:0, launch-per-step kernels, "
                           "device tensors staged through the host by gloo: the timings of this mode only show that the leg runs, 2 steps)" if one_dev
                           else "RCCL all-reduce(sum) of per-layer gradient buckets, issued from inside the HIP backward, before clip_grad_norm_"),
            "ms_per_step": ms_dp, "per_rank_ms_per_step": per_rank_dp,
            "x_real_time": world * B * 3.2 / ms_dp * 1e3, "frames_per_s": world * B * 400 / ms_dp * 1e3,
            "ms_per_step_without_exchange": ms_alone, "per_rank_ms_per_step_without_exchange": per_rank_alone,
            "exposed_all_reduce_ms": ms_dp - ms_alone,
            "all_reduce_bytes_per_step": grad_bytes, "bucket_bytes": buckets, "buckets_issued_inside_backward": issued,
            "replicas_identical_after_dp_steps": identical, "last_loss_per_rank": losses,
            "rccl_max_channels": os.environ.get("NCCL_MAX_NCHANNELS")}


def training_leg(dev, layers=3, B=16, steps=6, warmup=3):
    """BASELINE config 4 on one GPU: forward -> loss_dc -> backward -> (all-reduce: world 1) -> clip -> Adam, features and
    labels from the HIP front end (synthetic wsj0-2mix loader), eager launches; plus a roofline block for the backward
    recurrence kernel timed by itself with HIP events."""
    from onssen_amd import _abi, nn as onn
    from onssen_amd.data.synthetic_wsj0_2mix import wsj0_2mix_dataloader
    from onssen_amd.dist import train_step
    from onssen_amd.hip import get_lib
    from onssen_amd.loss import loss_dc
    from onssen_amd.nn._core import _XcdStatus
    from onssen_amd.utils import build_optimizer
    lib = get_lib()
    fo = dict(batch_size=B, frame_length=400, sampling_rate=8000, window_size=256, hop_size=64, db_threshold=40)
    torch.manual_seed(0)
    model = onn.deep_clustering(129, 600, layers, 20, dropout=0.3).to(dev).train()
    opt = build_optimizer(model.parameters(), {"name": "adam", "lr": 1e-3})
    batches = []
    for i, b in enumerate(wsj0_2mix_dataloader("dc", fo, "tr", device=str(dev))):
        batches.append(b)
        if i >= 2:
            break
    for i in range(warmup):
        train_step(model, opt, loss_dc, *batches[i % len(batches)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = train_step(model, opt, loss_dc, *batches[i % len(batches)])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    # ---- the backward recurrence by itself: saved state of one layer from a training forward, events around the launch
    H, T, ug = 600, 400, 20
    pk = model.rnn._train_packed.get(ug)
    Hp, NP = pk.Hp, pk.NP
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(T, B, 2 * Hp, device=dev).tanh_()
    y = torch.empty(T, B, 2, Hp, device=dev)
    gates, cs = torch.empty(T, B, 2, NP, device=dev), torch.empty(T, B, 2, Hp, device=dev)
    ws = torch.zeros(lib.blstm_workspace_bytes(B, T, 2 * Hp, H, 1, ug), dtype=torch.uint8, device=dev)
    lib.lstm_train_forward(x.data_ptr(), 2 * Hp, B * 2 * Hp, B, T, 2 * Hp, H, ug, pk.wih_img[1].data_ptr(), pk.whh_x3[1].data_ptr(),
                           pk.bias[1].data_ptr(), y.data_ptr(), gates.data_ptr(), cs.data_ptr(), ws.data_ptr(), ws.numel(), st)
    keep = gates.clone()
    dy = torch.randn(T, B, 2, Hp, device=dev) * 0.01
    form = _abi.LSTM_BWD_XCD
    wsb = torch.zeros(lib.lstm_train_backward_workspace_bytes(B, H, ug, form), dtype=torch.uint8, device=dev)
    whh = pk.whh_bwd(form)[1]
    db_rows = torch.empty(B, 2 * NP, device=dev)
    tot = 0.0
    for r in range(5):
        gates.copy_(keep)                       # the backward recurrence overwrites the gates in place
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.lstm_train_backward(B, T, H, ug, whh.data_ptr(), dy.data_ptr(), gates.data_ptr(), cs.data_ptr(), wsb.data_ptr(),
                                wsb.numel(), form, st, db_rows.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        if r:
            tot += e0.elapsed_time(e1)
    t_bwd = tot / 4 * 1e-3
    # and the saved-state forward the same way
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(4):
        lib.lstm_train_forward(x.data_ptr(), 2 * Hp, B * 2 * Hp, B, T, 2 * Hp, H, ug, pk.wih_img[1].data_ptr(), pk.whh_x3[1].data_ptr(),
                               pk.bias[1].data_ptr(), y.data_ptr(), gates.data_ptr(), cs.data_ptr(), ws.data_ptr(), ws.numel(), st)
    e1.record()
    torch.cuda.synchronize()
    t_fwd_layer = e0.elapsed_time(e1) / 4 * 1e-3
    _XcdStatus.poll(wait=True)
    flop = 2.0 * 2 * B * 4 * H * H * T
    peak = BF16_MFMA_PEAK_TFLOPS / 3.0
    return {"workload": f"deep_clustering {layers}xBLSTM-600 training step, {B} x 400-frame chunks, dropout 0.3, Adam; features + labels "
                        "from the HIP front end; eager launches",
            "ms_per_step": ms, "x_real_time": B * 3.2 / ms * 1e3, "frames_per_s": B * 400 / ms * 1e3, "last_loss": loss,
            "roofline_backward_recurrence": {
                "kernel": "lstm_xcd_bwd_kernel", "bound": "mfma", "achieved": flop / t_bwd / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": flop / t_bwd / 1e12 / peak, "us_per_launch": t_bwd * 1e6, "us_per_time_step": t_bwd / T * 1e6,
                "algorithmic_flop_per_launch": flop, "launches_per_step": layers,
                "bound_note": "a serial chain of T dependent steps; the exchange is a reduce-scatter of fp32 partial sums "
                              "through the XCD's L2 (DESIGN.md section 3)"},
            "forward_layer_with_saved_state_ms": t_fwd_layer * 1e3}


def self_launch(n):
    """``python bench.py --gpus N`` without a launcher: re-run this command line as N ranks of ONE node under
    ``python -m torch.distributed.run`` (one process per GPU, rendezvous on 127.0.0.1, a free port) and return its exit code.
    Rank 0's JSON line goes to this process's stdout unchanged.  With fewer than N devices the run is refused unless
    ONSSEN_BENCH_ONE_DEVICE=1 (harness self-test: every rank on cuda:0 over gloo, launch-per-step kernels -- two processes'
    persistent launches on one device would starve each other's exchange groups)."""
    import socket
    import subprocess
    env = dict(os.environ)
    one_dev = env.get("ONSSEN_BENCH_ONE_DEVICE") == "1"
    have = torch.cuda.device_count()
    if have < n and not one_dev:
        print(f"bench.py: --gpus {n} but this node shows {have} device(s); ONSSEN_BENCH_ONE_DEVICE=1 runs the {n} ranks on "
              "cuda:0 as a harness self-test", file=sys.stderr)
        return 2
    if one_dev:
        env.setdefault("ONSSEN_XCD", "0")
        env.setdefault("ONSSEN_DC_PERSISTENT", "0")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = launch_command(n, sys.argv[1:], port)
    print("bench.py: launching " + " ".join(cmd[1:8]) + " ...", file=sys.stderr)
    return subprocess.call(cmd, env=env)


def launch_command(n, argv, port):
    """The driver's own command line for N ranks (one process per GPU of ONE node, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="dc_l2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="override chunks per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs legs (other BASELINE configs, B = 1 latency, training step)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-pipeline", action="store_true", help="deep clustering, 2 layers, <= 32 chunks: time the one-batch-at-a-time step instead "
                    "of the two-batch software pipeline (separation.DCPipeline)")
    ap.add_argument("--preheat", type=float, default=0.3, help="seconds of untimed graph replays before the W warm-up steps (sustained clocks)")
    ap.add_argument("--dp-deadline", type=float, default=300.0, help="N > 1: seconds the data-parallel training leg may take before it is abandoned")
    ap.add_argument("--precision", default=os.environ.get("ONSSEN_PRECISION", "bf16x3"), choices=["f32", "bf16x3", "bf16"],
                    help="f32 = exact-fp32 MFMA; bf16x3 = split-bf16 (3 bf16 MFMAs per fp32 product, fp32 accumulate); "
                         "bf16 = opt-in plain bf16 products (fp32 accumulate), outside the 1e-4 parity contract")
    args = ap.parse_args()

    os.environ["ONSSEN_PRECISION"] = args.precision
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:      # plain `python bench.py --gpus N`: become the launcher
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE = {world} ranks")
    import datetime
    import torch.distributed as dist
    # ONSSEN_BENCH_ONE_DEVICE=1 (harness self-test on a 1-GPU box): every rank uses cuda:0 and the gloo backend
    one_dev = os.environ.get("ONSSEN_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from onssen_amd.dist import init_process_group      # RCCL with its channel count capped beside the persistent recurrences
        kw = dict(rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
        if one_dev:
            init_process_group("gloo", **kw)
        else:
            init_process_group("nccl", device_id=dev, **kw)

    from onssen_amd import nn as onn
    from onssen_amd.features import mask_istft, stft_logmag
    from onssen_amd.hip import get_lib
    from onssen_amd.synthetic import make_state_dict, synth_batch
    get_lib()   # fail loudly if the HIP library is missing

    kind, H, L, B = CONFIGS[args.config]
    if args.batch:
        B = args.batch
    wl = build_workload(args.config, B, dev, rank)
    kind, H, L, F, D = wl["kind"], wl["H"], wl["L"], wl["F"], wl["D"]
    global SR, NFFT, HOP, T_FRAMES, N_SAMPLES
    SR, NFFT, HOP, T_FRAMES, N_SAMPLES = wl["SR"], wl["NFFT"], wl["HOP"], wl["T"], wl["N"]
    model, wav, wav_np, bin_masks, sd, step = wl["model"], wl["wav"], wl["wav_np"], wl["bin_masks"], wl["sd"], wl["step"]

    with torch.no_grad():
        run, graph = capture(step, not args.no_graph)
        # ---- the headline step as a STREAM of batches (round 6): deep clustering, two layers, <= 32 chunks run through the two-batch
        #      software pipeline -- replay k finishes batch k-1 (layer 1, fc_dc, 2-means, masks, iSTFT) and starts batch k (STFT, target
        #      map, layer 0, layer 1's projection), the two recurrences in ONE persistent launch on 16-row groups.  Every replay does the
        #      complete work of one batch; the one-batch-at-a-time step (`run_single`) is timed beside it, outside the timed region.
        pipe, run_single, pipe_note = None, run, None
        if (kind == "deep_clustering" and L == 2 and B <= 32 and H <= 640 and args.precision == "bf16x3" and not args.no_pipeline
                and os.environ.get("ONSSEN_XCD", "1") == "1"):
            from onssen_amd.separation import DCPipeline
            try:
                pipe = DCPipeline(model, B, N_SAMPLES, NFFT, HOP, graph=not args.no_graph)
                pipe.push(wav, check=False)             # batch 0 in flight; from here the parities alternate on resident inputs
                pipe.wav[1].copy_(wav)
                run = pipe.replay
            except RuntimeError as e:
                pipe, pipe_note = None, f"{e}"[:200]

        # untimed pre-heat (not one of the W warm-up steps): a fresh process has just spent seconds importing, packing and capturing
        # with the GPU idle in between, and K = 20 steps are a 40 ms window -- measured on one box 2.20 ms per step in such a window
        # against 1.89-1.98 for every later leg of the same process.  0.3 s of replays bring the device to its sustained clocks first.
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.preheat:
            for _ in range(10):
                run()
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            run()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        per_rank_ms = [1e3 * elapsed / args.steps]
        if world > 1:
            te = torch.tensor([elapsed], device="cpu" if one_dev else dev, dtype=torch.float64)
            every = [torch.empty_like(te) for _ in range(world)]
            dist.all_gather(every, te)                     # every rank's own clock: a straggler shows up by rank
            per_rank_ms = [1e3 * float(t.item()) / args.steps for t in every]
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            elapsed = float(te.item())

        # ---- the clustering's work depends on the data: Lloyd passes the timed replays actually ran, and the SAME captured step
        #      on a SECOND set of B distinct utterances (other seeds), outside the timed region
        lloyd, second = None, None
        single = None
        if kind == "deep_clustering" and rank == 0:
            def lloyd_now():       # the pipeline clusters batch k-1 in the workspace of the OTHER parity than the step it has just run
                return lloyd_iterations(B, T_FRAMES, F, D, ws=pipe.cws[pipe.count & 1] if pipe is not None else None)

            def set_inputs(w):
                wav.copy_(w)
                if pipe is not None:
                    pipe.wav[0].copy_(w); pipe.wav[1].copy_(w)
            lloyd = lloyd_now()
            try:
                keep = wav.clone()
                set_inputs(torch.from_numpy(synth_batch(77, B, N_SAMPLES, SR)).to(dev))
                ms2 = time_replays(run, 20, warm=100)      # (the host has just synthesised the new batch with the GPU idle: back to sustained clocks first)
                second = {"inputs": f"{B} distinct synthetic utterances (seeds 77000..)", "ms_per_step": ms2,
                          "x_real_time": B * (T_FRAMES * HOP / SR) / ms2 * 1e3, "lloyd_iterations": lloyd_now()}
                set_inputs(keep)
                run(); run()
            except Exception as e:
                second = {"error": f"{type(e).__name__}: {e}"[:200]}
            if pipe is not None:
                ms1 = time_replays(run_single, 20, warm=40)
                single = {"ms_per_step": ms1, "x_real_time": B * (T_FRAMES * HOP / SR) / ms1 * 1e3,
                          "what": "the same batch one at a time (separation.separate_dc's launches in one hipGraph: STFT -> layer 0 with its fused "
                                  "projection -> layer 1 -> fc_dc -> 2-means -> iSTFT, stacked 8-row recurrence groups on all 8 XCDs): the step "
                                  "of rounds 4-6a, and the per-batch latency floor; the pipelined step above trades one batch of latency for "
                                  "the 16-row groups' cost per time step"}

        # ---- the step of rounds 1-3 (binary masks resident in HBM instead of the clustering): a named secondary, outside
        #      the timed region, rank 0 only; captured and replayed like the headline
        dc_e2e = None
        if kind == "deep_clustering" and rank == 0:
            run_rm, g_rm = capture(wl["step_resident_masks"], not args.no_graph)
            ms_rm = time_replays(run_rm, 10)
            dc_e2e = {"ms_per_step": ms_rm, "x_real_time": B * (T_FRAMES * HOP / SR) / ms_rm * 1e3,
                      "launch": "hipGraph replay" if g_rm is not None else "eager",
                      "what": "waveform -> STFT -> BLSTM -> embedding; mask-apply + iSTFT with RESIDENT random binary masks "
                              "(no clustering: the headline step of rounds 1-3)"}

        # ---- per-kernel timing leg (HIP events on the launch stream), outside the timed region
        roof = (kernel_roofline(model.chimera if kind == "phase_net" else model, wav, dev,
                                "chimera" if kind == "phase_net" else kind, F, H, L, B, D, pipe=pipe) if rank == 0 else None)

    ms_per_step = 1e3 * elapsed / args.steps
    frames = B * T_FRAMES * world * args.steps
    audio_s = B * world * args.steps * (T_FRAMES * HOP / SR)
    result = {
        "metric": "real_time_factor",
        "value": audio_s / elapsed,
        "unit": "audio-seconds separated per wall-second (x real time), whole job",
        "frames_per_s": frames / elapsed,
        "sep_hours_per_s": audio_s / elapsed / 3600.0,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "per_rank_ms_per_step": per_rank_ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32": "f32", "bf16": "bf16 products, fp32 accumulate / gates / state (opt-in, below the 1e-4 parity contract)",
                  "bf16x3": "f32 as split-bf16 (bf16x3 MFMA, fp32 accumulate)"}[args.precision],
        "data": "synthetic",
        "config": {"workload": f"wsj0-2mix-style {kind} ({args.config}): {L}xBLSTM-{H}, F={F}, D={D}, {SR // 1000} kHz STFT "
                               f"{NFFT}/{HOP}, {B} x {T_FRAMES}-frame chunks per GPU; step = STFT+log-mag -> BLSTM "
                               + {"deep_clustering": "-> fc_dc + L2-normalise -> threshold (max - 40 dB) + 2-means on the active bins' "
                                                     "embeddings (device, <= 20 Lloyd iterations) -> binary masks -> mask-apply + iSTFT "
                                                     "(2 speakers)",
                                  "chimera": "-> fc_dc + L2-normalise, fc_mi + sigmoid -> mask-apply + iSTFT (2 speakers)",
                                  "phase_net": "-> embedding + mask heads -> phase BLSTM over (masked magnitude, phase) "
                                               "-> unit-norm phase head -> mask-apply + iSTFT (2 speakers)"}[kind],
                   "chunks_per_gpu": B, "frames_per_chunk": T_FRAMES, "launch": "hipGraph replay" if graph else "eager",
                   "inputs": f"{min(B, 64)} distinct synthetic 2-speaker mixtures per rank (seeds 1000 (1 + rank) + u)" + (" tiled to the batch" if B > 64 else "")
                             + ", random-init weights (seed 0)",
                   "untimed_preheat_s": args.preheat,
                   "parallelism": f"utterance-sharded x{world}, no data-path collective"},
    }
    from onssen_amd.nn._core import _XcdStatus, recurrence_plan
    torch.cuda.synchronize()
    _XcdStatus.poll(wait=True)     # raises if a persistent launch aborted
    if pipe is not None:
        result["config"]["pipeline"] = {
            "depth": 2, "batch_latency_ms": 2 * ms_per_step,
            "what": "separation.DCPipeline: a stream of 32-chunk batches, one per step; step k = [STFT, target map, layer-0 projection of "
                    "batch k] -> ONE persistent launch [layer 1 of batch k-1 on 4 XCDs || layer 0 of batch k on the other 4, 16-row groups] "
                    "-> [layer-1 projection of batch k; fc_dc, 2-means, masks, iSTFT of batch k-1].  Every step carries the complete work of "
                    "one batch (the timed region is K steady-state steps); results are bit-identical to separate_dc on 16-row groups "
                    "(tests/test_gpu_pipeline.py)"}
        st = pipe.ws[1120:1132].cpu().view(torch.int32)
        if int(st[0]) != 0 or int(st[2]) != 0:
            raise SystemExit(f"pipelined persistent recurrence aborted / saw non-finite activations during the timed region (words {st.tolist()})")
        _XcdStatus.safe_protocol_seen |= int(st[1]) == 1
        for cw in pipe.cws:
            word = int(cw[pipe.cstat:pipe.cstat + 4].cpu().view(torch.int32)[0])
            if word != 0:
                raise SystemExit(f"persistent 2-means launch of the pipeline gave up a bounded wait (status {word & 0xffffffff:#x})")
    elif pipe_note:
        result["config"]["pipeline"] = {"depth": 1, "why_not": pipe_note}
    if roof is not None and roof.get("legs_ms"):
        legs = sum(v for v in roof["legs_ms"].values() if v)
        roof["legs_sum_ms"] = legs
        # (each leg is timed by itself, four calls back to back in its own graph; inside the step's graph the 13 launches' tails and
        #  ramp-ups overlap by a few microseconds each: the pipelined step measured 1.555 ms against 1.589 ms of legs -- 3 % allowed)
        roof["legs_le_step"] = bool(legs <= ms_per_step * 1.03)
    for mod in (model, getattr(model, "chimera", None)):      # the graph replays' own status words
        for buf in (mod._ws.cache.values() if mod is not None else ()):
            st = buf[1120:1132].cpu().view(torch.int32)
            if int(st[0]) != 0 or int(st[2]) != 0:
                raise SystemExit(f"persistent recurrence aborted / saw non-finite activations during the timed region (words {st.tolist()})")
            _XcdStatus.safe_protocol_seen |= int(st[1]) == 1
    if kind == "deep_clustering":     # ... and the persistent Lloyd launch's: a bounded wait that gave up inside a graph replay would
        from onssen_amd import separation      # otherwise pass unnoticed (the replays do not post their status words)
        from onssen_amd.hip import get_lib as _gl
        for key, buf in separation._CLUSTER_WS.items():
            off = int(_gl().dll.onssen_dc_cluster_status_offset(key[1], key[4]))
            word = int(buf[off:off + 4].cpu().view(torch.int32)[0])
            if word != 0:
                raise SystemExit(f"persistent 2-means launch gave up a bounded wait during the timed region (status {word & 0xffffffff:#x})")
    result["config"]["recurrence"] = ("XCD-local persistent kernel (one launch per layer)" if recurrence_plan(B, H)[1] & 4
                                      else "one launch per time step")
    result["config"]["xcd_placement_independent_protocol_used"] = _XcdStatus.safe_protocol_seen
    if world > 1 and not args.no_extra:
        # ---- data-parallel training leg (the only path with a collective): every rank takes part.  The headline above is
        #      already final; a watchdog makes sure a wedged collective can only cost this leg, never the JSON line.
        import threading

        def give_up():
            if rank == 0:
                result["roofline"] = roof
                result["dp_training_step_dc_l3_b16"] = {"error": f"did not finish within {args.dp_deadline} s (a rank stuck in a collective?)"}
                print(json.dumps(result), flush=True)
            os._exit(0)                  # (the error is IN the line; peers that are stuck leave through their own watchdog)
        dog = threading.Timer(args.dp_deadline, give_up)
        dog.daemon = True
        dog.start()
        try:
            dist.barrier()               # rank 0 comes from its per-kernel timing leg: start the steps together
            dp = dp_training_leg(dev, rank, world, one_dev, **(dict(steps=2, warmup=1) if one_dev else {}))
        except Exception as e:
            dp = {"error": f"rank {rank}: {type(e).__name__}: {e}"[:400]}
        dog.cancel()
        result["dp_training_step_dc_l3_b16"] = dp
    if world > 1:
        # ---- per-rank health of the persistent kernels and the collective layer's settings IN FORCE (every rank reports; rank 0 prints):
        #      the first hardware N = 8 run must be able to tell "a rank fell back / aborted" from "RCCL got more channels than the two
        #      spare CUs per XCD hold" from the line alone
        from onssen_amd.nn._core import _XcdPolicy
        mine = torch.tensor([int(_XcdStatus.safe_protocol_seen), _XcdPolicy.aborts, _XcdPolicy.recovered, _XcdPolicy.persistent_launches,
                             _XcdPolicy.fallback_launches, int(os.environ.get("NCCL_MAX_NCHANNELS", "-1")), torch.cuda.current_device()],
                            device="cpu" if one_dev else dev, dtype=torch.int64)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        keys = ("xcd_placement_independent_protocol_used", "persistent_launch_aborts", "calls_rerun_after_abort", "persistent_stack_launches",
                "launch_per_step_stack_launches", "NCCL_MAX_NCHANNELS", "device_index")
        result["per_rank_health"] = [dict(zip(keys, (int(v) for v in e.tolist()))) for e in every]
        result["collective_layer"] = {"backend": "gloo (ONSSEN_BENCH_ONE_DEVICE harness self-test)" if one_dev else "nccl = RCCL over xGMI",
                                      "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"), "NCCL_MIN_NCHANNELS": os.environ.get("NCCL_MIN_NCHANNELS"),
                                      "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                                      "why": "one RCCL workgroup per channel; <= 16 channels = <= 2 workgroups per XCD = the CUs a 30-member "
                                             "recurrence group leaves free (onssen_amd/dist.py: configure_rccl)"}
    if rank == 0:
        result["roofline"] = roof
        if dc_e2e is not None:
            result["resident_mask_step"] = dc_e2e
        if lloyd is not None:
            result["lloyd_iterations"] = lloyd
        if second is not None:
            result["second_input_set"] = second
        if single is not None:
            result["one_batch_at_a_time_step"] = single
        from onssen_amd import options as _opt
        result["config"]["non_default_options"] = {k: v["value"] for k, v in _opt.describe().items() if v["value"] != v["default"] and k != "world_size"}
        if world == 1 and not args.no_extra and args.config == "dc_l2" and args.precision == "bf16x3":
            result["extra_configs"] = extra_configs(dev)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(sd, kind, wav_np, bin_masks.cpu().numpy())
        print(json.dumps(result), flush=True)
    if world > 1:
        if "error" in result.get("dp_training_step_dc_l3_b16", {}):
            os._exit(0)  # this rank left the leg early: peers may sit in a collective it will never join -- do not wait for them
        dist.barrier()   # rank 0 finishes its per-kernel timing leg before anyone tears the group down
        dist.destroy_process_group()


def kernel_roofline(model, wav, dev, kind, F, H, L, B, D, recurrence_only=False, pipe=None):
    """Roofline block for the dominant kernel, timed live with HIP events on the launch stream
    (each call captured in its own hipGraph and replayed, so host launch cost is excluded).

    The recurrence and the GEMMs are priced against the MFMA rate of their arithmetic: dense bf16 / 3 for the
    split-bf16 form, the exact-fp32 MFMA rate otherwise.  XCD-local persistent form: lstm_xcd_kernel is ONE launch
    per layer (and 64 batch rows) whose duration is the event span of a single-layer call minus its input-projection
    GEMM and input split.  Launch-per-step form: the span of the T dependent launches / T (it includes the
    dependent-launch gap the serial chain really pays)."""
    from onssen_amd.features import stft_logmag
    from onssen_amd import _abi
    from onssen_amd.hip import get_lib
    from onssen_amd.nn._core import _stream, recurrence_plan
    lib = get_lib()
    T = T_FRAMES
    ug, flags = recurrence_plan(B, H)
    pk = model._packed.get(ug)
    Hp, NP = pk.Hp, pk.NP
    x3 = bool(flags & 2)
    bf16_only = bool(flags & _abi.BLSTM_BF16)          # opt-in plain bf16 products
    ebf = _abi.EPI_BF16 if bf16_only else 0
    images = bool(flags & _abi.BLSTM_XCD) and x3   # activations travel as x3 images, GEMMs run on pre-split operands
    wih = pk.wih_img if images else pk.wih_x3 if x3 else pk.wih
    whh = pk.whh_x3 if x3 else pk.whh

    def lin(A, a_s0, a_s1, K, Wf, ldf, W3, ld3, bias, N, mode, group, out_ptr, c_s0, c_s1):
        if x3:
            lib.linear_bf16x3(A, a_s0, a_s1, B, T * B, K, W3, ld3, bias, N, mode, group, 1e-12, None, out_ptr, c_s0, c_s1,
                              st())
        else:
            lib.linear(A, a_s0, a_s1, B, T * B, K, Wf, ldf, bias, N, mode, group, 1e-12, None, out_ptr, c_s0, c_s1, st())
    y = torch.empty(T, B, 2, Hp, device=dev)
    ws = torch.zeros(lib.blstm_workspace_bytes(B, T, max(F, 2 * Hp), H, 1, ug), dtype=torch.uint8, device=dev)
    xin = torch.randn(B, T, F, device=dev)
    yin = torch.randn(T, B, 2 * Hp, device=dev).tanh_()
    st = _stream   # evaluated at call time: under graph capture the current stream is the capture stream
    lyr = 1 if L > 1 else 0          # time a layer with the steady-state shape (in = 2H) when there is one
    # XCD form: like the stack, the layer leaves only its x3 image (no fp32 rows: the heads read the image)
    y_ptr = None if images else y.data_ptr()

    def layer():
        if lyr == 0:
            lib.blstm_forward(xin.data_ptr(), xin.stride(0), xin.stride(1), B, T, F, H, 1, ug, [wih[0].data_ptr()],
                              [whh[0].data_ptr()], [pk.bias[0].data_ptr()], y_ptr, ws.data_ptr(),
                              ws.numel(), flags, st())
        else:   # feed y-shaped input through layer 1's weights: x (B,T,2Hp) strides of the time-major buffer
            lib.blstm_forward(yin.data_ptr(), 2 * Hp, B * 2 * Hp, B, T, 2 * Hp, H, 1, ug, [wih[1].data_ptr()],
                              [whh[1].data_ptr()], [pk.bias[1].data_ptr()], y_ptr, ws.data_ptr(),
                              ws.numel(), flags, st())

    gbuf = ws[_abi.BLSTM_WS_HEADER:]
    F4, F32 = (F + 3) // 4 * 4, (F + 31) // 32 * 32
    K1, K132 = 2 * Hp, (2 * Hp + 31) // 32 * 32
    hd = model._head_dc if kind == "chimera" else model._head
    hp = hd.get(Hp)
    out = torch.empty(B, T, hp.N, device=dev)
    if images:
        img_x = torch.empty(T * B, F32 // 32, 2, 32, device=dev, dtype=torch.int16)
        img_y = torch.empty(T * B, K132 // 32, 2, 32, device=dev, dtype=torch.int16)
        lib.x3_image(yin.data_ptr(), B * 2 * Hp, 2 * Hp, B, T * B, K1, img_y.data_ptr(), st())

        def image_in():    # the single-layer call below splits its fp32 input first; in the stack only layer 0 does
            if lyr == 0:
                lib.x3_image(xin.data_ptr(), xin.stride(1), xin.stride(0), B, T * B, F, img_x.data_ptr(), st())
            else:
                lib.x3_image(yin.data_ptr(), B * 2 * Hp, 2 * Hp, B, T * B, K1, img_y.data_ptr(), st())

        def gemm0():       # layer 0 as the stack runs it: split the features, then the GEMM
            lib.x3_image(xin.data_ptr(), xin.stride(1), xin.stride(0), B, T * B, F, img_x.data_ptr(), st())
            lib.linear_x3p(img_x.data_ptr(), T * B, F, pk.wih_img[0].data_ptr(), pk.bias[0].data_ptr(), 2 * NP, ebf, 0, 1e-12,
                           gbuf.data_ptr(), B, B * 2 * NP, 2 * NP, st())

        def gemm_in():
            if lyr == 0:
                gemm0()
            else:
                lib.linear_x3p(img_y.data_ptr(), T * B, K1, pk.wih_img[1].data_ptr(), pk.bias[1].data_ptr(), 2 * NP, ebf, 0,
                               1e-12, gbuf.data_ptr(), B, B * 2 * NP, 2 * NP, st())

        def head():
            lib.linear_x3p(img_y.data_ptr(), T * B, K1, hp.img.data_ptr(), hp.b.data_ptr(), hp.N, 1 | ebf, D, 1e-12,
                           out.data_ptr(), B, hp.N, T * hp.N, st())
    else:
        def image_in():
            pass

        def gemm0():
            lin(xin.data_ptr(), xin.stride(1), xin.stride(0), F, pk.wih[0].data_ptr(), F4, pk.wih_x3[0].data_ptr(), F32,
                pk.bias[0].data_ptr(), 2 * NP, 0, 0, gbuf.data_ptr(), B * 2 * NP, 2 * NP)

        def gemm_in():
            if lyr == 0:
                gemm0()
            else:
                lin(yin.data_ptr(), B * 2 * Hp, 2 * Hp, K1, pk.wih[1].data_ptr(), K1, pk.wih_x3[1].data_ptr(), K132,
                    pk.bias[1].data_ptr(), 2 * NP, 0, 0, gbuf.data_ptr(), B * 2 * NP, 2 * NP)

        def head():
            lin(yin.data_ptr(), B * 2 * Hp, 2 * Hp, K1, hp.w.data_ptr(), K1, hp.planes.data_ptr(), hp.ld3, hp.b.data_ptr(),
                hp.N, 1, D, out.data_ptr(), hp.N, T * hp.N)

    def timed(fn, reps=5, inner=4):
        """Seconds per call: `inner` calls captured back to back in ONE hipGraph (kernel boundaries like inside the step's own
        graph; a graph launch per call would add its ~5-10 us to every short kernel), replayed `reps` times between events."""
        fn()
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(inner):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (reps * inner) * 1e-3

    only_rec = recurrence_only and bool(flags & _abi.BLSTM_XCD)      # (the batch sweep times the persistent launch directly)
    t_layer, t_gin, t_g0, t_head = (0.0,) * 4 if only_rec else (timed(layer), timed(gemm_in), timed(gemm0), timed(head))
    t_img = timed(image_in) if images and lyr and not only_rec else 0.0         # (for lyr == 0 the split is part of gemm0 = gemm_in)
    if flags & _abi.BLSTM_XCD:
        # the recurrence kernel BY ITSELF: the same call with ONSSEN_BLSTM_G_READY -- G is what `layer` left in the workspace,
        # no split, no GEMM.  (Rounds 1-2 reported t_layer - t_gin - t_img, which overstated the kernel by ~6 %.)
        def rec_only():
            a = (xin, xin.stride(0), xin.stride(1), F, 0) if lyr == 0 else (yin, 2 * Hp, B * 2 * Hp, 2 * Hp, 1)
            lib.blstm_forward(a[0].data_ptr(), a[1], a[2], B, T, a[3], H, 1, ug, [wih[a[4]].data_ptr()], [whh[a[4]].data_ptr()],
                              [pk.bias[a[4]].data_ptr()], y_ptr, ws.data_ptr(), ws.numel(), flags | _abi.BLSTM_G_READY, st())
        layer()
        t_rec = timed(rec_only)
    else:
        t_rec = t_layer - t_gin - t_img                        # launch-per-step forms: by difference
    if recurrence_only:                                        # the batch sweep: the dominant kernel's numbers only
        flop_rec = 2.0 * 2 * B * 4 * H * H * T
        xcd = bool(flags & _abi.BLSTM_XCD)
        launches = -(-B // 64) if xcd else T
        peak = BF16_MFMA_PEAK_TFLOPS if bf16_only else BF16_MFMA_PEAK_TFLOPS / 3.0 if x3 else FP32_MFMA_PEAK_TFLOPS
        return {"achieved": flop_rec / t_rec / 1e12, "peak": peak, "frac": flop_rec / t_rec / 1e12 / peak,
                "us_per_time_step": t_rec / T * 1e6, "launches_per_step": launches * L}
    # does the step fuse the first layer's projection into its recurrence launch (run_blstm's rule)?  Then layer 0 is ONE call:
    # the feature split + the fused recurrence, and the layer-0 GEMM below is not part of the step
    from onssen_amd import options
    fuse_env = options.get("fuse_first_layer")
    fused0 = bool(images and pk.wih_frag0 is not None and (fuse_env == "1" or (fuse_env == "auto" and B > 16)))
    t_l0f = None
    if fused0:
        fl = flags | _abi.BLSTM_FUSE_IN0 | (_abi.BLSTM_FUSE_TAIL if pk.bias0_tail is not None else 0)
        b0 = pk.bias0_tail if pk.bias0_tail is not None else pk.bias[0]

        def layer0_fused():
            lib.blstm_forward(xin.data_ptr(), xin.stride(0), xin.stride(1), B, T, F, H, 1, ug, [pk.wih_frag0.data_ptr()],
                              [whh[0].data_ptr()], [b0.data_ptr()], y_ptr, ws.data_ptr(), ws.numel(), fl, st())
        t_l0f = timed(layer0_fused)
    # front / back end by themselves (HBM-bound rows of SURVEY 8d)
    from onssen_amd.features import mask_istft
    lm_ri = stft_logmag(wav, NFFT, HOP)
    mk = torch.rand(B, T, F, 2, device=dev)
    t_stft = timed(lambda: stft_logmag(wav, NFFT, HOP))
    t_istft = timed(lambda: mask_istft(lm_ri[1], mk, HOP, N_SAMPLES))
    t_cluster, dc_legs = None, None
    if kind == "deep_clustering":          # the clustering back end on the step's own embeddings (its time depends on the data)
        from onssen_amd.nn._core import heads_take_image, run_blstm
        from onssen_amd.separation import dc_masks
        if images and heads_take_image(B, H, (D,)) and os.environ.get("ONSSEN_DC_COMPACT", "1") == "1":
            # the compacted form (what the step runs): threshold -> target map | fc_dc GEMM scattering the active rows | cluster
            nbk, comp_off, dest_off = lib.dc_compact_layout(B, T, F, D)
            wsk = torch.zeros(nbk, dtype=torch.uint8, device=dev)
            yk = run_blstm(model._packed, model._ws, lm_ri[0], need_y=False)
            wsy, offy = yk.x3_image
            mk2 = torch.empty(B, T, F, 2, device=dev)
            lmk = lm_ri[0].contiguous()

            def k_index():
                lib.dc_index(lmk.data_ptr(), B, T, F, D, 40.0, wsk.data_ptr(), nbk, st())

            def k_head():
                lib.linear_x3p_compact(wsy.data_ptr() + offy, T * B, K1, hp.img.data_ptr(), hp.b.data_ptr(), hp.N, D, 1e-12,
                                       wsk.data_ptr() + dest_off, T * F, F, wsk.data_ptr() + comp_off, B, T * F * D, bf16_only, st())

            def k_index_cluster():
                k_index()
                lib.dc_cluster_compact(B, T, F, D, 20, mk2.data_ptr(), wsk.data_ptr(), nbk, st())
            k_index(); k_head()
            t_idx, t_headc = timed(k_index), timed(k_head)
            t_cl = timed(k_index_cluster) - t_idx
            n_act = int(wsk[int(lib.dll.onssen_dc_cluster_status_offset(B, D)) - B * 72 * 4:int(lib.dll.onssen_dc_cluster_status_offset(B, D))]
                        .view(torch.int32).view(B, 72)[:, 64].sum())
            dc_legs = {"threshold_target_map": t_idx * 1e3, "fc_dc_l2norm_active_rows_only": t_headc * 1e3,
                       "init_lloyd_masks": t_cl * 1e3, "active_bin_fraction": n_act / float(B * T * F)}
            t_cluster = t_idx + t_cl + (t_headc - t_head)      # what the back end adds to the step that had resident masks
        else:
            emb_k, = model([lm_ri[0]])
            t_cluster = timed(lambda: dc_masks(emb_k, lm_ri[0]))
    stft_bytes = B * (N_SAMPLES * 4 + T * F * 4 * 3)           # waveform in; log-magnitude + (Re, Im) out
    istft_bytes = B * (T * F * 4 * 2 + T * F * 4 * 2 + 2 * N_SAMPLES * 4)   # (Re, Im) + two masks in; two waveforms out
    flop_rec = 2.0 * 2 * B * 4 * H * H * T                     # h W_hh^T, both directions, 2 FLOP/MAC
    flop_gin = 2.0 * B * T * 8 * H * (2 * H if lyr else F)
    flop_head = 2.0 * B * T * hp.N * 2 * H
    xcd = bool(flags & _abi.BLSTM_XCD)
    traffic, traffic_src = None, None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        tj = json.load(open(tf))
        key = "xcd_recurrence_hbm_bytes_per_launch" if xcd else "recurrence_hbm_bytes_per_launch"
        latest = tj.get("latest", {})          # written by tools/profile_round.sh (tools/traffic_from_pmc.py): names its round
        if xcd and key in latest:
            traffic, traffic_src = latest[key], f"PMC passes of round {latest.get('round')} (profiles/traffic.json: latest; {latest.get('kernel', '')})"
        else:
            traffic, traffic_src = tj.get(key), "PMC passes of round 1 (profiles/traffic.json)"
    # persistent form: one launch per layer and 64 batch rows; launch-per-step form: T launches per layer
    launches = -(-B // 64) if xcd else T
    # ceiling for ALGORITHMIC (fp32-equivalent) FLOPs: the exact-fp32 MFMA rate, or a third of the dense bf16
    # MFMA rate when every product is three bf16 MFMAs
    peak = BF16_MFMA_PEAK_TFLOPS if bf16_only else BF16_MFMA_PEAK_TFLOPS / 3.0 if x3 else FP32_MFMA_PEAK_TFLOPS
    rec = {"kernel": "lstm_xcd_kernel" if xcd else "lstm_step_kernel", "bound": "mfma",
           "achieved": flop_rec / t_rec / 1e12, "peak": peak, "unit": "TFLOP/s",
           "frac": flop_rec / t_rec / 1e12 / peak, "traffic": traffic, "traffic_source": traffic_src,
           "peak_note": "dense bf16 MFMA 2500 TF" if bf16_only else "dense bf16 MFMA 2500 TF / 3 (split-bf16)" if x3 else "exact-fp32 MFMA",
           "bound_note": "a serial chain of T dependent time steps: each is cell update -> tagged h stores -> L2 -> polled "
                         "fragment loads -> MFMAs -> LDS reduction; the exchange (every member reads all of h: 576 KB per step "
                         "and XCD at B=32) streams at the XCD's L2 -> CU bandwidth (~0.5 KB per clock), the rest is that chain's "
                         "latency -- not MFMA issue or HBM; see DESIGN.md section 3",
           "us_per_launch": t_rec / launches * 1e6, "us_per_time_step": t_rec / T * 1e6,
           "launches_per_step": launches * L, "algorithmic_flop_per_launch": flop_rec / launches,
           "unit_group": ug, "share_of_step_ms": (t_rec * (L - 1) + t_l0f if fused0 and lyr else t_rec * L) * 1e3}
    gem = {"kernel": "linear_x3q_kernel (onssen_linear_x3p; 256x320 / 256x256 tiles, LDS-DMA staging)" if images else "linear_x3_kernel" if x3 else "linear_kernel", "bound": "mfma", "peak": peak, "unit": "TFLOP/s",
           "achieved_by_call": {"input_proj_l0": 2.0 * B * T * 8 * H * F / t_g0 / 1e12,
                                "input_proj_l1": (flop_gin / t_gin / 1e12) if lyr else None,
                                "fc_dc_l2norm": flop_head / t_head / 1e12},
           "ms_by_call": {"input_proj_l0": t_g0 * 1e3, "input_proj_l1": t_gin * 1e3 if lyr else None,
                          "fc_dc_l2norm": t_head * 1e3},
           "share_of_step_ms": (t_g0 + (L - 1) * t_gin * (1 if lyr else 0) + t_head) * 1e3}
    rec["other_kernels"] = gem
    rec["recurrence_timing"] = "direct (ONSSEN_BLSTM_G_READY call, HIP events around hipGraph replays)" if xcd else "by difference"
    rec["layer_call_ms"] = t_layer * 1e3
    rec["hbm_kernels"] = {
        "stft_logmag_kernel": {"ms": t_stft * 1e3, "algorithmic_bytes": stft_bytes, "achieved_GBs": stft_bytes / t_stft / 1e9,
                               "frac_of_hbm_peak": stft_bytes / t_stft / 1e9 / HBM_PEAK_GBS},
        "mask_istft_kernel": {"ms": t_istft * 1e3, "algorithmic_bytes": istft_bytes, "achieved_GBs": istft_bytes / t_istft / 1e9,
                              "frac_of_hbm_peak": istft_bytes / t_istft / 1e9 / HBM_PEAK_GBS}}
    # every leg of the step, each timed by itself: they must sum to <= the step they decompose
    if kind == "deep_clustering" and fused0:
        rec["legs_ms"] = {"stft_logmag": t_stft * 1e3, "layer0_split_plus_fused_recurrence": t_l0f * 1e3,
                          "input_proj_deeper_layers": (L - 1) * t_gin * 1e3 if lyr else 0.0,
                          "recurrence_deeper_layers": (L - 1) * t_rec * 1e3 if lyr else 0.0,
                          "fc_dc_l2norm": t_head * 1e3, "threshold_2means_masks": t_cluster * 1e3, "mask_istft": t_istft * 1e3}
        if dc_legs:
            rec["dc_back_end_legs_ms"] = dc_legs
        rec["first_layer"] = {"fused_input_projection": True, "ms": t_l0f * 1e3, "us_per_time_step": t_l0f / T * 1e6,
                              "note": "x W_ih^T inside the recurrence launch (no G, no layer-0 GEMM); includes the 7 us feature split; "
                                      "the unfused alternative would be input_proj_l0 + one plain recurrence launch = "
                                      f"{(t_g0 + t_rec) * 1e3:.3f} ms"}
    elif kind == "deep_clustering":
        rec["legs_ms"] = {"stft_logmag": t_stft * 1e3, "input_proj_l0_with_split": t_g0 * 1e3,
                          "input_proj_deeper_layers": (L - 1) * t_gin * 1e3 if lyr else 0.0, "recurrence_all_layers": L * t_rec * 1e3,
                          "fc_dc_l2norm": t_head * 1e3, "threshold_2means_masks": t_cluster * 1e3, "mask_istft": t_istft * 1e3}
        if dc_legs:
            rec["dc_back_end_legs_ms"] = dc_legs
    if pipe is not None and dc_legs and xcd:
        # ---- the pipelined step's dominant kernel: ONE persistent launch that runs layer 1 of batch k-1 and layer 0 of batch k on
        #      16-row groups, timed by itself (ONSSEN_BLSTM_G_READY: both projections are what the last step left in the workspace)
        def pair_only():
            lib.blstm_pipe2_forward(pipe.logmag[0].data_ptr(), T * F, F, B, T, F, H, pipe.ug, [t.data_ptr() for t in pk.wih_img],
                                    [t.data_ptr() for t in pk.whh_x3], [t.data_ptr() for t in pk.bias], pipe.ws.data_ptr(), pipe.wnb,
                                    pipe.flags | _abi.BLSTM_G_READY, st())
        t_pair = timed(pair_only)
        flop_pair = 2.0 * flop_rec
        seq = dict(rec)
        seq.pop("other_kernels", None); seq.pop("hbm_kernels", None)
        rec.update({
            "kernel": "lstm_xcd_kernel<5, 8, false, 3, false> (pair launch of onssen_blstm_pipe2_forward_f32: layer 1 of batch k-1 || layer 0 of batch k)",
            "achieved": flop_pair / t_pair / 1e12, "frac": flop_pair / t_pair / 1e12 / peak,
            "us_per_launch": t_pair * 1e6, "us_per_time_step": t_pair / T * 1e6, "launches_per_step": 1,
            "algorithmic_flop_per_launch": flop_pair, "share_of_step_ms": t_pair * 1e3,
            "bound_note": "two independent serial chains of T dependent time steps in one launch, each on 4 XCDs (16-row groups: "
                          "cell update -> tagged h stores -> L2 -> polled fragment loads -> 285 MFMAs per member -> LDS reduction); "
                          "the chains' latency binds, not MFMA issue or HBM; see DESIGN.md section 3",
            "recurrence_timing": "direct (onssen_blstm_pipe2_forward_f32 with ONSSEN_BLSTM_G_READY, HIP events around hipGraph replays)",
        })
        tr = None
        if os.path.exists(tf):
            tr = json.load(open(tf)).get("latest", {}).get("xcd_pair_recurrence_hbm_bytes_per_launch")
        rec["traffic"] = tr
        rec["traffic_source"] = "PMC passes (profiles/traffic.json: latest.xcd_pair_recurrence_hbm_bytes_per_launch)" if tr else None
        rec["legs_ms"] = {"stft_logmag": t_stft * 1e3, "threshold_target_map": dc_legs["threshold_target_map"],
                          "input_proj_l0_with_split": t_g0 * 1e3, "pair_recurrence_l1_prev_l0_this": t_pair * 1e3,
                          "input_proj_l1": t_gin * 1e3, "fc_dc_l2norm_active_rows_only": dc_legs["fc_dc_l2norm_active_rows_only"],
                          "init_lloyd_masks": dc_legs["init_lloyd_masks"], "mask_istft": t_istft * 1e3}
        rec.pop("first_layer", None)
        rec["one_batch_at_a_time_form"] = seq           # the sequential step's dominant kernel (8-row stacked groups), as in rounds 2-6a
    return rec


def _host_cpu():
    """(model name, physical cores, hardware threads) of this host from /proc/cpuinfo (physical = distinct (package, core id) pairs)."""
    model, cores, threads, phys, cid = "unknown", set(), 0, None, None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("processor"):
                threads += 1
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
                cores.add((phys, cid))
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    n_phys = len(cores) or max(1, threads // 2)
    return model, min(n_phys, usable), usable


def cpu_baseline(sd, kind, wav_np, masks_np):
    """The oracle restatement (NumPy STFT/iSTFT + ATen-on-CPU network + sklearn KMeans) timed on this host's cores beside the GPU
    line, by BASELINE.md section 3's protocol (round 6): fp32, eval mode, no_grad, 3 warm-ups, median of 10 at B = 1 and at B = 32
    (T = 400), thread counts up to ALL PHYSICAL cores tried and the best one used (ATen's LSTM gets slower when every core of a big
    host is used; the counts tried and the count used are reported), the STFT / iSTFT legs threaded over the utterances.
    ``value`` = the whole hot path at B = 32 with the reference's own back end: ``KMeans(n_clusters=2, random_state=0)`` of the
    reference's era (``n_init=10``, the default until sklearn 1.4); the same with sklearn >= 1.4's ``n_init='auto'`` (one k-means++
    run: a tenth of the clustering work) is reported beside it.  Passes are capped by time (at least 3) so that the default bench
    stays within minutes; the counts actually run are in the record."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import np_oracle as O
    from oracle import torch_cpu as TC
    n_thr = torch.get_num_threads()
    cpu_model, n_phys, n_usable = _host_cpu()
    audio_s = T_FRAMES * HOP / SR
    fwd = {"chimera": TC.chimera_forward, "phase_net": TC.phase_net_forward}.get(kind, TC.deep_clustering_forward)
    thr_set = sorted({t for t in (4, 8, 16, 32, 64, n_phys) if t <= n_phys} | {min(n_phys, 8)})
    pool = ThreadPoolExecutor(max_workers=min(n_phys, 32))

    def batch_of(nb):
        w = wav_np[:min(len(wav_np), nb)]
        if len(w) < nb:
            w = np.concatenate([w] * (nb // len(w) + 1))[:nb]
        return w

    def whole_path(wav_s, n_init):
        """One pass of the whole path on ``wav_s`` utterances; returns per-leg seconds."""
        nb = len(wav_s)
        t0 = time.perf_counter()
        X = list(pool.map(lambda w: O.stft(w, NFFT, HOP), wav_s))
        lm = np.stack([O.log_magnitude(x) for x in X])
        t1 = time.perf_counter()
        if kind == "chimera":
            _, a, b = TC.chimera_forward(sd, lm)
            mk = np.stack([a.numpy(), b.numpy()], 1)
            t2 = t3 = time.perf_counter()
        elif kind == "phase_net":
            ph = np.stack([np.stack([x.real, x.imag], -1) for x in X]).astype(np.float32)
            _, a, b, _, _ = TC.phase_net_forward(sd, lm, ph)
            mk = np.stack([a.numpy(), b.numpy()], 1)
            t2 = t3 = time.perf_counter()
        else:                                    # the reference's own back end: sklearn KMeans on the active bins (evaluate.py:36-41)
            from sklearn.cluster import KMeans
            emb = TC.deep_clustering_forward(sd, lm).numpy()
            t2 = time.perf_counter()
            mk = np.zeros((nb, 2) + lm.shape[1:], np.float64)
            for i in range(nb):                  # one utterance at a time, like the reference's evaluation loop (batch 1)
                act = O.dc_active_bins(lm[i])
                lab = KMeans(n_clusters=2, random_state=0, n_init=n_init).fit_predict(emb[i][act])
                mk[i] = O.dc_binary_masks(lm[i], lab)
            t3 = time.perf_counter()
        list(pool.map(lambda i: O.mask_istft(X[i], mk[i], HOP, N_SAMPLES), range(nb)))
        t4 = time.perf_counter()
        return {"stft": t1 - t0, "network": t2 - t1, "cluster": t3 - t2, "istft": t4 - t3, "total": t4 - t0}

    def median_of(fn, warm, reps, cap_s):
        for _ in range(warm):
            fn()
        rows, t_start = [], time.perf_counter()
        while len(rows) < reps and (len(rows) < 3 or time.perf_counter() - t_start < cap_s):
            t0 = time.perf_counter()
            r = fn()
            rows.append((time.perf_counter() - t0, r))
        rows.sort(key=lambda x: x[0])
        return rows[len(rows) // 2][0], rows[len(rows) // 2][1], len(rows)

    # ---- network only: best thread count up to all physical cores, then median of 10 at that count
    net, best_thr = {}, {}
    if kind != "phase_net":
        for nb, cap in ((1, 3.0), (32, 8.0)):
            lm_b = np.stack([O.log_magnitude(O.stft(w, NFFT, HOP)) for w in batch_of(nb)])
            scan = {}
            for thr in thr_set:
                torch.set_num_threads(thr)
                scan[thr] = median_of(lambda: fwd(sd, lm_b), 1, 3, 0.0)[0]
            thr = min(scan, key=scan.get)
            torch.set_num_threads(thr)
            m, _, n_p = median_of(lambda: fwd(sd, lm_b), 3, 10, cap)
            best_thr[nb] = thr
            net[f"B{nb}"] = {"ms": 1e3 * m, "x_real_time": nb * audio_s / m, "passes": n_p, "threads": thr,
                             "threads_tried": {str(k): round(1e3 * v, 2) for k, v in scan.items()}}
    # ---- the whole path: B = 1 and B = 32 at the network's best thread count; the era-faithful KMeans(n_init=10) and n_init='auto'
    whole = {}
    n_inits = (10, "auto") if kind == "deep_clustering" else (10,)
    for nb, cap in ((1, 3.0), (min(32, max(len(wav_np), 1)), 24.0)):
        torch.set_num_threads(best_thr.get(nb if nb == 1 else 32, min(n_phys, 16)))
        ws = batch_of(nb)
        for n_init in n_inits:
            m, legs, n_p = median_of(lambda: whole_path(ws, n_init), 1 if nb > 1 else 3, 10, cap if n_init == 10 else cap / 2)
            whole[f"B{nb}" + ("" if n_init == 10 else "_kmeans_auto")] = {
                "ms": 1e3 * m, "x_real_time": nb * audio_s / m, "frames_per_s": nb * T_FRAMES / m, "passes": n_p,
                "threads": torch.get_num_threads(), "legs_ms": {k: round(1e3 * v, 2) for k, v in legs.items()}}
    torch.set_num_threads(n_thr)
    pool.shutdown()
    nb_big = min(32, max(len(wav_np), 1))
    head = whole[f"B{nb_big}"]
    return {"value": head["x_real_time"], "unit": "audio-seconds separated per wall-second (x real time)",
            "frames_per_s": head["frames_per_s"], "cores": head["threads"], "cores_physical": n_phys, "threads_used": head["threads"],
            "host_threads": n_usable, "cpu_model": cpu_model, "kind": "port", "kmeans_n_init": 10 if kind == "deep_clustering" else None,
            "protocol": "BASELINE.md section 3: fp32, eval, no_grad, warm-ups then median of up to 10 passes (time-capped, >= 3), B = 1 and "
                        "B = 32 x 400 frames, thread counts up to all physical cores scanned on the network and the best one used",
            "sample": f"all {nb_big} chunks of the batch per pass, median of {head['passes']} passes: NumPy fp64 STFT / iSTFT threaded over the "
                      f"utterances + ATen/oneDNN fp32 network on {head['threads']} of {n_phys} physical cores"
                      + ("; sklearn KMeans(2, random_state=0, n_init=10) per utterance on the active bins' embeddings like "
                         "egs/wsj0-2mix/deep_clustering/evaluate.py:36-41" if kind == "deep_clustering" else ""),
            "whole_path": whole, "network_only": net}


if __name__ == "__main__":
    main()
