#!/usr/bin/env python
"""GPU box: does the path SEPARATE?  (round 5; VERDICT r04 "missing 2")

Every clustering / separation test of rounds 1-4 ran on a random-init network (SI-SDR ~0.8 dB: nothing to cluster) or on
planted clusters.  The reference's published result is a TRAINED model (egs/wsj0-2mix/deep_clustering/RESULT:1, 8.858 dB on
WSJ0, which this image does not have).  This probe trains deep clustering on the synthetic corpus with the package's own
training step (``dist.train_step``: HIP forward / loss_dc / backward, clip, Adam -- onssen/utils/train.py:75-86), then on
HELD-OUT synthetic mixtures (other seeds = other voices) runs the reference's evaluation
(egs/wsj0-2mix/deep_clustering/evaluate.py:31-45: threshold at max - 40 dB, 2-means on the active bins' embeddings, binary
masks, iSTFT, SI-SDR of the best permutation) twice: with the device 2-means of this package and with
``sklearn.cluster.KMeans(2, random_state=0)`` exactly as upstream does.  Reported: loss curve, SI-SDR of the mixture / device
masks / sklearn masks / ideal binary masks, permutation-invariant mask agreement on the active bins, Lloyd passes used.

    python tools/trained_probe.py [--steps 2000 --hidden 600 --layers 2 --eval 16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def train(model, steps, dev, batch=16, frames=400, seed=0, log_every=100, voices=96, log=print):
    """``steps`` optimizer steps of deep clustering on the on-device voice-pair corpus.  Returns the loss curve
    [(step, mean loss of the last ``log_every`` steps)] and the seconds spent."""
    from onssen_amd.data import SyntheticVoicePairs
    from onssen_amd.dist import train_step
    from onssen_amd.loss import loss_dc
    from onssen_amd.utils import build_optimizer
    fo = dict(batch_size=batch, frame_length=frames, sampling_rate=8000, window_size=256, hop_size=64, db_threshold=40)
    data = SyntheticVoicePairs(fo, device=dev, voices=voices, seed=seed)
    opt = build_optimizer(model.parameters(), {"name": "adam", "lr": 1e-3})      # egs/wsj0-2mix/deep_clustering/config.json
    model.train()
    curve, acc = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(1, steps + 1):
        inp, lab = next(data)
        acc.append(train_step(model, opt, loss_dc, inp, lab))
        if it % log_every == 0 or it == steps:
            curve.append((it, float(np.mean(acc))))
            log(f"  step {it:5d}  loss {curve[-1][1]:10.2f}  ({time.perf_counter() - t0:.1f} s)")
            acc = []
    torch.cuda.synchronize()
    return curve, time.perf_counter() - t0


def _masks_from_labels(act, labels):
    """(T,F) bool, labels of the active bins -> (T,F,2) float masks like evaluate.py:38-41 (silent bins 0 in both)."""
    m = torch.zeros(act.shape + (2,), device=act.device)
    lab = torch.as_tensor(labels, device=act.device).float()
    m[..., 0][act] = lab
    m[..., 1][act] = 1.0 - lab
    return m


def evaluate(model, dev, n_utt=16, seed0=900_000, log=print):
    """Held-out mixtures, one utterance per forward like upstream (onssen/utils/test.py:29-41).  Returns a dict of means and
    the per-utterance rows."""
    from sklearn.cluster import KMeans
    from onssen_amd import separation
    from onssen_amd.evaluate import batch_SDR_torch
    from onssen_amd.features import mask_istft, stft_logmag
    from onssen_amd.hip import get_lib
    from onssen_amd.separation import dc_masks
    from onssen_amd.synthetic import synth_mixture
    lib = get_lib()
    model.eval()
    rng = np.random.default_rng(seed0)
    rows = []
    with torch.no_grad():
        for u in range(n_utt):
            n = int(rng.integers(3 * 8000, 6 * 8000)) // 64 * 64
            mix, s1, s2 = synth_mixture(seed0 + u, n, 8000, return_sources=True)
            wav = torch.from_numpy(mix[None]).to(dev)
            ref = torch.from_numpy(np.stack([s1, s2])[None]).to(dev)
            logmag, ri = stft_logmag(wav, 256, 64)
            emb, = model([logmag])
            T, F, D = emb.shape[1:]
            # --- device back end (threshold + farthest-point init + Lloyd, csrc/labels_cluster.inc)
            mk_dev = dc_masks(emb, logmag)
            torch.cuda.synchronize()
            key = next(k for k in separation._CLUSTER_WS if tuple(k[1:5]) == (1, T, F, D))
            so = int(lib.dll.onssen_dc_cluster_status_offset(1, D))
            iw = separation._CLUSTER_WS[key][so - 72 * 4:so].view(torch.int32).cpu().numpy()
            lloyd = int(iw[66] & 0xffff)
            # --- upstream's back end: sklearn on the host (evaluate.py:36-38)
            act = logmag[0] >= (logmag[0].max() - 2.0)
            lab = KMeans(n_clusters=2, random_state=0, n_init=10).fit_predict(emb[0][act].cpu().numpy())
            mk_skl = _masks_from_labels(act, lab)[None]
            # --- ideal binary mask (the training label rule, onssen/data/feature_utils.py:83-95): the ceiling
            _, r1 = stft_logmag(ref[0, :1], 256, 64)
            _, r2 = stft_logmag(ref[0, 1:], 256, 64)
            ibm = (r1[0].pow(2).sum(-1) > r2[0].pow(2).sum(-1))
            mk_ibm = _masks_from_labels(act, ibm[act].float())[None]
            sdr = {}
            for name, mk in (("device", mk_dev), ("sklearn", mk_skl), ("ideal_binary", mk_ibm)):
                est = mask_istft(ri, mk, 64, n)
                sdr[name] = float(batch_SDR_torch(est, ref)[0])
            sdr["mixture"] = float(batch_SDR_torch(torch.stack([wav, wav], 1), ref)[0])
            a = (mk_dev[0, ..., 0][act] == mk_skl[0, ..., 0][act]).float().mean().item()
            b = (mk_dev[0, ..., 0][act] == mk_ibm[0, ..., 0][act]).float().mean().item()
            rows.append({"utt": u, "seconds": n / 8000.0, "active_bins": int(act.sum()), "lloyd_passes": lloyd,
                         "agree_device_sklearn": max(a, 1 - a), "agree_device_ideal": max(b, 1 - b), **{f"sdr_{k}": v for k, v in sdr.items()}})
            r = rows[-1]
            log(f"  utt {u:2d} {r['seconds']:.1f} s: SI-SDR mixture {r['sdr_mixture']:6.2f}  device {r['sdr_device']:6.2f}  sklearn {r['sdr_sklearn']:6.2f}  "
                f"ideal {r['sdr_ideal_binary']:6.2f} dB | masks agree with sklearn {100 * r['agree_device_sklearn']:.2f} %  with IBM "
                f"{100 * r['agree_device_ideal']:.1f} % | Lloyd passes {lloyd}")
    mean = {k: float(np.mean([r[k] for r in rows])) for k in rows[0] if k not in ("utt",)}
    mean["min_agree_device_sklearn"] = float(min(r["agree_device_sklearn"] for r in rows))
    mean["max_abs_sdr_gap_device_sklearn"] = float(max(abs(r["sdr_device"] - r["sdr_sklearn"]) for r in rows))
    return mean, rows


def routes(model, dev, B=32, n=25536, log=print):
    """The same trained network through every inference route at the HEADLINE's shape (B chunks of 400 frames): whole batch
    with the fused first layer (default above 16 rows), unfused, launch-per-step recurrence, exact fp32, one chunk per
    forward, and the compacted clustering route (no embedding round trip) -- SI-SDR of each and the largest embedding
    difference against the batch-1 forwards."""
    from onssen_amd import options
    from onssen_amd.evaluate import batch_SDR_torch
    from onssen_amd.features import mask_istft, stft_logmag
    from onssen_amd.separation import dc_masks, dc_masks_from_features
    from onssen_amd.synthetic import synth_mixture
    model.eval()
    trips = [synth_mixture(910_000 + u, n, 8000, return_sources=True) for u in range(B)]
    wav = torch.from_numpy(np.stack([t[0] for t in trips])).to(dev)
    ref = torch.from_numpy(np.stack([np.stack(t[1:]) for t in trips])).to(dev)
    out = {}
    with torch.no_grad():
        logmag, ri = stft_logmag(wav, 256, 64)
        one = torch.cat([model([logmag[b:b + 1].contiguous()])[0] for b in range(B)])
        def sdr_of(masks):
            return float(batch_SDR_torch(mask_istft(ri, masks, 64, n), ref).mean())
        out["one_chunk_per_forward"] = {"si_sdr": sdr_of(torch.cat([dc_masks(one[b:b + 1].contiguous(), logmag[b:b + 1].contiguous()) for b in range(B)]))}
        for name, opts in (("batch_default", {}), ("batch_unfused", {"fuse_first_layer": "0"}), ("batch_steps", {"recurrence": "steps"}),
                           ("batch_f32", {"precision": "f32"})):
            old = options.configure(**opts)
            emb, = model([logmag])
            out[name] = {"si_sdr": sdr_of(dc_masks(emb, logmag)), "max_abs_diff_vs_one_chunk_forwards": float((emb - one).abs().max())}
            options.configure(**{k: old[k] for k in opts})
        mk = dc_masks_from_features(model, logmag)
        out["batch_compact_route"] = {"si_sdr": sdr_of(mk) if mk is not None else None}
        out["mixture"] = {"si_sdr": float(batch_SDR_torch(torch.stack([wav, wav], 1), ref).mean())}
    for k, v in out.items():
        log(f"  route {k:24s} {v}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--hidden", type=int, default=600)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--eval", type=int, default=16)
    ap.add_argument("--checkpoints", default="0,500", help="also evaluate after these step counts (comma separated)")
    ap.add_argument("--save", default="", help="write the trained state_dict here (torch.save)")
    ap.add_argument("--routes", action="store_true", help="after training: the same network through every inference route at B = 32 x 400 frames")
    args = ap.parse_args()
    from onssen_amd import nn as onn
    from onssen_amd.nn._core import _XcdStatus
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = onn.deep_clustering(129, args.hidden, args.layers, 20, dropout=0.3).to(dev)
    print(f"deep_clustering {args.layers}xBLSTM-{args.hidden}, synthetic voice-pair corpus (96 voices, 16 x 400-frame chunks per step), Adam 1e-3, clip 5")
    marks = sorted({int(c) for c in args.checkpoints.split(",") if c} | {args.steps})
    done, out = 0, {"checkpoints": []}
    for mark in marks:
        if mark > done:
            curve, secs = train(model, mark - done, dev, seed=done)
            print(f"trained steps {done}..{mark}: {secs:.1f} s = {secs / (mark - done) * 1e3:.2f} ms per step (corpus + step)")
            out.setdefault("loss_curve", []).extend([(done + s, l) for s, l in curve])
            done = mark
        print(f"--- held-out evaluation after {done} steps")
        mean, rows = evaluate(model, dev, args.eval)
        print(f"after {done:5d} steps: SI-SDR mixture {mean['sdr_mixture']:.2f}  device 2-means {mean['sdr_device']:.2f}  sklearn KMeans {mean['sdr_sklearn']:.2f}  "
              f"ideal binary mask {mean['sdr_ideal_binary']:.2f} dB | device vs sklearn masks: mean {100 * mean['agree_device_sklearn']:.3f} % "
              f"min {100 * mean['min_agree_device_sklearn']:.3f} % of the active bins, max |SI-SDR gap| {mean['max_abs_sdr_gap_device_sklearn']:.3f} dB | "
              f"Lloyd passes mean {mean['lloyd_passes']:.1f}")
        out["checkpoints"].append({"steps": done, "mean": mean, "rows": rows})
    if args.routes:
        print("--- inference routes on the trained network (32 x 400-frame chunks, held out)")
        out["routes"] = routes(model, dev)
    _XcdStatus.flush()
    if args.save:
        torch.save({"model": model.state_dict()}, args.save)
    print(json.dumps({k: v for k, v in out.items() if k != "checkpoints"} | {"summary": [{"steps": c["steps"], **c["mean"]} for c in out["checkpoints"]]}))


if __name__ == "__main__":
    main()
