"""Parity at the reference's EVALUATION length (round 5): onssen/data/wsj0_2mix.py:231-245 yields whole utterances and
onssen/utils/test.py:29-41 runs them one per forward -- 500 to 1000 frames at 8 kHz / hop 64, not the 400-frame training
chunks.  The recurrence is a serial chain: whatever error the split-bf16 products make is fed back a thousand times.  These
tests pin it: deep_clustering (H = 600, L = 2 and 3) at B = 1, T = 1000, and a ragged batch of 16 utterances of 375-1000
frames, against ``oracle.np_oracle`` (the reference's fp32 arithmetic; fixtures pin it to the reference's own outputs) in the
default split-bf16 mode and in exact-fp32 mode.  The error at T = 100 / 400 / 1000 (also against the oracle run in fp64, which
separates this path's error from the fp32 oracle's own round-off) is printed and, on the GPU box, written to
gpurun_out/eval_length_errors.json (DESIGN.md carries the table)."""
import json
import os

import numpy as np
import pytest
import torch

from onssen_amd.synthetic import make_state_dict, synth_mixture
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu

# the contract of each mode (tests/test_gpu_parity.py `prec`): |a - b| <= atol + 1e-4 |b| elementwise, rel-L2 per embedding <= 1e-4
ATOL = {"bf16x3": 5e-5, "f32": 1e-5}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from onssen_amd.hip import get_lib
    get_lib()
    return torch.device("cuda:0")


def build(dev, L, H=600, F=129, seed=11):
    from onssen_amd import nn as onn
    sd = make_state_dict("deep_clustering", F, H, L, 20, 2, seed=seed, gain=1.0)
    m = onn.deep_clustering(F, H, L, 20)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.to(dev).eval(), sd


def utterance_features(seed, T):
    """Log-magnitude of a synthetic 2-speaker mixture of exactly T frames (the oracle's front end: the network is what is under test)."""
    return O.log_magnitude(O.stft(synth_mixture(seed, (T - 1) * 64), 256, 64)).astype(np.float32)


def errors(got, ref):
    got64, ref64 = got.astype(np.float64), ref.astype(np.float64)
    d = np.abs(got64 - ref64)
    rl2 = np.linalg.norm(got64 - ref64, axis=-1) / np.maximum(np.linalg.norm(ref64, axis=-1), 1e-30)
    return {"max_abs": float(d.max()), "max_excess_over_1e-4_rel": float((d - 1e-4 * np.abs(ref64)).max()), "max_rel_l2": float(rl2.max()),
            "mean_rel_l2": float(rl2.mean())}


def record(name, rows):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        path = os.path.join(out, "eval_length_errors.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[name] = rows
        json.dump(data, open(path, "w"), indent=1)


@pytest.mark.parametrize("L", [2, 3])
@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
def test_dc_whole_utterance_1000_frames_matches_the_oracle(dev, monkeypatch, mode, L):
    monkeypatch.setenv("ONSSEN_PRECISION", mode)
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    m, sd = build(dev, L)
    full = utterance_features(4200 + L, 1000)
    rows = {}
    for T in (100, 400, 1000):
        x = full[None, :T]
        with torch.no_grad():
            emb, = m([torch.from_numpy(x).to(dev)])
        got = emb.cpu().numpy()
        ref32 = O.deep_clustering_forward(sd, x)                       # the reference's arithmetic (fp32)
        ref64 = O.deep_clustering_forward(sd, x, dtype=np.float64)     # the same network without round-off
        rows[T] = {"vs_fp32_oracle": errors(got, ref32), "vs_fp64_oracle": errors(got, ref64),
                   "fp32_oracle_vs_fp64": errors(ref32, ref64)}
        e = rows[T]["vs_fp32_oracle"]
        print(f"dc L={L} {mode} B=1 T={T}: vs fp32 oracle max|d| {e['max_abs']:.2e} rel-L2 max {e['max_rel_l2']:.2e} | vs fp64 "
              f"max|d| {rows[T]['vs_fp64_oracle']['max_abs']:.2e} | fp32 oracle's own round-off {rows[T]['fp32_oracle_vs_fp64']['max_abs']:.2e}")
        assert np.isfinite(got).all()
        np.testing.assert_allclose(np.linalg.norm(got, axis=-1), 1.0, atol=1e-5)
    record(f"dc_l{L}_{mode}_b1", rows)
    for T, r in rows.items():
        e = r["vs_fp32_oracle"]
        assert e["max_excess_over_1e-4_rel"] <= ATOL[mode], (T, e)
        assert e["max_rel_l2"] <= 1e-4, (T, e)
    # the serial chain does not amplify the error: 10x the steps may not cost 4x the error
    assert rows[1000]["vs_fp64_oracle"]["max_abs"] <= 4.0 * max(rows[100]["vs_fp64_oracle"]["max_abs"], 2e-6)


@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
def test_dc_ragged_batch_of_16_evaluation_utterances_matches_the_oracle(dev, monkeypatch, mode):
    """K = 16 whole utterances of 375-1000 frames in ONE forward (``frames=``): every row against the oracle run on that
    utterance alone, like upstream's batch-1 loop (rows 0, 5, 10, 15: the longest, the shortest and two in between get the
    full oracle; all rows are compared with their own batch-1 forward bit for bit in the default mode)."""
    monkeypatch.setenv("ONSSEN_PRECISION", mode)
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    m, sd = build(dev, 2)
    rng = np.random.default_rng(16)
    frames = [int(t) for t in rng.integers(375, 1001, 16)]
    frames[0], frames[5] = 1000, 375
    T = max(frames)
    x = np.full((16, T, 129), np.nan, np.float32)                     # NaN padding: nothing of it may spread
    for b, Tb in enumerate(frames):
        x[b, :Tb] = utterance_features(7000 + b, Tb)
    xd = torch.from_numpy(x).to(dev)
    with torch.no_grad():
        emb, = m([xd], frames=frames)
        assert torch.isfinite(emb).all()
        if mode == "bf16x3":
            for b in range(16):
                one, = m([xd[b:b + 1, :frames[b]].contiguous()])
                assert torch.equal(emb[b, :frames[b]], one[0]), f"row {b} differs from its batch-1 forward"
    rows = {}
    for b in (0, 5, 10, 15):
        ref = O.deep_clustering_forward(sd, x[b:b + 1, :frames[b]])[0]
        e = rows[f"row{b}_T{frames[b]}"] = errors(emb[b, :frames[b]].cpu().numpy(), ref)
        print(f"ragged K=16 {mode} row {b} ({frames[b]} frames): max|d| {e['max_abs']:.2e} rel-L2 max {e['max_rel_l2']:.2e}")
        assert e["max_excess_over_1e-4_rel"] <= ATOL[mode], (b, e)
        assert e["max_rel_l2"] <= 1e-4, (b, e)
    record(f"dc_l2_{mode}_ragged16", rows)
