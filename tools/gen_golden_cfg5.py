#!/usr/bin/env python
"""Golden fixture for BASELINE config 5 at its FULL shape: the reference's phase_net (chimera++ 4 x
BLSTM-600 + the phase BLSTM, onssen/nn/phase_network.py:8-67), 16 kHz, STFT 512/128, 1-second chunks -> T = 126 frames,
F = 257, batch 2.  Runs only in the build container (imports /root/reference like tools/gen_golden.py); commits the
input RECIPE (seeds) and a strided subsample + per-frame checksums of every output.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_cfg5.py
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_golden import OUT, load_ref_pkg, run_ref  # noqa: E402
from onssen_amd.synthetic import make_state_dict, synth_mixture  # noqa: E402
from oracle import np_oracle  # noqa: E402


def cfg5_inputs(seed, B, n_samples=16000, n_fft=512, hop=128):
    """(x_mag (B,T,F) log-magnitude, x_phase (B,T,F,2) = (Re, Im)) of seeded synthetic 16 kHz mixtures -- the recipe the GPU
    test repeats (feature_utils.get_log_magnitude / get_phase on get_stft's output, onssen/data/feature_utils.py:5-64)."""
    mags, phs = [], []
    for b in range(B):
        X = np_oracle.stft(synth_mixture(seed * 100 + b, n_samples=n_samples, sr=16000), n_fft, hop)
        mags.append(np_oracle.log_magnitude(X))
        phs.append(np_oracle.phase_re_im(X))
    return np.stack(mags).astype(np.float32), np.stack(phs).astype(np.float32)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_nn = load_ref_pkg("ref_nn", "nn")
    F, H, L, B, seed, x_seed = 257, 600, 4, 2, 0, 9
    sys.modules["ref_nn.phase_network"].output_dim = F            # the undefined free variable
    sd = make_state_dict("phase_net", F, H, L, 20, 2, seed=seed)
    x_mag, x_phase = cfg5_inputs(x_seed, B)
    assert x_mag.shape == (B, 126, F) and x_phase.shape == (B, 126, F, 2), (x_mag.shape, x_phase.shape)
    m = ref_nn.phase_net(F, H, L, 20)
    emb, mask_a, mask_b, ph_a, ph_b = run_ref(m, sd, [x_mag, x_phase])
    rec = {"kind": "phase_net", "F": F, "H": H, "L": L, "D": 20, "C": 2, "seed": seed, "gain": 1.0, "x_seed": x_seed, "B": B,
           "T": 126, "n_fft": 512, "hop": 128, "n_samples": 16000,
           "emb_sub": emb[:, ::9, ::16, :].copy(), "emb_sum_per_frame": emb.astype(np.float64).sum(axis=(2, 3)).astype(np.float32),
           "mask_A_sub": mask_a[:, ::5, :].copy(), "mask_B_sub": mask_b[:, ::5, :].copy(),
           "phase_A_sub": ph_a[:, ::5, ::4, :].copy(), "phase_B_sub": ph_b[:, ::5, ::4, :].copy(),
           "phase_A_sum_per_frame": ph_a.astype(np.float64).sum(axis=(2, 3)).astype(np.float32),
           "phase_B_sum_per_frame": ph_b.astype(np.float64).sum(axis=(2, 3)).astype(np.float32)}
    fn = f"{OUT}/g2_cfg5_phase_L4.npz"
    np.savez_compressed(fn, **rec)
    print("wrote", fn, {k: v.shape for k, v in rec.items() if hasattr(v, "shape") and v.ndim})


if __name__ == "__main__":
    main()
