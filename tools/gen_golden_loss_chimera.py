#!/usr/bin/env python
"""Golden values of the reference's loss_chimera_msa / loss_chimera_psa on random outputs / labels;
writes tests/golden/g7_loss_chimera.npz."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gen_golden import OUT, load_ref_pkg   # noqa: E402


def main():
    ref = load_ref_pkg("ref_loss", "loss")
    rng = np.random.default_rng(7)
    B, T, F, D = 2, 6, 129, 20
    emb = rng.standard_normal((B, T, F, D)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=-1, keepdims=True)
    masks = rng.random((B, T, F, 2)).astype(np.float32)
    lab = rng.integers(0, 3, (B, T, F))
    one_hot = np.stack([lab == 0, lab == 1], -1).astype(np.float64)
    mag = (np.abs(rng.standard_normal((B, T, F))) + 1e-3).astype(np.float32)
    s1 = (mag * rng.random((B, T, F))).astype(np.float32)
    s2 = (mag * rng.random((B, T, F))).astype(np.float32)
    c1 = np.cos(rng.uniform(-np.pi, np.pi, (B, T, F))).astype(np.float32)
    c2 = np.cos(rng.uniform(-np.pi, np.pi, (B, T, F))).astype(np.float32)
    tt = torch.from_numpy
    out = [tt(emb), tt(masks[..., 0]), tt(masks[..., 1])]
    msa = ref.loss_chimera_msa(out, [tt(one_hot), tt(mag), tt(s1), tt(s2)]).numpy()
    psa = ref.loss_chimera_psa(out, [tt(one_hot), tt(mag), tt(s1), tt(s2), tt(c1), tt(c2)]).numpy()
    np.savez_compressed(f"{OUT}/g7_loss_chimera.npz", emb=emb, masks=masks, one_hot=one_hot, mag=mag, s1=s1, s2=s2, c1=c1, c2=c2,
                        msa=msa, psa=psa)
    print("msa", msa.shape, msa.mean(), "psa", psa.mean())


if __name__ == "__main__":
    main()
