#!/usr/bin/env python
"""Golden vectors for batch_SDR_torch: runs the reference's own onssen/evaluate/sdr.py in the build
container and commits inputs / outputs (tests/golden/g6_batch_sdr.npz)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gen_golden import OUT, load_ref_pkg   # noqa: E402


def main():
    ref = load_ref_pkg("ref_eval", "evaluate")
    rng = np.random.default_rng(6)
    rec = {}
    for tag, B, C, n, use_mask in (("c2", 3, 2, 4000, False), ("c3m", 2, 3, 2500, True)):
        org = rng.standard_normal((B, C, n)).astype(np.float32) * rng.uniform(0.2, 1.5, (B, C, 1)).astype(np.float32)
        perm = [rng.permutation(C) for _ in range(B)]
        est = np.stack([org[b, perm[b]] for b in range(B)]) * 0.8 + 0.3 * rng.standard_normal((B, C, n)).astype(np.float32) + 0.05
        est = est.astype(np.float32)
        mask = None
        if use_mask:
            mask = np.ones((B, n), np.float32)
            for b in range(B):
                mask[b, n - 300 * (b + 1):] = 0
        sdr, idx = ref.batch_SDR_torch(torch.from_numpy(est), torch.from_numpy(org),
                                       None if mask is None else torch.from_numpy(mask), return_perm=True)
        rec.update({f"{tag}_est": est, f"{tag}_org": org, f"{tag}_sdr": sdr.numpy(), f"{tag}_perm": idx.numpy()})
        if mask is not None:
            rec[f"{tag}_mask"] = mask
        print(tag, sdr.numpy(), idx.numpy())
    np.savez_compressed(f"{OUT}/g6_batch_sdr.npz", **rec)


if __name__ == "__main__":
    main()
