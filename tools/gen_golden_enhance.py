#!/usr/bin/env python
"""Golden vectors for the `enhance` model: imports the reference's own onssen.nn.enhance in the build
container (the reference never travels) and commits inputs / outputs under tests/golden/ (g5_*).  Run from the repo
root: PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_enhance.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gen_golden import OUT, load_ref_pkg, logmag_input, run_ref   # noqa: E402
from onssen_amd.synthetic import make_state_dict                        # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_nn = load_ref_pkg("ref_nn", "nn")
    F = 129
    for H, L, B, T in [(16, 2, 2, 12), (32, 1, 3, 20)]:
        seed = 40 + H + L
        sd = make_state_dict("enhance", F, H, L, seed=seed, gain=2.0)
        x = logmag_input(seed, B, T, F, 256, 64)
        mag = (10.0 ** x).astype(np.float32)
        m = ref_nn.enhance(F, H, L)
        out, = run_ref(m, sd, [x, mag])
        fn = f"{OUT}/g5_enhance_H{H}_L{L}.npz"
        np.savez_compressed(fn, kind="enhance", F=F, H=H, L=L, D=20, C=2, seed=seed, gain=2.0, x=x, mag_noisy=mag, out_clean=out)
        print("wrote", fn, out.shape, float(np.abs(out).max()))


if __name__ == "__main__":
    main()
