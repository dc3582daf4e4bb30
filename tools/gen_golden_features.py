#!/usr/bin/env python
"""Golden fixture G3 for the label / feature helpers: get_log_magnitude, get_phase,
get_cos_difference, get_one_hot of the reference (onssen/data/feature_utils.py:49-95) are pure NumPy; the module only
fails to import because of its top-level ``import librosa`` -- a stub module in sys.modules gets past that (SURVEY 8c
route 2).  Runs only in the build container; commits the synthetic STFT inputs (this repo's own restatement of librosa's
stft on seeded synthetic speakers: the reference's get_stft cannot run here) and the reference's outputs.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_features.py
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from onssen_amd.synthetic import synth_mixture  # noqa: E402
from oracle import np_oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def load_feature_utils():
    for name in ("librosa", "librosa.core", "librosa.feature"):
        sys.modules.setdefault(name, types.ModuleType(name))
    spec = importlib.util.spec_from_file_location("ref_feature_utils", "/root/reference/onssen/data/feature_utils.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    fu = load_feature_utils()
    rec = {}
    for tag, n_fft, hop, n, sr in (("a", 256, 64, 64 * 39, 8000), ("b", 512, 128, 128 * 20, 16000)):
        mix, s1, s2 = synth_mixture(17 if tag == "a" else 18, n, sr, return_sources=True)
        S1, S2, X = (np_oracle.stft(s, n_fft, hop) for s in (s1, s2, mix))
        feat = fu.get_log_magnitude(X)
        rec.update({f"{tag}_X": X, f"{tag}_S1": S1, f"{tag}_S2": S2,
                    f"{tag}_log_magnitude": feat, f"{tag}_log_magnitude_eps3": fu.get_log_magnitude(X, 1e-3),
                    f"{tag}_phase": fu.get_phase(X),
                    f"{tag}_cos_s1": fu.get_cos_difference(X, S1), f"{tag}_cos_s2": fu.get_cos_difference(X, S2),
                    f"{tag}_one_hot_40": fu.get_one_hot(feat, np.abs(S1), np.abs(S2), 40),
                    f"{tag}_one_hot_20": fu.get_one_hot(feat, np.abs(S1), np.abs(S2), 20)})
    fn = f"{OUT}/g3_features.npz"
    np.savez_compressed(fn, **rec)
    print("wrote", fn, {k: (v.shape, str(v.dtype)) for k, v in rec.items()})


if __name__ == "__main__":
    main()
