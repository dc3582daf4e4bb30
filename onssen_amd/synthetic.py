"""Deterministic synthetic weights and wsj0-2mix-style mixtures.

There is no WSJ0 corpus and no trained checkpoint in this environment, so the
tests, the golden-vector generator and bench.py all draw weights and audio
from the counter-based generators below (SURVEY 8c "G2", 8d "Synthetic
inputs").  Pure NumPy: usable on hosts without a GPU.
"""
import numpy as np


def lstm_param_shapes(in_dim, H, L, prefix="rnn."):
    """state_dict entries of nn.LSTM(in_dim, H, L, bidirectional=True) in
    PyTorch's registration order (SURVEY 8b state_dict layout)."""
    out = []
    for k in range(L):
        ik = in_dim if k == 0 else 2 * H
        for sfx in ("", "_reverse"):
            out += [(f"{prefix}weight_ih_l{k}{sfx}", (4 * H, ik)),
                    (f"{prefix}weight_hh_l{k}{sfx}", (4 * H, H)),
                    (f"{prefix}bias_ih_l{k}{sfx}", (4 * H,)),
                    (f"{prefix}bias_hh_l{k}{sfx}", (4 * H,))]
    return out


def make_state_dict(kind, input_dim, hidden_dim, num_layers, embedding_dim=20,
                    num_speaker=2, seed=0, gain=1.0):
    """Reference-layout state_dict (numpy float32) for ``kind`` in
    {"deep_clustering", "chimera", "phase_net", "enhance"}.

    LSTM / Linear tensors ~ U(-1/sqrt(fan), 1/sqrt(fan)) like PyTorch's default
    init (times ``gain``); BatchNorm affine and running statistics are drawn
    non-trivially so that the eval-mode BN fold is exercised.
    """
    rng = np.random.default_rng(seed)
    F, H, L, D, C = input_dim, hidden_dim, num_layers, embedding_dim, num_speaker
    sd = {}

    def uni(shape, bound):
        return (rng.uniform(-bound, bound, size=shape) * gain).astype(np.float32)

    def lstm(prefix, in_dim):
        for name, shape in lstm_param_shapes(in_dim, H, L, prefix):
            sd[name] = uni(shape, 1.0 / np.sqrt(H))

    def linear(prefix, out_f, in_f):
        sd[prefix + "weight"] = uni((out_f, in_f), 1.0 / np.sqrt(in_f))
        sd[prefix + "bias"] = uni((out_f,), 1.0 / np.sqrt(in_f))

    def bn(prefix, n):
        sd[prefix + "weight"] = rng.uniform(0.5, 1.5, n).astype(np.float32)
        sd[prefix + "bias"] = rng.uniform(-0.2, 0.2, n).astype(np.float32)
        sd[prefix + "running_mean"] = rng.uniform(-0.1, 0.1, n).astype(np.float32)
        sd[prefix + "running_var"] = rng.uniform(0.05, 0.5, n).astype(np.float32)
        sd[prefix + "num_batches_tracked"] = np.asarray(7, dtype=np.int64)

    if kind == "deep_clustering":
        lstm("rnn.", F)
        bn("bn.", 2 * H)
        linear("fc_dc.", F * D, 2 * H)
    elif kind == "chimera":
        lstm("rnn.", F)
        linear("fc_dc.", F * D, 2 * H)
        linear("fc_mi.", F * C, 2 * H)
    elif kind == "phase_net":
        lstm("rnn.", 3 * F)
        bn("bn.", 2 * H)
        linear("fc_phase.", C * F, 2 * H)
        lstm("chimera.rnn.", F)
        linear("chimera.fc_dc.", F * D, 2 * H)
        linear("chimera.fc_mi.", F * C, 2 * H)
    elif kind == "enhance":
        lstm("rnn.", F)
        bn("bn.", 2 * H)
        linear("fc_mi.", F, 2 * H)
        linear("fc_pre.", F, F)
        linear("fc_post.", F, F)
    else:
        raise ValueError(kind)
    return sd


def _speaker(rng, n, sr):
    """One synthetic 'voice': 20 harmonics of a random-walk f0 (80-255 Hz),
    1/k roll-off, 3-5 Hz syllabic envelope with random pauses."""
    t = np.arange(n) / sr
    f0 = 80.0 + 175.0 * rng.random()
    walk = np.cumsum(rng.normal(0.0, 0.6, n // 160 + 2))
    f0_t = np.clip(f0 + np.interp(np.arange(n), np.arange(len(walk)) * 160, walk), 80.0, 255.0)
    ph = 2 * np.pi * np.cumsum(f0_t) / sr
    sig = np.zeros(n)
    for k in range(1, 21):
        if k * 255.0 < sr / 2:
            sig += np.sin(k * ph + rng.uniform(0, 2 * np.pi)) / k
    env = 0.5 * (1 + np.sin(2 * np.pi * rng.uniform(3, 5) * t + rng.uniform(0, 2 * np.pi)))
    gate = np.repeat((rng.random(n // 800 + 1) > 0.2).astype(float), 800)[:n]
    gate = np.convolve(gate, np.ones(160) / 160, mode="same")[:n]   # 'same' returns max(len) samples
    return sig * env * gate


def synth_voice(seed, n_samples, sr=8000):
    """One synthetic voice by itself (unit standard deviation, float32): the building block of ``synth_mixture``, for
    corpora that pair voices on the fly (``onssen_amd.data.synthetic_wsj0_2mix.SyntheticVoicePairs``)."""
    v = _speaker(np.random.default_rng(seed), n_samples, sr)
    return (v / (np.std(v) + 1e-9)).astype(np.float32)


def synth_mixture(seed, n_samples=25536, sr=8000, return_sources=False):
    """2-speaker mixture, float32 in [-1, 1], peak 0.9.  25 536
    samples at hop 64 give exactly 400 STFT frames."""
    rng = np.random.default_rng(seed)
    s1 = _speaker(rng, n_samples, sr)
    s2 = _speaker(rng, n_samples, sr)
    s2 *= 10 ** (rng.uniform(-2.5, 2.5) / 20) * (np.std(s1) + 1e-9) / (np.std(s2) + 1e-9)
    noise = rng.normal(0, 1, n_samples) * max(1e-2 * np.std(s1 + s2), 1e-4)   # never an all-zero mixture
    mix = s1 + s2 + noise
    scale = 0.9 / np.max(np.abs(mix))
    if return_sources:
        return ((mix * scale).astype(np.float32), (s1 * scale).astype(np.float32),
                (s2 * scale).astype(np.float32))
    return (mix * scale).astype(np.float32)


def synth_batch(config_id, batch, n_samples=25536, sr=8000):
    """(B, n_samples) float32; seeds 1000*config + utterance index."""
    return np.stack([synth_mixture(1000 * config_id + u, n_samples, sr) for u in range(batch)])
