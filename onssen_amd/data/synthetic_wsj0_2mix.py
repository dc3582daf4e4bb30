"""Synthetic wsj0-2mix-style loader with the reference's yield contract.

The reference's ``wsj0_2mix_dataloader(model_name, feature_options, partition, device)``
(onssen/data/wsj0_2mix.py:26-37) globs the licensed WSJ0 corpus, runs three librosa STFTs per sample on
the host and yields ``(input_list, label_list)``.  There is no corpus here, so utterances come from
``onssen_amd.synthetic`` -- and the features are computed on the GPU: one batched STFT launch for
(mix, s1, s2), a random 400-frame crop, then the label kernel (wsj0_2mix.py:103-158).  Same keys of
``feature_options``, same list layouts per ``model_name``:

    "dc"        [feature_mix] , [one_hot, mag_mix]
    "chimera"   [feature_mix] , [one_hot, mag_mix, mag_s1, mag_s2]
    "chimera++" [feature_mix] , [one_hot, mag_mix, mag_s1, mag_s2, cos_s1, cos_s2]
    "phase"     [feature_mix, phase_mix] , [one_hot, mag_mix, mag_s1, mag_s2, phase_s1, phase_s2]

``one_hot`` is float64 like upstream's (np.zeros default, feature_utils.py:86; the losses cast).  Partition "tt" follows
the EVALUATION contract (wsj0_2mix_eval_dataset, wsj0_2mix.py:166-245): whole utterances, batch 1,
``[feature_mix (1,T,F)] , [stft_r_mix (1,T,F), stft_i_mix (1,T,F), sig_ref (1,2,n)]`` with n padded to a multiple of 32.
"""
import numpy as np
import torch

from ..features import stft_logmag, training_labels
from ..synthetic import synth_mixture


class SyntheticWsj02mix:
    def __init__(self, model_name, feature_options, partition="tr", device="cuda:0", num_batches=8, seed=0):
        fo = feature_options
        g = (lambda k: fo[k]) if isinstance(fo, dict) else (lambda k: getattr(fo, k))
        self.model_name = model_name
        self.batch_size, self.frame_length = int(g("batch_size")), int(g("frame_length"))
        self.sampling_rate, self.window_size, self.hop_size = int(g("sampling_rate")), int(g("window_size")), int(g("hop_size"))
        self.db_threshold = float(g("db_threshold"))
        self.device = torch.device(device if device is not None else "cuda:0")
        self.num_batches = num_batches
        self.partition = partition
        self.seed = seed + {"tr": 0, "cv": 10_000, "tt": 20_000}.get(partition, 30_000)
        self.n_samples = self.hop_size * (self.frame_length + 40)     # a little longer than one chunk: crops differ

    def __len__(self):
        return self.num_batches

    def _iter_eval(self):
        for it in range(self.num_batches):
            n = self.hop_size * (self.frame_length + 17 * (it % 3)) + 5 * it          # utterances differ in length
            mix, s1, s2 = synth_mixture(self.seed + it, n, self.sampling_rate, return_sources=True)
            gap = 32 - n % 32                                                          # get_sigs (wsj0_2mix.py:216-228)
            pad = lambda a: np.pad(a, (0, gap))
            # the STFT of the file AS IT IS (get_stft(fn), wsj0_2mix.py:231-233); only the reference signals are padded to a
            # multiple of 32 samples (get_sigs, :216-228) -- the estimate is then istft(..., length = padded length)
            wav = torch.from_numpy(np.ascontiguousarray(mix)[None]).to(self.device)
            logmag, ri = stft_logmag(wav, self.window_size, self.hop_size)
            sig_ref = torch.from_numpy(np.stack([pad(s1), pad(s2)])[None]).to(self.device)
            yield [logmag], [ri[..., 0].contiguous(), ri[..., 1].contiguous(), sig_ref]

    def __iter__(self):
        if self.partition == "tt":
            yield from self._iter_eval()
            return
        rng = np.random.default_rng(self.seed)
        B, L = self.batch_size, self.frame_length
        for it in range(self.num_batches):
            trip = [synth_mixture(self.seed + it * B + b, self.n_samples, self.sampling_rate, return_sources=True)
                    for b in range(B)]
            wav = torch.from_numpy(np.stack([np.stack(t) for t in trip])).to(self.device)     # (B, 3, n)
            logmag, ri = stft_logmag(wav.view(3 * B, -1), self.window_size, self.hop_size)
            T, F = logmag.shape[1], logmag.shape[2]
            logmag, ri = logmag.view(B, 3, T, F), ri.view(B, 3, T, F, 2)
            start = int(rng.integers(0, T - L))            # np.random.randint crop (wsj0_2mix.py:125-128)
            feat = logmag[:, 0, start:start + L].contiguous()
            mix, s1, s2 = (ri[:, i, start:start + L].contiguous() for i in range(3))
            with_cos = self.model_name == "chimera++"
            out = training_labels(mix, s1, s2, feat, self.db_threshold, with_cos=with_cos)
            one_hot, mm, m1, m2 = out[:4]
            one_hot = one_hot.double()
            if self.model_name == "dc":
                yield [feat], [one_hot, mm]
            elif self.model_name == "chimera":
                yield [feat], [one_hot, mm, m1, m2]
            elif self.model_name == "chimera++":
                yield [feat], [one_hot, mm, m1, m2, out[4], out[5]]
            elif self.model_name == "phase":
                yield [feat, mix], [one_hot, mm, m1, m2, s1, s2]
            else:
                raise ValueError(f"unknown model_name {self.model_name!r}")


class SyntheticVoicePairs:
    """An endless deep-clustering TRAINING corpus mixed on the device (round 5): ``voices`` synthetic voices are generated once
    on the host (``synthetic.synth_voice``) and kept in HBM; every batch pairs two different voices per chunk at a relative
    level drawn from U(-2.5, +2.5) dB (the wsj0-2mix convention), adds the -40 dB noise floor, peak-normalises,
    and runs the same HIP front end as ``SyntheticWsj02mix`` (one batched STFT of (mix, s1, s2), a random crop, the label
    kernel).  Yield contract of the reference's "dc" loader (onssen/data/wsj0_2mix.py:103-158):
    ``[feature_mix (B,L,F)] , [one_hot (B,L,F,2) float64, mag_mix (B,L,F)]``.  ``voices * (voices - 1)`` pairings x levels x crops:
    a network cannot memorise it in a few thousand steps, and the held-out mixtures of ``synthetic.synth_mixture`` (other
    seeds: other voices) measure generalisation.  No host work per batch: it keeps up with a 5 ms training step."""

    def __init__(self, feature_options, device="cuda:0", voices=96, seed=0):
        from ..synthetic import synth_voice
        fo = feature_options
        g = (lambda k: fo[k]) if isinstance(fo, dict) else (lambda k: getattr(fo, k))
        self.batch_size, self.frame_length = int(g("batch_size")), int(g("frame_length"))
        self.sampling_rate, self.window_size, self.hop_size = int(g("sampling_rate")), int(g("window_size")), int(g("hop_size"))
        self.db_threshold = float(g("db_threshold"))
        self.device = torch.device(device)
        self.n_samples = self.hop_size * (self.frame_length + 40)
        host = np.stack([synth_voice(500_000 + seed * 10_000 + v, self.n_samples, self.sampling_rate) for v in range(voices)])
        self.voices = torch.from_numpy(host).to(self.device)
        self.gen = torch.Generator(device=self.device).manual_seed(seed)

    def __iter__(self):
        return self

    def __next__(self):
        B, L, V, dev, g = self.batch_size, self.frame_length, self.voices.shape[0], self.device, self.gen
        i = torch.randint(0, V, (B,), device=dev, generator=g)
        j = (i + torch.randint(1, V, (B,), device=dev, generator=g)) % V            # a different voice
        level = 10.0 ** ((torch.rand(B, device=dev, generator=g) * 5.0 - 2.5) / 20.0)
        s1, s2 = self.voices[i], self.voices[j] * level[:, None]
        clean = s1 + s2
        noise = torch.randn(B, self.n_samples, device=dev, generator=g) * (1e-2 * clean.std(dim=1, keepdim=True))
        mix = clean + noise
        scale = 0.9 / mix.abs().amax(dim=1, keepdim=True)
        wav = torch.stack([mix * scale, s1 * scale, s2 * scale], 1).contiguous()    # (B, 3, n)
        logmag, ri = stft_logmag(wav.view(3 * B, -1), self.window_size, self.hop_size)
        T, F = logmag.shape[1], logmag.shape[2]
        logmag, ri = logmag.view(B, 3, T, F), ri.view(B, 3, T, F, 2)
        start = int(torch.randint(0, T - L, (1,), device="cpu"))
        feat = logmag[:, 0, start:start + L].contiguous()
        mx, a, b = (ri[:, k, start:start + L].contiguous() for k in range(3))
        one_hot, mm = training_labels(mx, a, b, feat, self.db_threshold)[:2]
        return [feat], [one_hot.double(), mm]


def wsj0_2mix_dataloader(model_name, feature_options, partition, device=None):
    """Same call signature as onssen.data.wsj0_2mix_dataloader (wsj0_2mix.py:26); always synthetic -- the package-level factory
    (onssen_amd.data.wsj0_2mix_dataloader) reads real files when feature_options.data_path has them."""
    return SyntheticWsj02mix(model_name, feature_options, partition, device)
