"""wsj0-2mix loader over real files with the reference's yield contract (SURVEY rows H1 / N4 tail).

``wsj0_2mix_dataloader(model_name, feature_options, partition, device)`` of the reference (onssen/data/wsj0_2mix.py:26-37)
globs ``<data_path>/wav8k/min/<partition>/mix/*.wav`` (``:78-79``), reads mix / s1 / s2 with librosa / torchaudio and computes
three host STFTs, a random ``frame_length`` crop and the label features per sample (``:103-158``).  Here the files are read on
the host (``read_wav``: RIFF PCM 8/16/24/32-bit and IEEE float, what ``librosa.load(fn, sr=None)`` returns for them: float32,
channels averaged) and everything after that runs on the GPU: one STFT launch per utterance for (mix, s1, s2)
(``onssen_stft_logmag_f32``), the crop, and ONE label-kernel launch per batch (``onssen_labels_f32``).  Same
``feature_options`` keys, same list layouts per ``model_name`` as the synthetic loader next door (and the reference):

    "dc"        [feature_mix] , [one_hot, mag_mix]
    "chimera"   [feature_mix] , [one_hot, mag_mix, mag_s1, mag_s2]
    "chimera++" [feature_mix] , [one_hot, mag_mix, mag_s1, mag_s2, cos_s1, cos_s2]
    "phase"     [feature_mix, phase_mix] , [one_hot, mag_mix, mag_s1, mag_s2, phase_s1, phase_s2]

Partitions "tr" / "cv": batches of ``batch_size`` in shuffled order (the reference's ``DataLoader(shuffle=True)``), an
utterance of at most ``frame_length`` frames is repeated ``frame_length // T + 1`` times before the crop (``:118-123``).
Partition "tt": whole utterances, batch 1, ``[feature_mix (1,T,F)] , [stft_r (1,T,F), stft_i (1,T,F), sig_ref (1,2,n)]`` with the
signals zero-padded by ``32 - n % 32`` samples (``get_sigs``, ``:216-228``; upstream's STFT-model branch calls a ``get_ref_sig``
that does not exist, ``:237`` -- the padded pair is what its time-domain branch builds).

Not librosa: a file whose rate differs from ``sampling_rate`` is resampled with ``scipy.signal.resample_poly`` (librosa's
``resample`` uses a different kernel: results differ in the last digits; wsj0-2mix ``wav8k`` files never take this branch).
"""
import glob
import os

import numpy as np
import torch

from ..features import stft_logmag, training_labels


def read_wav(fn):
    """(float32 mono signal, sample rate) of a RIFF file -- ``librosa.load(fn, sr=None)`` for the formats wsj0-2mix ships in."""
    from scipy.io import wavfile
    rate, data = wavfile.read(fn)
    if data.dtype == np.uint8:                       # 8-bit PCM is unsigned
        sig = (data.astype(np.float32) - 128.0) / 128.0
    elif data.dtype == np.int16:
        sig = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:                     # 24-bit comes back left-justified in 32
        sig = data.astype(np.float32) / 2147483648.0
    elif data.dtype in (np.float32, np.float64):
        sig = data.astype(np.float32)
    else:
        raise ValueError(f"{fn}: unsupported sample type {data.dtype}")
    if sig.ndim == 2:
        sig = sig.mean(axis=1, dtype=np.float32)
    return np.ascontiguousarray(sig, dtype=np.float32), int(rate)


def write_wav(fn, sig, rate, subtype="PCM_16"):
    """float signal in [-1, 1) -> RIFF file (the separated signals of an evaluation run; ``subtype`` PCM_16 or FLOAT)."""
    from scipy.io import wavfile
    sig = np.asarray(sig, dtype=np.float32)
    if subtype == "PCM_16":
        wavfile.write(fn, rate, np.clip(np.rint(sig * 32768.0), -32768, 32767).astype(np.int16))
    elif subtype == "FLOAT":
        wavfile.write(fn, rate, sig)
    else:
        raise ValueError(f"unknown subtype {subtype!r}")


def _load(fn, sampling_rate):
    sig, rate = read_wav(fn)
    if rate != sampling_rate:                        # feature_utils.get_stft:17-20 resamples instead of failing
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(rate), int(sampling_rate))
        sig = resample_poly(sig, sampling_rate // g, rate // g).astype(np.float32)
    return sig


class Wsj02mixFiles:
    """Iterable over ``(input_list, label_list)`` batches of one partition; ``len()`` = number of batches."""

    def __init__(self, model_name, feature_options, partition="tr", device="cuda:0", shuffle=None, seed=None):
        fo = feature_options
        g = (lambda k: fo[k]) if isinstance(fo, dict) else (lambda k: getattr(fo, k))
        if model_name not in ("dc", "chimera", "chimera++", "phase"):
            raise ValueError(f"unknown model_name {model_name!r}")
        self.model_name = model_name
        self.batch_size, self.frame_length = int(g("batch_size")), int(g("frame_length"))
        self.sampling_rate, self.window_size, self.hop_size = int(g("sampling_rate")), int(g("window_size")), int(g("hop_size"))
        self.db_threshold = float(g("db_threshold"))
        self.device = torch.device(device if device is not None else "cuda:0")
        self.partition = partition
        self.file_list = sorted(glob.glob(os.path.join(g("data_path"), "wav8k", "min", partition, "mix", "*.wav")))
        self.shuffle = (partition != "tt") if shuffle is None else shuffle
        self.rng = np.random.default_rng(seed)      # seed=None: fresh entropy, like the reference's unseeded np.random

    def __len__(self):
        n = len(self.file_list)
        return n if self.partition == "tt" else -(-n // self.batch_size)

    def _sources(self, fn):
        sep = os.sep + "mix" + os.sep
        return fn, fn.replace(sep, os.sep + "s1" + os.sep), fn.replace(sep, os.sep + "s2" + os.sep)

    def _iter_eval(self):
        for fn in self.file_list:
            mix, s1, s2 = (_load(f, self.sampling_rate) for f in self._sources(fn))
            gap = 32 - len(mix) % 32
            pad = lambda a: np.pad(a, (0, gap))
            # the STFT of the file AS IT IS (get_stft(fn), wsj0_2mix.py:231-233); only the reference signals are padded to a
            # multiple of 32 samples (get_sigs, :216-228) -- the estimate is then istft(..., length = padded length)
            wav = torch.from_numpy(np.ascontiguousarray(mix)[None]).to(self.device)
            logmag, ri = stft_logmag(wav, self.window_size, self.hop_size)
            sig_ref = torch.from_numpy(np.stack([pad(s1), pad(s2)])[None]).to(self.device)
            yield [logmag], [ri[..., 0].contiguous(), ri[..., 1].contiguous(), sig_ref]

    def _utterance(self, fn):
        """(3, T', F) log-magnitude and (3, T', F, 2) spectrum of (mix, s1, s2), repeated to more than frame_length frames, cropped."""
        sigs = [_load(f, self.sampling_rate) for f in self._sources(fn)]
        n = min(len(s) for s in sigs)
        wav = torch.from_numpy(np.stack([s[:n] for s in sigs])).to(self.device)
        logmag, ri = stft_logmag(wav, self.window_size, self.hop_size)
        T, L = logmag.shape[1], self.frame_length
        if T <= L:                                   # "pad in a double-copy fashion" (wsj0_2mix.py:118-123)
            times = L // T + 1
            logmag, ri = logmag.repeat(1, times, 1), ri.repeat(1, times, 1, 1)
        start = int(self.rng.integers(0, logmag.shape[1] - L))
        return logmag[:, start:start + L], ri[:, start:start + L]

    def __iter__(self):
        if self.partition == "tt":
            yield from self._iter_eval()
            return
        order = self.rng.permutation(len(self.file_list)) if self.shuffle else np.arange(len(self.file_list))
        for i0 in range(0, len(order), self.batch_size):
            items = [self._utterance(self.file_list[i]) for i in order[i0:i0 + self.batch_size]]
            logmag = torch.stack([it[0] for it in items])          # (B, 3, L, F)
            ri = torch.stack([it[1] for it in items])              # (B, 3, L, F, 2)
            feat = logmag[:, 0].contiguous()
            mix, s1, s2 = (ri[:, i].contiguous() for i in range(3))
            out = training_labels(mix, s1, s2, feat, self.db_threshold, with_cos=self.model_name == "chimera++")
            one_hot, mm, m1, m2 = out[:4]
            one_hot = one_hot.double()                             # np.zeros default dtype upstream (feature_utils.py:86)
            if self.model_name == "dc":
                yield [feat], [one_hot, mm]
            elif self.model_name == "chimera":
                yield [feat], [one_hot, mm, m1, m2]
            elif self.model_name == "chimera++":
                yield [feat], [one_hot, mm, m1, m2, out[4], out[5]]
            else:
                yield [feat, mix], [one_hot, mm, m1, m2, s1, s2]
