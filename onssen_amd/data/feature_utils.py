"""``onssen.data.feature_utils`` under the reference's own names (data front-end surface).

A reference-side caller -- ``wsj0_2mix_dataset.get_feature`` (onssen/data/wsj0_2mix.py:114-152) -- imports ``get_stft``,
``get_log_magnitude``, ``get_phase``, ``get_cos_difference`` and ``get_one_hot`` and passes NumPy arrays between them.  The same
names, argument meaning and return types here (NumPy in, NumPy out), with the arithmetic on the device:

    get_stft(fn, sampling_rate, window_size, hop_size) -> (T, F) complex64       feature_utils.py:5-21
        the file is read by ``read_wav`` (what ``librosa.load(fn, sr=None)`` returns for RIFF files), resampled when its rate
        differs (``:17-20``), and goes through ONE ``stft_logmag`` launch (``onssen_stft_logmag_f32``: fp64 FFT in the LDS)
    get_log_magnitude(stft, epsilon=1e-7) -> (T, F) float32                      feature_utils.py:49-51   onssen_log_magnitude_f32
    get_phase(stft) -> (T, F, 2) float32 (Re, Im)                                feature_utils.py:54-64   (a copy: no arithmetic)
    get_angle(stft) / get_cos_difference(stft_1, stft_2) -> (T, F) float32       feature_utils.py:67-80   onssen_cos_difference_f32
    get_one_hot(feature_mix, mag_s1, mag_s2, db_threshold) -> (T, F, 2) float64  feature_utils.py:83-95   onssen_one_hot_f32

The batched loaders (``wsj0_2mix.Wsj02mixFiles``, ``synthetic_wsj0_2mix``) do NOT go through these: they keep everything on
the device and use the fused kernels (``features.stft_logmag`` / ``training_labels``: one launch per batch).  These entry points
are for code that was written against the reference's helpers; each call is a host -> device -> host round trip.
There is no CPU fallback: without a ROCm device (or without libonssen_hip.so) every function here raises.
"""
import numpy as np
import torch

from ..features import stft_logmag
from ..hip import get_lib


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("onssen_amd.data.feature_utils: needs a ROCm device; onssen_amd has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ri(stft, what):
    """complex ndarray (any shape) -> (device float32 tensor of interleaved (re, im) pairs, shape of the complex array)."""
    a = np.asarray(stft)
    if not np.iscomplexobj(a):
        raise TypeError(f"{what}: expected a complex STFT array, got dtype {a.dtype}")
    a = np.ascontiguousarray(a, dtype=np.complex64)
    if a.size == 0:
        raise ValueError(f"{what}: empty array")
    return torch.from_numpy(a.view(np.float32).reshape(-1)).to(_device()), a.shape


def get_stft(fn, sampling_rate, window_size, hop_size):
    """fn: path of the wav file; returns the frame x frequency complex64 STFT (librosa < 0.10 defaults: periodic Hann window of
    ``window_size``, centred frames, reflect padding, 1 + n // hop_size frames)."""
    from .wsj0_2mix import _load
    sig = _load(fn, int(sampling_rate))
    if sig.shape[0] <= window_size // 2:
        raise ValueError(f"{fn}: {sig.shape[0]} samples are too few for reflect padding of {window_size // 2}")
    _, ri = stft_logmag(torch.from_numpy(sig).to(_device()), int(window_size), int(hop_size))
    return np.ascontiguousarray(ri[0].cpu().numpy()).view(np.complex64)[..., 0]


def get_log_magnitude(stft, epsilon=1e-7):
    ri, shape = _ri(stft, "get_log_magnitude")
    out = torch.empty(ri.numel() // 2, device=ri.device, dtype=torch.float32)
    get_lib().log_magnitude(ri.data_ptr(), out.numel(), float(epsilon), out.data_ptr(), _stream())
    return out.cpu().numpy().reshape(shape)


def get_phase(stft):
    """frame x frequency complex array -> frame x frequency x 2 real array (real part, imaginary part)."""
    a = np.asarray(stft)
    return np.stack([np.real(a), np.imag(a)], axis=-1)


def get_cos_difference(stft_1, stft_2):
    a, shape = _ri(stft_1, "get_cos_difference")
    b, shape_b = _ri(stft_2, "get_cos_difference")
    if shape != shape_b:
        raise ValueError(f"get_cos_difference: shapes differ: {shape} vs {shape_b}")
    out = torch.empty(a.numel() // 2, device=a.device, dtype=torch.float32)
    get_lib().cos_difference(a.data_ptr(), b.data_ptr(), out.numel(), out.data_ptr(), _stream())
    return out.cpu().numpy().reshape(shape)


def get_angle(stft):
    """np.angle of the STFT, as cos-difference's building block upstream (feature_utils.py:67-74); host NumPy: the reference's
    callers only use it through get_cos_difference."""
    return np.angle(np.asarray(stft))


def get_one_hot(feature_mix, mag_s1, mag_s2, db_threshold):
    """(T, F) log-magnitude of the mixture and the two sources' magnitudes -> (T, F, 2) float64 labels: the louder source per
    bin (speaker 0 on ties: np.argmax), all-zero where feature_mix < max(feature_mix) - db_threshold / 20."""
    dev = _device()
    f, m1, m2 = (np.ascontiguousarray(np.asarray(x), dtype=np.float32) for x in (feature_mix, mag_s1, mag_s2))
    if not (f.shape == m1.shape == m2.shape) or f.size == 0:
        raise ValueError(f"get_one_hot: shapes {f.shape}, {m1.shape}, {m2.shape}")
    tf, t1, t2 = (torch.from_numpy(x.reshape(-1)).to(dev) for x in (f, m1, m2))
    out = torch.empty(f.size, 2, device=dev, dtype=torch.float32)
    umax = torch.empty(1, device=dev, dtype=torch.float32)
    get_lib().one_hot(tf.data_ptr(), t1.data_ptr(), t2.data_ptr(), 1, f.size, float(db_threshold), umax.data_ptr(),
                      out.data_ptr(), _stream())
    return out.cpu().numpy().astype(np.float64).reshape(f.shape + (2,))


__all__ = ["get_stft", "get_log_magnitude", "get_phase", "get_angle", "get_cos_difference", "get_one_hot"]
