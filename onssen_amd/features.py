"""GPU front end / back end of the separation path (K1, K2, K10).

Counterparts of onssen/data/feature_utils.py (get_stft, get_log_magnitude,
get_phase) and of the mask-apply + librosa.istft tail of
egs/wsj0-2mix/*/evaluate.py, operating on device tensors."""
import torch

from .hip import get_lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _lengths_i32(lengths, B, hi, device, what, lo=1):
    """Per-row extents of a ragged batch as an int32 device tensor.  Values that come from the host are validated there; a
    device tensor is taken as it is (checking it would cost a synchronisation per call): its values must lie in [lo, hi]."""
    if torch.is_tensor(lengths) and lengths.is_cuda:
        out = lengths.to(device=device, dtype=torch.int32).contiguous()
    else:
        host = torch.as_tensor(lengths, dtype=torch.int64).reshape(-1)
        if host.numel() != B or int(host.min()) < lo or int(host.max()) > hi:
            raise ValueError(f"{what}: expected {B} values in [{lo}, {hi}], got {host.tolist()}")
        out = host.to(device=device, dtype=torch.int32)
    if out.numel() != B:
        raise ValueError(f"{what}: expected {B} values, got {out.numel()}")
    return out


def stft_logmag(wav, window_size=256, hop_size=64, epsilon=1e-7, return_stft=True, lengths=None):
    """wav (B, n) or (n,) float32 cuda tensor ->
    (log_magnitude (B,T,F) float32, stft_ri (B,T,F,2) float32 | None).

    stft_ri[..., 0] + 1j*stft_ri[..., 1] is what feature_utils.get_stft
    returns per utterance (frame x frequency complex64; librosa<0.10 defaults:
    periodic Hann, centred, reflect padding); stft_ri itself is get_phase.

    ``lengths`` (B,): a RAGGED batch -- row b holds lengths[b] <= n valid samples (the rest is padding and is never
    read); its 1 + lengths[b] // hop frames are bit for bit those of a batch-1 call on wav[b, :lengths[b]], the frames
    after them are silence."""
    if wav.dim() == 1:
        wav = wav[None]
    if not wav.is_cuda:
        raise RuntimeError("stft_logmag: needs a tensor on a ROCm device; onssen_amd has no CPU fallback")
    wav = wav.float()
    if wav.stride(1) != 1:
        wav = wav.contiguous()
    B, n = wav.shape
    T, F = 1 + n // hop_size, window_size // 2 + 1
    logmag = torch.empty(B, T, F, device=wav.device, dtype=torch.float32)
    ri = torch.empty(B, T, F, 2, device=wav.device, dtype=torch.float32) if return_stft else None
    if lengths is not None:      # (every utterance must be longer than window_size / 2 samples: reflect padding)
        lengths = _lengths_i32(lengths, B, n, wav.device, "lengths", lo=window_size // 2 + 1)
    get_lib().stft_logmag(wav.data_ptr(), B, n, wav.stride(0), window_size, hop_size, float(epsilon),
                          logmag.data_ptr(), ri.data_ptr() if ri is not None else None, _stream(),
                          n_per_utt=lengths.data_ptr() if lengths is not None else None)
    return logmag, ri


def mask_istft(stft_ri, masks, hop_size=64, length=None, frames=None, lengths=None):
    """stft_ri (B,T,F,2); masks (B,T,F,C) (any strides) or None ->
    (B, C, length) float32: istft(stft * mask_c) per speaker, librosa
    semantics (hop, length=nsample).

    ``frames`` / ``lengths`` (B,): a RAGGED batch -- row b has frames[b] <= T frames and lengths[b] <= length output
    samples (zeros after them); inside them the result is bit for bit that of a batch-1 call on that utterance."""
    if not stft_ri.is_cuda:
        raise RuntimeError("mask_istft: needs tensors on a ROCm device; onssen_amd has no CPU fallback")
    stft_ri = stft_ri.float().contiguous()
    B, T, F, _ = stft_ri.shape
    n_fft = 2 * (F - 1)
    if length is None:
        length = hop_size * (T - 1)
    lib = get_lib()
    rag = {}
    if (frames is None) != (lengths is None):
        raise ValueError("mask_istft: a ragged batch needs both frames= and lengths=")
    if frames is not None:
        frames = _lengths_i32(frames, B, T, stft_ri.device, "frames")
        lengths = _lengths_i32(lengths, B, length, stft_ri.device, "lengths")
        rag = dict(frames=frames.data_ptr(), lengths=lengths.data_ptr())
    if masks is None:
        out = torch.empty(B, 1, length, device=stft_ri.device, dtype=torch.float32)
        lib.mask_istft(stft_ri.data_ptr(), None, 0, 0, 0, 0, B, 1, T, n_fft, hop_size, length, out.data_ptr(),
                       _stream(), **rag)
        return out
    masks = masks.float()
    C = masks.shape[3]
    out = torch.empty(B, C, length, device=stft_ri.device, dtype=torch.float32)
    lib.mask_istft(stft_ri.data_ptr(), masks.data_ptr(), masks.stride(0), masks.stride(3), masks.stride(1),
                   masks.stride(2), B, C, T, n_fft, hop_size, length, out.data_ptr(), _stream(), **rag)
    return out


def training_labels(stft_mix, stft_s1, stft_s2, feature_mix, db_threshold=40.0, with_cos=False):
    """Label-side features of a batch of chunks on the GPU: counterparts of get_one_hot,
    np.abs and get_cos_difference (onssen/data/feature_utils.py:77-95, wsj0_2mix.py:130-152).
    Inputs: (B,T,F,2) float32 STFTs as returned by stft_logmag, feature_mix (B,T,F).
    Returns one_hot (B,T,F,2) float32, mag_mix, mag_s1, mag_s2 (B,T,F) [, cos_s1, cos_s2]."""
    if not stft_mix.is_cuda:
        raise RuntimeError("training_labels: needs tensors on a ROCm device; onssen_amd has no CPU fallback")
    B, T, F, _ = stft_mix.shape
    dev = stft_mix.device
    stft_mix, stft_s1, stft_s2 = stft_mix.contiguous(), stft_s1.contiguous(), stft_s2.contiguous()
    feature_mix = feature_mix.contiguous()
    mk = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    one_hot, mm, m1, m2, umax = mk(B, T, F, 2), mk(B, T, F), mk(B, T, F), mk(B, T, F), mk(B)
    c1, c2 = (mk(B, T, F), mk(B, T, F)) if with_cos else (None, None)
    get_lib().labels(stft_mix.data_ptr(), stft_s1.data_ptr(), stft_s2.data_ptr(), feature_mix.data_ptr(), B, T, F,
                     float(db_threshold), umax.data_ptr(), one_hot.data_ptr(), mm.data_ptr(), m1.data_ptr(),
                     m2.data_ptr(), c1.data_ptr() if with_cos else None, c2.data_ptr() if with_cos else None, _stream())
    return (one_hot, mm, m1, m2, c1, c2) if with_cos else (one_hot, mm, m1, m2)
