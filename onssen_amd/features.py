"""GPU front end / back end of the separation path (K1, K2, K10).

Counterparts of onssen/data/feature_utils.py (get_stft, get_log_magnitude,
get_phase) and of the mask-apply + librosa.istft tail of
egs/wsj0-2mix/*/evaluate.py, operating on device tensors."""
import torch

from .hip import get_lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def stft_logmag(wav, window_size=256, hop_size=64, epsilon=1e-7, return_stft=True):
    """wav (B, n) or (n,) float32 cuda tensor ->
    (log_magnitude (B,T,F) float32, stft_ri (B,T,F,2) float32 | None).

    stft_ri[..., 0] + 1j*stft_ri[..., 1] is what feature_utils.get_stft
    returns per utterance (frame x frequency complex64; librosa<0.10 defaults:
    periodic Hann, centred, reflect padding); stft_ri itself is get_phase."""
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
    get_lib().stft_logmag(wav.data_ptr(), B, n, wav.stride(0), window_size, hop_size, float(epsilon),
                          logmag.data_ptr(), ri.data_ptr() if ri is not None else None, _stream())
    return logmag, ri


def mask_istft(stft_ri, masks, hop_size=64, length=None):
    """stft_ri (B,T,F,2); masks (B,T,F,C) (any strides) or None ->
    (B, C, length) float32: istft(stft * mask_c) per speaker, librosa
    semantics (hop, length=nsample)."""
    if not stft_ri.is_cuda:
        raise RuntimeError("mask_istft: needs tensors on a ROCm device; onssen_amd has no CPU fallback")
    stft_ri = stft_ri.float().contiguous()
    B, T, F, _ = stft_ri.shape
    n_fft = 2 * (F - 1)
    if length is None:
        length = hop_size * (T - 1)
    lib = get_lib()
    if masks is None:
        out = torch.empty(B, 1, length, device=stft_ri.device, dtype=torch.float32)
        lib.mask_istft(stft_ri.data_ptr(), None, 0, 0, 0, 0, B, 1, T, n_fft, hop_size, length, out.data_ptr(),
                       _stream())
        return out
    masks = masks.float()
    C = masks.shape[3]
    out = torch.empty(B, C, length, device=stft_ri.device, dtype=torch.float32)
    lib.mask_istft(stft_ri.data_ptr(), masks.data_ptr(), masks.stride(0), masks.stride(3), masks.stride(1),
                   masks.stride(2), B, C, T, n_fft, hop_size, length, out.data_ptr(), _stream())
    return out
