"""End-to-end separation: waveform -> K1..K10 -> waveforms.

Counterpart of tester.eval + get_est_sig (onssen/utils/test.py:29-41,
egs/wsj0-2mix/chimera/evaluate.py:23-45, deep_clustering/evaluate.py:22-47)
with the device->host->device hop removed for the mask-inference models."""
import numpy as np
import torch

from .features import mask_istft, stft_logmag


@torch.no_grad()
def separate_chimera(model, wav, window_size=256, hop_size=64):
    """wav (B, n) cuda float32 -> (B, 2, n): masks straight from the network."""
    logmag, ri = stft_logmag(wav, window_size, hop_size)
    _, masks = model.embedding_and_masks(logmag)
    return mask_istft(ri, masks, hop_size, wav.shape[-1])


@torch.no_grad()
def separate_dc(model, wav, window_size=256, hop_size=64, num_spk=2, db_threshold=40.0):
    """Deep-clustering back end: bins with feature >= max - 40/20 are clustered
    with KMeans(n_clusters=num_spk, random_state=0) on the host (sklearn, as
    upstream; 'next' row N2 moves it on-device), binary masks, silent bins 0 in
    both masks; mask-apply + iSTFT on the GPU."""
    from sklearn.cluster import KMeans
    logmag, ri = stft_logmag(wav, window_size, hop_size)
    emb, = model([logmag])
    B, T, F, D = emb.shape
    masks = torch.zeros(B, T, F, num_spk, device=wav.device, dtype=torch.float32)
    for b in range(B):   # upstream evaluates with batch 1 (evaluate.py:34-35)
        feat = logmag[b]
        act = feat >= (feat.max() - db_threshold / 20.0)
        e = emb[b][act].cpu().numpy()
        label = KMeans(n_clusters=num_spk, random_state=0, n_init=10).fit_predict(e)
        lab = torch.from_numpy(label.astype(np.int64)).to(wav.device)
        m = torch.zeros(int(act.sum()), num_spk, device=wav.device)
        m[:, 0] = lab.float()
        m[:, 1] = 1.0 - lab.float()
        masks[b][act] = m
    return mask_istft(ri, masks, hop_size, wav.shape[-1])
