"""Training losses needed by the data-parallel step.

Stock PyTorch ops with autograd (the fused HIP version is a 'next' row).
Semantics follow onssen/loss/loss_dc.py:6-44 and loss_util.py:4-11 exactly,
including their quirks: the affinity terms are Frobenius *norms* (not squared
norms) and the final product ``(B,) * (B,1)`` broadcasts to a (B, B) tensor
whose mean the trainer takes (onssen/utils/train.py:78-79).
"""
import torch


def _fro(x):
    return torch.sqrt((x * x).flatten(1).sum(dim=1))


def loss_dc(output, label):
    assert len(output) == 1, "Number of output must be 1 for Deep Clustering"
    assert len(label) == 2, "Number of label must be 2 for Deep Clustering"
    embedding, = output
    one_hot, mag_mix = label
    one_hot = one_hot.float()
    B, T, F, C = one_hot.shape
    D = embedding.shape[-1]
    V = embedding.reshape(B, T * F, D)
    Y = one_hot.reshape(B, T * F, C)
    mag = mag_mix.detach().reshape(B, T * F)
    V = Y.sum(2, keepdim=True) * V                       # silent TF bins do not contribute
    total = mag.sum(1, keepdim=True)
    w = torch.sqrt(mag / total).unsqueeze(-1)            # W_i = |x_i| / sum_j |x_j|, applied to both factors
    V, Y = V * w, Y * w
    vtv = torch.bmm(V.transpose(1, 2), V)
    vty = torch.bmm(V.transpose(1, 2), Y)
    yty = torch.bmm(Y.transpose(1, 2), Y)
    per_utt = _fro(vtv) - 2 * _fro(vty) + _fro(yty)      # (B,)
    return per_utt * total                               # (B,) * (B,1) -> (B,B), as upstream
