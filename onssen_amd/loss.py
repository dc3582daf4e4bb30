"""Training losses needed by the data-parallel step.

With autograd (training): stock PyTorch ops.  Without (``trainer.validate``, onssen/utils/train.py:91-99, runs the
loss under ``no_grad``): the HIP kernel ``onssen_loss_dc_f32``, which streams the embedding once.
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
    if embedding.is_cuda and not (torch.is_grad_enabled() and embedding.requires_grad) and D + C <= 34:
        return _loss_dc_hip(embedding.float().contiguous(), one_hot.contiguous(), mag_mix.float().contiguous(), B, T * F, D, C)
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


_WS = {}


def _loss_dc_hip(emb, one_hot, mag, B, TF, D, C):
    from .hip import get_lib
    lib = get_lib()
    dev = emb.device
    nbytes = lib.loss_dc_workspace_bytes(B)
    ws = _WS.get((dev, B))
    if ws is None:
        ws = _WS[(dev, B)] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    per_utt = torch.empty(B, device=dev, dtype=torch.float32)
    total = torch.empty(B, device=dev, dtype=torch.float32)
    lib.loss_dc(emb.data_ptr(), one_hot.data_ptr(), mag.data_ptr(), B, TF, D, C, per_utt.data_ptr(), total.data_ptr(),
                ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    return per_utt * total.unsqueeze(1)                  # (B,) * (B,1) -> (B,B), as upstream
