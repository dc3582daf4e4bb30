"""Training losses needed by the data-parallel step.

Chimera losses (onssen/loss/loss_chimera.py): the deep-clustering term as below, the mask-inference term on
``onssen_loss_mask_f32`` (value; the winning speaker assignment) and ``onssen_loss_mask_grad_f32`` (gradient, one pass).
On a GPU the loss_dc value comes from ``onssen_loss_dc_f32`` (one pass over the embedding builds the Gram of ``[V | Y]``) and, when
autograd needs it, the gradient from ``onssen_loss_dc_grad_f32`` (``dV = Z M`` from the same Gram: one more pass); on the CPU
(tests, the gloo path) PyTorch ops with the same single-Gram forward and analytic backward.  ``ONSSEN_LOSS_HIP=0`` keeps the
PyTorch form on the GPU.
Semantics follow onssen/loss/loss_dc.py:6-44 and loss_util.py:4-11 exactly,
including their quirks: the affinity terms are Frobenius *norms* (not squared
norms) and the final product ``(B,) * (B,1)`` broadcasts to a (B, B) tensor
whose mean the trainer takes (onssen/utils/train.py:78-79).
"""
import os

import torch

from . import options


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
    needs_grad = torch.is_grad_enabled() and embedding.requires_grad
    if embedding.is_cuda and D + C <= 34 and (not needs_grad or (C <= 4 and options.get("loss") == "1")):
        emb, oh, mag = embedding.float().contiguous(), one_hot.contiguous(), mag_mix.float().contiguous()
        if not needs_grad:
            return _loss_dc_hip(emb, oh, mag, B, T * F, D, C)
        per_utt, total = _LossDcHip.apply(emb.view(B, T * F, D), oh, mag)
        return per_utt * total.unsqueeze(1)              # (B,) * (B,1) -> (B,B), as upstream
    V = embedding.reshape(B, T * F, D)
    Y = one_hot.reshape(B, T * F, C)
    mag = mag_mix.detach().reshape(B, T * F)
    total = mag.sum(1, keepdim=True)
    w = torch.sqrt(mag / total).unsqueeze(-1)            # W_i = |x_i| / sum_j |x_j|, applied to both factors
    scale = Y.sum(2, keepdim=True) * w                   # silent TF bins do not contribute
    per_utt = _AffinityNorms.apply(V, scale.detach(), (Y * w).detach())      # (B,)
    return per_utt * total                               # (B,) * (B,1) -> (B,B), as upstream


class _AffinityNorms(torch.autograd.Function):
    """||Vm^T Vm||_F - 2 ||Vm^T Ym||_F + ||Ym^T Ym||_F per utterance with Vm = scale * V (onssen/loss/loss_dc.py:36-42),
    as ONE Gram product of Z = [Vm | Ym] forward and ONE product backward instead of the three `bmm`s (and six in
    autograd's backward) of the literal form -- each contracts over all T*F bins.  Same value and gradient:
      d||Vm^T Vm|| = 2 Vm (Vm^T Vm) / ||.||,   d||Vm^T Ym|| = Ym (Vm^T Ym)^T / ||.||."""

    @staticmethod
    def forward(ctx, V, scale, Ym):
        D = V.shape[2]
        Z = torch.cat([V * scale, Ym], 2)
        G = torch.bmm(Z.transpose(1, 2), Z)              # (B, D+C, D+C): blocks Vm^T Vm, Vm^T Ym, Ym^T Ym
        nvv, nvy, nyy = _fro(G[:, :D, :D]), _fro(G[:, :D, D:]), _fro(G[:, D:, D:])
        ctx.save_for_backward(Z, scale, G, nvv, nvy)
        ctx.D = D
        return nvv - 2 * nvy + nyy

    @staticmethod
    def backward(ctx, g):
        Z, scale, G, nvv, nvy = ctx.saved_tensors
        D = ctx.D
        # dL/dVm = Z M with M = [2 Gvv / ||Gvv|| ; -2 Gvy^T / ||Gvy||]   (a zero norm has a zero block: no contribution)
        top = 2 * G[:, :D, :D] / nvv.clamp_min(1e-30)[:, None, None]
        bot = -2 * G[:, :D, D:].transpose(1, 2) / nvy.clamp_min(1e-30)[:, None, None]
        dVm = torch.bmm(Z, torch.cat([top, bot], 1))
        return dVm * scale * g[:, None, None], None, None


_WS = {}


def _loss_dc_launch(emb, one_hot, mag, B, TF, D, C, own_ws=False):
    from .hip import get_lib
    lib = get_lib()
    dev = emb.device
    nbytes = lib.loss_dc_workspace_bytes(B)
    if own_ws:      # kept alive by the autograd graph: the backward pass reads the partial Grams the forward left in it
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    else:
        ws = _WS.get((dev, B))
        if ws is None:
            ws = _WS[(dev, B)] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    per_utt = torch.empty(B, device=dev, dtype=torch.float32)
    total = torch.empty(B, device=dev, dtype=torch.float32)
    lib.loss_dc(emb.data_ptr(), one_hot.data_ptr(), mag.data_ptr(), B, TF, D, C, per_utt.data_ptr(), total.data_ptr(),
                ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    return per_utt, total, ws


def _loss_dc_hip(emb, one_hot, mag, B, TF, D, C):
    per_utt, total, _ = _loss_dc_launch(emb, one_hot, mag, B, TF, D, C)
    return per_utt * total.unsqueeze(1)                  # (B,) * (B,1) -> (B,B), as upstream


class _LossDcHip(torch.autograd.Function):
    """(per_utt, total_mag) of loss_dc with the embedding's gradient from onssen_loss_dc_grad_f32 (one_hot and mag_mix are
    labels: no gradient, like upstream's detached weights)."""

    @staticmethod
    def forward(ctx, emb, one_hot, mag):
        B, TF, D = emb.shape
        C = one_hot.shape[-1]
        per_utt, total, ws = _loss_dc_launch(emb, one_hot, mag, B, TF, D, C, own_ws=True)
        ctx.save_for_backward(emb, one_hot, mag, ws)
        ctx.mark_non_differentiable(total)
        return per_utt, total

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g, _g_total):
        from .hip import get_lib
        emb, one_hot, mag, ws = ctx.saved_tensors
        B, TF, D = emb.shape
        d_emb = torch.empty_like(emb)
        g = g.float().contiguous()
        get_lib().loss_dc_grad(emb.data_ptr(), one_hot.data_ptr(), mag.data_ptr(), B, TF, D, one_hot.shape[-1], g.data_ptr(),
                               d_emb.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        return d_emb, None, None


# ----------------------------------------------------------------------------- chimera / chimera++ losses
def _l1(x):
    return x.reshape(x.shape[0], -1).abs().sum(dim=1)


def _mask_term(mask_A, mask_B, mag_mix, t1, t2):
    l_ab = _l1(mask_A * mag_mix - t1) + _l1(mask_B * mag_mix - t2)
    l_ba = _l1(mask_B * mag_mix - t1) + _l1(mask_A * mag_mix - t2)
    return torch.min(l_ab, l_ba)


def _mask_term_hip(mask_A, mask_B, mag_mix, s1, s2, c1=None, c2=None):
    """The mask-inference term on the device (no autograd): one pass over the maps, onssen_loss_mask_f32."""
    from .hip import get_lib
    B = mag_mix.shape[0]
    TF = mag_mix[0].numel()
    base = getattr(mask_A, "_base", None)
    if (base is None or getattr(mask_B, "_base", None) is not base or mask_A.stride() != mask_B.stride()
            or mask_A.stride(2) * mask_A.shape[2] != mask_A.stride(1)):
        mask_A, mask_B = mask_A.contiguous(), mask_B.contiguous()   # not the strided views of one (B,T,F,2) buffer
    if mask_A.dtype != torch.float32 or mask_B.dtype != torch.float32:   # autocast / a user-supplied half estimate: the kernel reads fp32
        mask_A, mask_B = mask_A.float().contiguous(), mask_B.float().contiguous()
    f32 = lambda t: None if t is None else t.float().contiguous()
    mag, s1, s2, c1, c2 = f32(mag_mix), f32(s1), f32(s2), f32(c1), f32(c2)
    out = torch.empty(B, device=mag.device, dtype=torch.float32)
    lib = get_lib()
    ws = torch.empty(lib.loss_mask_workspace_bytes(B), dtype=torch.uint8, device=mag.device)
    # element (b, e = t*F + f) of a mask view sits at b*stride(0) + e*stride(2) when stride(1) = F*stride(2)
    lib.loss_mask(mask_A.data_ptr(), mask_B.data_ptr(), mask_A.stride(0), mask_A.stride(2), mag.data_ptr(), s1.data_ptr(),
                  s2.data_ptr(), c1.data_ptr() if c1 is not None else None, c2.data_ptr() if c2 is not None else None,
                  B, TF, out.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    return out


def _no_grad_needed(*ts):
    return not (torch.is_grad_enabled() and any(t.requires_grad for t in ts))


class _MaskTermHip(torch.autograd.Function):
    """The mask-inference term with its gradient on the device: ``onssen_loss_mask_f32`` picks the better
    speaker assignment per utterance in the forward pass, ``onssen_loss_mask_grad_f32`` writes d/d(mask_A, mask_B) in one pass
    over the maps.  ``masks`` is the interleaved (B, ..., 2) buffer both mask views come from (targets carry no gradient,
    like upstream's labels)."""

    @staticmethod
    def forward(ctx, masks, mag, s1, s2, c1, c2):
        from .hip import get_lib
        lib = get_lib()
        B = mag.shape[0]
        TF = mag[0].numel()
        out = torch.empty(B, device=mag.device, dtype=torch.float32)
        perm = torch.empty(B, device=mag.device, dtype=torch.int32)
        ws = torch.empty(lib.loss_mask_workspace_bytes(B), dtype=torch.uint8, device=mag.device)
        p = masks.data_ptr()
        lib.loss_mask(p, p + 4, 2 * TF, 2, mag.data_ptr(), s1.data_ptr(), s2.data_ptr(),
                      c1.data_ptr() if c1 is not None else None, c2.data_ptr() if c2 is not None else None, B, TF,
                      out.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream, perm=perm.data_ptr())
        ctx.save_for_backward(masks, mag, s1, s2, perm, *([c1, c2] if c1 is not None else []))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from .hip import get_lib
        masks, mag, s1, s2, perm, *cs = ctx.saved_tensors
        c1, c2 = cs if cs else (None, None)
        B = mag.shape[0]
        TF = mag[0].numel()
        d = torch.empty_like(masks)
        g = g.float().contiguous()
        p, q = masks.data_ptr(), d.data_ptr()
        get_lib().loss_mask_grad(p, p + 4, 2 * TF, 2, mag.data_ptr(), s1.data_ptr(), s2.data_ptr(),
                                 c1.data_ptr() if c1 is not None else None, c2.data_ptr() if c2 is not None else None, B, TF,
                                 g.data_ptr(), perm.data_ptr(), q, q + 4, 2 * TF, 2, torch.cuda.current_stream().cuda_stream)
        return d, None, None, None, None, None


def _mask_term_hip_autograd(mask_A, mask_B, mag_mix, s1, s2, c1=None, c2=None):
    """Mask term WITH a gradient on the HIP kernels.  The network's two masks are strided views of one (B,T,F,2) buffer
    (onssen/nn/chimera.py:42-45): the Function is applied to that buffer itself (the autograd graph runs through the root
    tensor, not through two select views whose gradients would be scattered into zeros and added); anything else is
    interleaved by torch.stack first."""
    base = getattr(mask_A, "_base", None)
    dense, acc = [], 2                              # strides of a map whose elements are 2 apart in a contiguous buffer
    for n_i in reversed(mask_A.shape):
        dense.insert(0, acc)
        acc *= n_i
    if not (base is not None and getattr(mask_B, "_base", None) is base and base.dtype == torch.float32 and base.is_contiguous()
            and base.numel() == 2 * mask_A.numel() and mask_A.stride() == mask_B.stride() == tuple(dense)
            and mask_A.storage_offset() == base.storage_offset() and mask_B.storage_offset() == base.storage_offset() + 1):
        base = torch.stack([mask_A.float(), mask_B.float()], -1)
    f32 = lambda t: None if t is None else t.detach().float().contiguous()
    return _MaskTermHip.apply(base, f32(mag_mix), f32(s1), f32(s2), f32(c1), f32(c2))


def _use_hip_mask_grad(mask_A):
    return mask_A.is_cuda and options.get("loss") == "1"


def loss_chimera_msa(output, label):
    """onssen/loss/loss_chimera.py:6-31: 0.975 * loss_dc + 0.025 * magnitude-spectrum-approximation mask loss with the
    better of the two speaker assignments ((B,B) like loss_dc, as upstream)."""
    embedding, mask_A, mask_B = output
    one_hot, mag_mix, mag_s1, mag_s2 = label
    le = loss_dc([embedding], [one_hot, mag_mix])
    if mask_A.is_cuda and _no_grad_needed(mask_A, mask_B):
        lm = _mask_term_hip(mask_A, mask_B, mag_mix, mag_s1, mag_s2)
    elif _use_hip_mask_grad(mask_A):
        lm = _mask_term_hip_autograd(mask_A, mask_B, mag_mix, mag_s1, mag_s2)
    else:
        lm = _mask_term(mask_A, mask_B, mag_mix, mag_s1, mag_s2)
    return le * 0.975 + lm * 0.025


def loss_chimera_psa(output, label):
    """onssen/loss/loss_chimera.py:33-59: as MSA with the phase-sensitive targets min(|x|, relu(|s| cos(theta)))."""
    embedding, mask_A, mask_B = output
    one_hot, mag_mix, mag_s1, mag_s2, cos_s1, cos_s2 = label
    le = loss_dc([embedding], [one_hot, mag_mix])
    if mask_A.is_cuda and _no_grad_needed(mask_A, mask_B):
        lm = _mask_term_hip(mask_A, mask_B, mag_mix, mag_s1, mag_s2, cos_s1, cos_s2)
    elif _use_hip_mask_grad(mask_A):
        lm = _mask_term_hip_autograd(mask_A, mask_B, mag_mix, mag_s1, mag_s2, cos_s1, cos_s2)
    else:
        t1 = torch.min(mag_mix, torch.relu(mag_s1 * cos_s1))
        t2 = torch.min(mag_mix, torch.relu(mag_s2 * cos_s2))
        lm = _mask_term(mask_A, mask_B, mag_mix, t1, t2)
    return le * 0.975 + lm * 0.025
