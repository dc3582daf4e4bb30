"""Evaluation metric of the separation recipes on the device: batch SI-SDR with the best source
permutation, same signature and results as onssen.evaluate.batch_SDR_torch (onssen/evaluate/sdr.py:40-87)."""
import torch

from .hip import get_lib

_WS = {}


def batch_SDR_torch(estimation, origin, mask=None, return_perm=False):
    """estimation, origin (batch, nsource, nsample); mask optional (batch, nsample).  Returns (batch,) mean SDR of the
    best permutation (and, with return_perm, the permutation's index in lexicographic order, like upstream)."""
    assert estimation.size() == origin.size(), "Estimation and original sources should have same shape."
    B, C, n = estimation.size()
    assert C < n, "Axis 1 should be the number of sources, and axis 2 should be the signal."
    if not estimation.is_cuda:
        raise RuntimeError("onssen_amd.evaluate.batch_SDR_torch runs on a ROCm device; there is no CPU fallback")
    if C > 4:
        raise ValueError("onssen_batch_sdr_f32 scans the permutations of at most 4 sources")
    lib, dev = get_lib(), estimation.device
    est, org = estimation.float().contiguous(), origin.float().contiguous()
    mk = mask.float().contiguous() if mask is not None else None
    ws = _WS.get((dev, B))
    if ws is None:
        ws = _WS[(dev, B)] = torch.empty(lib.batch_sdr_workspace_bytes(B), dtype=torch.uint8, device=dev)
    sdr = torch.empty(B, device=dev, dtype=torch.float32)
    perm = torch.empty(B, device=dev, dtype=torch.int32)
    lib.batch_sdr(est.data_ptr(), org.data_ptr(), mk.data_ptr() if mk is not None else None, B, C, n, sdr.data_ptr(),
                  perm.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    return (sdr, perm.long()) if return_perm else sdr
