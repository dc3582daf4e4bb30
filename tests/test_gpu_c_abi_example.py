"""The drop-in boundary without Python in the compute path: examples/separate_dc.c -- a plain C99 host program over
include/onssen_hip.h and the HIP runtime, no torch -- separates a batch of mixtures with the parameters of a
``deep_clustering`` module and must reproduce ``onssen_amd.separation.separate_dc`` bit for bit (same library, same launch
sequence; what egs/wsj0-2mix/deep_clustering/evaluate.py:31-45 computes)."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "separate_dc")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from onssen_amd.hip import get_lib
    get_lib()   # fail loudly if libonssen_hip.so is missing
    return torch.device("cuda:0")


def _write_case(path, model, wav, n_fft, hop):
    B, n = wav.shape
    H, L, D = model.hidden_dim, model.num_layers, model.embedding_dim
    sd = {k: v.detach().cpu().float().numpy() for k, v in model.state_dict().items()}
    with open(path, "wb") as f:
        f.write(struct.pack("<8i", 0x44435345, B, n, n_fft, hop, H, L, D))
        for l in range(L):
            for sfx in ("", "_reverse"):
                for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                    f.write(np.ascontiguousarray(sd[f"rnn.{name}_l{l}{sfx}"]).tobytes())
        for name in ("bn.weight", "bn.bias", "bn.running_mean", "bn.running_var"):
            f.write(np.ascontiguousarray(sd[name]).tobytes())
        f.write(struct.pack("<f", float(model.bn.eps)))
        f.write(np.ascontiguousarray(sd["fc_dc.weight"]).tobytes())
        f.write(np.ascontiguousarray(sd["fc_dc.bias"]).tobytes())
        f.write(np.ascontiguousarray(wav.cpu().numpy(), dtype=np.float32).tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("B,n,H,L", [(4, 6400, 600, 2), (3, 9000, 300, 3), (1, 25536, 600, 2)])
def test_c_host_program_reproduces_separate_dc(dev, tmp_path, monkeypatch, B, n, H, L):
    from onssen_amd import nn as onn
    from onssen_amd.separation import separate_dc
    from onssen_amd.synthetic import synth_mixture
    if not os.path.exists(EXE):
        import __graft_entry__ as g          # (normally built beforehand; gcc and the library are on the GPU box too)
        g.build()
    assert os.path.exists(EXE), "examples/separate_dc is not built: run __graft_entry__.build()"
    monkeypatch.setenv("ONSSEN_FUSE_IN0", "0")          # the example runs the unfused launch sequence (the default up to 16 rows)
    torch.manual_seed(B + H)
    model = onn.deep_clustering(129, H, L, 20).to(dev).eval()
    with torch.no_grad():                               # non-trivial BatchNorm statistics
        model.bn.running_mean.normal_(0.0, 0.1)
        model.bn.running_var.uniform_(0.5, 1.5)
        model.bn.weight.uniform_(0.8, 1.2)
        model.bn.bias.normal_(0.0, 0.1)
    wav = torch.from_numpy(np.stack([synth_mixture(700 + b, n) for b in range(B)]).astype(np.float32)).to(dev)
    with torch.no_grad():
        ref = separate_dc(model, wav).cpu().numpy()
    src, dst = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    _write_case(src, model, wav, 256, 64)
    run = subprocess.run([EXE, src, dst], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    got = np.fromfile(dst, dtype=np.float32).reshape(B, 2, n)
    np.testing.assert_array_equal(got, ref)
    assert np.abs(ref).max() > 1e-3                     # it did separate something
