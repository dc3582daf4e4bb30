"""Parity of the HIP path (through the C ABI / onssen_amd modules) against the
oracle and the committed golden vectors.  Runs on the GPU box only (-m gpu);
nothing here reads /root/reference.

Tolerances (fp32 path): elementwise |a-b| <= 1e-5 + 1e-4*|b| on
embeddings / masks, per-vector rel-L2 <= 1e-4; log-magnitude is compared
through |X| (log10 of a near-zero bin amplifies fp32 round-off)."""
import numpy as np
import pytest
import torch

from onssen_amd import _abi
from onssen_amd.synthetic import make_state_dict, synth_mixture
from oracle import np_oracle as O
from oracle import torch_cpu as TC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from onssen_amd.hip import get_lib
    get_lib()   # fail loudly if libonssen_hip.so is missing
    return torch.device("cuda:0")


@pytest.fixture(params=["bf16x3", "bf16x3-steps", "f32", "f32-steps"])
def prec(request, monkeypatch):
    """The product's forward paths: split-bf16 with the XCD-local persistent recurrence (default), split-bf16 with one
    launch per time step, exact-fp32 MFMA in the persistent recurrence (round 3) and with one launch per time step.
    Tolerances: exact fp32 -> |a-b| <= 1e-5 + 1e-4|b| (measured <= 1.4e-6); split-bf16 (~1e-5 relative per dot
    product) -> |a-b| <= 3e-5 + 1e-4|b| on the TINY fixtures and the gain >= 1.5 oracle shapes (round 6: tightened from 5e-5 to what
    the runs support with margin -- the worst case of the whole suite is 2.4e-5 on g1_deep_clustering_H8_L1, whose gain-2.0 weights
    and 16-wide head leave some 20-vectors with a small norm in front of F.normalize, which amplifies the absolute error; every
    other comparison is <= 1.4e-5) and the STRICT 1e-5 at BASELINE sizes (test_full_size_golden_subsample: measured <= 4.2e-6);
    per-vector rel-L2 <= 1e-4 (the north-star figure) in all.  Every comparison prints what it measured (``check``) and the
    numbers are written to gpurun_out/parity_errors.json (committed per round as profiles/rNN_parity_errors.json)."""
    monkeypatch.setenv("ONSSEN_PRECISION", request.param.split("-")[0])
    monkeypatch.setenv("ONSSEN_XCD", "0" if request.param.endswith("steps") else "1")
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    return {"name": request.param, "atol": 1e-5 if request.param.startswith("f32") else 3e-5}


PARITY_LOG = {}      # test id -> measured errors; written to gpurun_out/parity_errors.json when the module ends (profiles/r06_parity_errors.json)


def check(tag, got, ref, atol, rtol=1e-4, vectors=True):
    """assert_allclose that first PRINTS what it measured (max |a-b|, the worst ratio |a-b| / (atol + rtol|b|), max per-vector
    rel-L2), so that the numbers behind every golden comparison can be read from the GPU test log / the JSON beside it."""
    got, ref = np.asarray(got), np.asarray(ref)
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    rec = {"max_abs": float(d.max()), "worst_ratio_of_bound": float((d / (atol + rtol * np.abs(ref))).max()), "atol": atol, "rtol": rtol}
    if vectors and got.ndim >= 2 and got.shape[-1] > 1:
        rec["max_rel_l2"] = float(rel_l2(got, ref).max())
    PARITY_LOG[tag] = rec
    print(f"parity {tag}: max abs {rec['max_abs']:.3e} ({rec['worst_ratio_of_bound']:.2f} of the bound atol {atol:g} + {rtol:g}|b|)"
          + (f", max rel-L2 {rec['max_rel_l2']:.3e}" if "max_rel_l2" in rec else ""))
    np.testing.assert_allclose(got, ref, atol=atol, rtol=rtol, err_msg=tag)


@pytest.fixture(scope="module", autouse=True)
def _write_parity_log():
    yield
    import json, os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_errors.json"), "w") as f:
            json.dump(PARITY_LOG, f, indent=1, sort_keys=True)
    except OSError:
        pass


def rel_l2(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return np.linalg.norm(a - b, axis=-1) / np.maximum(np.linalg.norm(b, axis=-1), 1e-30)


def build(kind, z_or_cfg, dev):
    from onssen_amd import nn as onn
    c = {k: (z_or_cfg[k].item() if hasattr(z_or_cfg[k], "item") else z_or_cfg[k]) for k in
         ("F", "H", "L", "D", "C", "seed", "gain")}
    sd = make_state_dict(kind, int(c["F"]), int(c["H"]), int(c["L"]), int(c["D"]), int(c["C"]), seed=int(c["seed"]),
                         gain=float(c["gain"]))
    cls = {"deep_clustering": onn.deep_clustering, "chimera": onn.chimera, "phase_net": onn.phase_net}[kind]
    m = cls(int(c["F"]), int(c["H"]), int(c["L"]), int(c["D"]))
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.to(dev).eval(), sd


def logmag_input(seed, B, T):
    return np.stack([O.log_magnitude(O.stft(synth_mixture(seed * 100 + b, (T - 1) * 64), 256, 64))
                     for b in range(B)])


# ---------------------------------------------------------------- kernels through the ABI
@pytest.mark.parametrize("M,K,N,mode,group", [(1000, 1200, 2580, _abi.EPI_L2NORM, 20), (333, 129, 4800, _abi.EPI_BIAS, 0),
                                              (257, 1200, 258, _abi.EPI_SIGMOID, 0), (128, 64, 80, _abi.EPI_BIAS, 0)])
def test_linear_kernel(dev, M, K, N, mode, group):
    from onssen_amd.hip import get_lib
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g)
    ldw = (K + 3) // 4 * 4
    W = torch.zeros(N, ldw)
    W[:, :K] = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    Ad, Wd, bd = A.to(dev), W.to(dev), bias.to(dev)
    out = torch.full((M, N), float("nan"), device=dev)
    get_lib().linear(Ad.data_ptr(), K, 0, 1, M, K, Wd.data_ptr(), ldw, bd.data_ptr(), N, mode, group, 1e-12, None,
                     out.data_ptr(), N, 0, torch.cuda.current_stream().cuda_stream)
    ref = A.double() @ W[:, :K].double().T + bias.double()
    if mode == _abi.EPI_L2NORM:
        r = ref.view(M, N // group, group)
        ref = (r / r.norm(dim=-1, keepdim=True).clamp_min(1e-12)).view(M, N)
    elif mode == _abi.EPI_SIGMOID:
        ref = torch.sigmoid(ref)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("M,K,N,mode,group", [(1000, 1200, 2580, _abi.EPI_L2NORM, 20), (333, 129, 4800, _abi.EPI_BIAS, 0),
                                              (257, 1200, 258, _abi.EPI_SIGMOID, 0), (12800, 1200, 4800, _abi.EPI_BIAS, 0)])
def test_linear_split_bf16_kernel(dev, M, K, N, mode, group):
    from onssen_amd.hip import get_lib
    lib = get_lib()
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    Ad, Wd, bd = A.to(dev), W.to(dev), bias.to(dev)
    ld = (K + 31) // 32 * 32
    planes = torch.empty(2, N, ld, device=dev, dtype=torch.int16)
    st = torch.cuda.current_stream().cuda_stream
    lib.linear_pack_bf16x3(Wd.data_ptr(), N, K, K, ld, planes.data_ptr(), st)
    out = torch.full((M, N), float("nan"), device=dev)
    lib.linear_bf16x3(Ad.data_ptr(), K, 0, 1, M, K, planes.data_ptr(), ld, bd.data_ptr(), N, mode, group, 1e-12, None,
                      out.data_ptr(), N, 0, st)
    ref = A.double() @ W.double().T + bias.double()
    if mode == _abi.EPI_L2NORM:
        r = ref.view(M, N // group, group)
        ref = (r / r.norm(dim=-1, keepdim=True).clamp_min(1e-12)).view(M, N)
    elif mode == _abi.EPI_SIGMOID:
        ref = torch.sigmoid(ref)
    err = (out.cpu().double() - ref).abs().max().item()
    print(f"split-bf16 GEMM M={M} K={K} N={N} mode={mode}: max abs err {err:.3e}")
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=5e-5, rtol=1e-4)


@pytest.mark.parametrize("M,K,N,mode,group", [(1000, 1200, 2580, _abi.EPI_L2NORM, 20), (333, 129, 4800, _abi.EPI_BIAS, 0),
                                              (257, 1200, 258, _abi.EPI_SIGMOID, 0), (12800, 1200, 4800, _abi.EPI_BIAS, 0),
                                              (700, 600, 320, _abi.EPI_L2NORM, 40)])
def test_linear_x3_image_kernel(dev, M, K, N, mode, group):
    """onssen_x3_image_f32 + onssen_linear_x3p: pre-split operands, register epilogue."""
    from onssen_amd.hip import get_lib
    lib = get_lib()
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    Ad, Wd, bd = A.to(dev), W.to(dev), bias.to(dev)
    KB = (K + 31) // 32
    a_img = torch.empty(M, KB, 2, 32, device=dev, dtype=torch.int16)
    w_img = torch.empty(N, KB, 2, 32, device=dev, dtype=torch.int16)
    st = torch.cuda.current_stream().cuda_stream
    lib.x3_image(Ad.data_ptr(), K, 0, 1, M, K, a_img.data_ptr(), st)
    lib.x3_image(Wd.data_ptr(), K, 0, 1, N, K, w_img.data_ptr(), st)
    out = torch.full((M, N), float("nan"), device=dev)
    lib.linear_x3p(a_img.data_ptr(), M, K, w_img.data_ptr(), bd.data_ptr(), N, mode, group, 1e-12, out.data_ptr(), 1, N, 0, st)
    ref = A.double() @ W.double().T + bias.double()
    if mode == _abi.EPI_L2NORM:
        r = ref.view(M, N // group, group)
        ref = (r / r.norm(dim=-1, keepdim=True).clamp_min(1e-12)).view(M, N)
    elif mode == _abi.EPI_SIGMOID:
        ref = torch.sigmoid(ref)
    err = (out.cpu().double() - ref).abs().max().item()
    print(f"x3-image GEMM M={M} K={K} N={N} mode={mode}: max abs err {err:.3e}")
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=5e-5, rtol=1e-4)


def test_linear_x3_image_kernel_plain_bf16_products(dev):
    """ONSSEN_EPI_BF16: only the hi halves of the images are multiplied -- exactly the GEMM of the bf16-rounded operands
    with fp32 accumulation."""
    from onssen_amd.hip import get_lib
    lib = get_lib()
    M, K, N = 700, 1200, 480
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / K ** 0.5; bias = torch.randn(N, generator=g)
    Ad, Wd, bd = A.to(dev), W.to(dev), bias.to(dev)
    KB = (K + 31) // 32
    a_img = torch.empty(M, KB, 2, 32, device=dev, dtype=torch.int16); w_img = torch.empty(N, KB, 2, 32, device=dev, dtype=torch.int16)
    st = torch.cuda.current_stream().cuda_stream
    lib.x3_image(Ad.data_ptr(), K, 0, 1, M, K, a_img.data_ptr(), st)
    lib.x3_image(Wd.data_ptr(), K, 0, 1, N, K, w_img.data_ptr(), st)
    out = torch.full((M, N), float("nan"), device=dev)
    lib.linear_x3p(a_img.data_ptr(), M, K, w_img.data_ptr(), bd.data_ptr(), N, _abi.EPI_BIAS | _abi.EPI_BF16, 0, 0.0,
                   out.data_ptr(), 1, N, 0, st)
    ref = A.bfloat16().double() @ W.bfloat16().double().T + bias.double()
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=2e-5, rtol=1e-5)
    full = A.double() @ W.double().T + bias.double()
    assert (out.cpu().double() - full).abs().max() < 0.05       # ... and ~2^-9 per product away from the fp32 result


# ---------------------------------------------------------------- golden vectors of the reference
@pytest.mark.parametrize("name", ["g1_deep_clustering_H8_L1", "g1_deep_clustering_H32_L2"])
def test_dc_tiny_golden(dev, golden_dir, prec, name):
    z = np.load(f"{golden_dir}/{name}.npz")
    m, _ = build("deep_clustering", z, dev)
    with torch.no_grad():
        emb, = m([torch.from_numpy(z["x"]).to(dev)])
    emb = emb.cpu().numpy()
    assert emb.shape == z["out_embedding"].shape
    check(f"{name}[{prec['name']}] embedding", emb, z["out_embedding"], prec["atol"])
    assert rel_l2(emb, z["out_embedding"]).max() < 1e-4


def test_chimera_tiny_golden(dev, golden_dir, prec):
    z = np.load(f"{golden_dir}/g1_chimera_H32_L2.npz")
    m, _ = build("chimera", z, dev)
    with torch.no_grad():
        e, a, b = m([torch.from_numpy(z["x"]).to(dev)])
    assert a.shape == z["out_mask_A"].shape and not a.is_contiguous()   # strided views like upstream
    check(f"g1_chimera_H32_L2[{prec['name']}] embedding", e.cpu().numpy(), z["out_embedding"], prec["atol"])
    check(f"g1_chimera_H32_L2[{prec['name']}] mask_A", a.cpu().numpy(), z["out_mask_A"], prec["atol"], vectors=False)
    check(f"g1_chimera_H32_L2[{prec['name']}] mask_B", b.cpu().numpy(), z["out_mask_B"], prec["atol"], vectors=False)


def test_phase_net_tiny_golden(dev, golden_dir, prec):
    z = np.load(f"{golden_dir}/g1_phase_net_H16_L2.npz")
    m, _ = build("phase_net", z, dev)
    with torch.no_grad():
        outs = m([torch.from_numpy(z["x"]).to(dev), torch.from_numpy(z["x_phase"]).to(dev)])
    for o, n in zip(outs, ["embedding", "mask_A", "mask_B", "phase_A", "phase_B"]):
        # 2-vector normalisation amplifies round-off: (re, im) of a bin whose raw pair is short (measured: 2.5e-4 split-bf16, 1.7e-5 fp32)
        tol = (1e-4 if prec["atol"] == 1e-5 else 4e-4) if n.startswith("phase") else prec["atol"]
        check(f"g1_phase_net_H16_L2[{prec['name']}] {n}", o.cpu().numpy(), z["out_" + n], tol, vectors=n == "embedding")


@pytest.mark.parametrize("tag,kind", [("cfg1_dc_L2", "deep_clustering"), ("cfg1_dc_L3", "deep_clustering"),
                                      ("cfg3_chimera_L4", "chimera")])
def test_full_size_golden_subsample(dev, golden_dir, prec, tag, kind):
    """BASELINE configs at full width (H=600, T=400) against the reference's
    strided output subsample and per-frame checksums."""
    z = np.load(f"{golden_dir}/g2_{tag}.npz")
    m, _ = build(kind, z, dev)
    x = logmag_input(int(z["x_seed"]), int(z["B"]), int(z["T"]))
    with torch.no_grad():
        outs = m([torch.from_numpy(x).to(dev)])
    emb = outs[0].cpu().numpy()
    err, rl2 = np.abs(emb[:, ::40, ::16, :] - z["emb_sub"]).max(), rel_l2(emb[:, ::40, ::16, :], z["emb_sub"]).max()
    print(f"[{prec['name']}] {tag}: max abs err {err:.3e}, max per-vector rel-L2 {rl2:.3e}")
    # at BASELINE sizes and PyTorch-scale weights both modes meet the strict elementwise bound
    check(f"g2_{tag}[{prec['name']}] embedding subsample", emb[:, ::40, ::16, :], z["emb_sub"], 1e-5)
    assert rl2 < 1e-4
    np.testing.assert_allclose(emb.astype(np.float64).sum(axis=(2, 3)), z["emb_sum_per_frame"], atol=5e-3)
    if kind == "chimera":
        check(f"g2_{tag}[{prec['name']}] mask_A subsample", outs[1].cpu().numpy()[:, ::8, :], z["mask_A_sub"], 1e-5, vectors=False)
        check(f"g2_{tag}[{prec['name']}] mask_B subsample", outs[2].cpu().numpy()[:, ::8, :], z["mask_B_sub"], 1e-5, vectors=False)


@pytest.mark.parametrize("tag,kind", [("cfg1_dc_L2", "deep_clustering"), ("cfg3_chimera_L4", "chimera")])
def test_optin_bf16_mode_full_size(dev, golden_dir, monkeypatch, tag, kind):
    """ONSSEN_PRECISION=bf16 (opt-in; BASELINE cfg2's literal dtype): plain bf16 products, fp32 accumulate / gates / state.
    OUTSIDE the 1e-4 contract -- its own tolerance: per-vector rel-L2 <= 1.5e-2 against the fp32 reference
    (the reference itself run in bf16 shows 3.7e-3 mean / 9.8e-3 max); masks within 1e-2 abs."""
    monkeypatch.setenv("ONSSEN_PRECISION", "bf16")
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    z = np.load(f"{golden_dir}/g2_{tag}.npz")
    m, _ = build(kind, z, dev)
    x = logmag_input(int(z["x_seed"]), int(z["B"]), int(z["T"]))
    with torch.no_grad():
        outs = m([torch.from_numpy(x).to(dev)])
    emb = outs[0].cpu().numpy()
    rl2 = rel_l2(emb[:, ::40, ::16, :], z["emb_sub"])
    print(f"[bf16] {tag}: per-vector rel-L2 mean {rl2.mean():.3e} max {rl2.max():.3e}")
    assert 1e-4 < rl2.max() < 1.5e-2        # engaged (not the split-bf16 path), and inside the mode's own budget
    np.testing.assert_allclose(np.linalg.norm(emb, axis=-1), 1.0, atol=1e-5)   # normalisation stays fp32
    if kind == "chimera":
        np.testing.assert_allclose(outs[1].cpu().numpy()[:, ::8, :], z["mask_A_sub"], atol=1e-2)
        np.testing.assert_allclose(outs[2].cpu().numpy()[:, ::8, :], z["mask_B_sub"], atol=1e-2)


def test_optin_bf16_mode_ragged_fused(dev, monkeypatch):
    """The same mode on a ragged batch > 32 rows (first layer's input projection fused into the recurrence launch)."""
    monkeypatch.setenv("ONSSEN_PRECISION", "bf16")
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    cfg = dict(F=129, H=128, L=2, D=20, C=2, seed=3, gain=1.5)
    m, sd = build("deep_clustering", cfg, dev)
    x = logmag_input(11, 70, 9)
    ref = TC.deep_clustering_forward(sd, x).numpy()
    with torch.no_grad():
        emb, = m([torch.from_numpy(x).to(dev)])
    rl2 = rel_l2(emb.cpu().numpy(), ref)
    print(f"[bf16] ragged fused: per-vector rel-L2 mean {rl2.mean():.3e} max {rl2.max():.3e}")
    assert 1e-4 < rl2.max() < 1.5e-2


@pytest.mark.parametrize("B,T", [(32, 400), (32, 24), (32, 6), (32, 2), (16, 150), (64, 150)])
def test_cfg2_literal_bf16_pinned_to_rounded_oracle(dev, monkeypatch, B, T):
    """BASELINE cfg2 in its literal dtype at its own batch (DC 2 x BLSTM-600, 32 x 400 frames; also the 4-row-group and
    16-row-group forms): ``ONSSEN_PRECISION=bf16`` against (i) the fp32 oracle inside the mode's own 1.5e-2 rel-L2 budget
    and (ii) the oracle's restatement of THAT arithmetic (``deep_clustering_forward_rounded``: bf16-rounded operands, fp32
    everything else), so that the mode is pinned and not only banded.

    (ii) cannot be bit-exact, and how close it gets depends on the sequence length: the kernel and NumPy accumulate in
    different orders, a 1e-7 difference in an h that sits on a bf16 rounding boundary moves that operand by one bf16 ulp
    (2^-8 relative), the next step's 600 units of that row then differ by ~1e-5 and flip a hundred times as often -- the
    two runs decorrelate at the bf16 level within tens of steps.  Measured on MI355X (round 4), per-vector rel-L2 vs the
    rounded restatement, mean / max:  T = 2: 5.1e-5 / 3.8e-4;  T = 6: 1.4e-4 / 1.0e-3;  T = 24: 4.1e-4 / 1.8e-3;  T = 400:
    6.7e-4 / 2.9e-3 (B = 16 / 64 at T = 150: 6.5e-4, 6.4e-4) -- against 2.4e-3 / 8.3e-3 vs the fp32 oracle at T = 400: the
    restatement explains ~3/4 of the mode's distance from fp32, and all of it to 5e-5 before the flips have fed back.
    Bounds: T <= 2: mean <= 1e-4, max <= 8e-4;  T <= 6: mean <= 2.5e-4, max <= 2e-3;  longer: mean <= 1.5e-3, max <= 6e-3;
    and always at least 2.5x closer (mean) to the restatement than to the fp32 oracle."""
    monkeypatch.setenv("ONSSEN_PRECISION", "bf16")
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    cfg = dict(F=129, H=600, L=2, D=20, C=2, seed=21, gain=1.0)
    m, sd = build("deep_clustering", cfg, dev)
    x = logmag_input(5, B, T)
    with torch.no_grad():
        emb, = m([torch.from_numpy(x).to(dev)])
    emb = emb.cpu().numpy()
    ref32 = O.deep_clustering_forward(sd, x)
    # B > 16: the first layer's projection runs inside the recurrence launch and its 129th input column stays fp32
    refbf = O.deep_clustering_forward_rounded(sd, x, exact_tail0=B > 16)
    band, pin = rel_l2(emb, ref32), rel_l2(emb, refbf)
    print(f"[bf16 B={B} T={T}] vs fp32 oracle: rel-L2 mean {band.mean():.3e} max {band.max():.3e}; "
          f"vs bf16-rounded oracle: mean {pin.mean():.3e} max {pin.max():.3e}")
    assert 1e-4 < band.max() < 1.5e-2
    lim = (1e-4, 8e-4) if T <= 2 else (2.5e-4, 2e-3) if T <= 6 else (1.5e-3, 6e-3)
    assert pin.mean() < lim[0] and pin.max() < lim[1] and pin.mean() * 2.5 < band.mean()
    np.testing.assert_allclose(np.linalg.norm(emb, axis=-1), 1.0, atol=1e-5)


# ---------------------------------------------------------------- oracle at other shapes / edge cases
@pytest.mark.parametrize("B,T,H,L", [(1, 400, 600, 2), (5, 37, 600, 2), (33, 21, 300, 3), (17, 1, 64, 2), (2, 50, 30, 1),
                                     (70, 9, 128, 2)])
def test_dc_matches_oracle_ragged_shapes(dev, prec, B, T, H, L):
    cfg = dict(F=129, H=H, L=L, D=20, C=2, seed=3, gain=1.5)
    m, sd = build("deep_clustering", cfg, dev)
    x = logmag_input(11, B, max(T, 3))[:, :T]
    ref = TC.deep_clustering_forward(sd, x).numpy()
    with torch.no_grad():
        emb, = m([torch.from_numpy(x).to(dev)])
    emb = emb.cpu().numpy()
    check(f"dc oracle B{B} T{T} H{H} L{L} gain 1.5 [{prec['name']}]", emb, ref, prec["atol"])
    assert rel_l2(emb, ref).max() < 1e-4
    np.testing.assert_allclose(np.linalg.norm(emb, axis=-1), 1.0, atol=1e-5)   # unit embeddings


@pytest.mark.parametrize("ug", [4, 8, 12, 20])
def test_unit_group_variants_agree(dev, monkeypatch, prec, ug):
    monkeypatch.setenv("ONSSEN_UG", str(ug))
    cfg = dict(F=129, H=120, L=2, D=20, C=2, seed=8, gain=1.5)
    m, sd = build("chimera", cfg, dev)
    x = logmag_input(12, 18, 30)
    ref = TC.chimera_forward(sd, x)
    with torch.no_grad():
        outs = m([torch.from_numpy(x).to(dev)])
    for k, (o, r) in enumerate(zip(outs, ref)):
        check(f"chimera H120 ug{ug} gain 1.5 [{prec['name']}] output {k}", o.cpu().numpy(), r.numpy(), prec["atol"], vectors=k == 0)


@pytest.mark.parametrize("xcd", ["1", "0"])
def test_deterministic_and_graph_replay(dev, monkeypatch, xcd):
    monkeypatch.setenv("ONSSEN_XCD", xcd)
    cfg = dict(F=129, H=64, L=2, D=20, C=2, seed=2, gain=1.0)
    m, _ = build("deep_clustering", cfg, dev)
    x = torch.from_numpy(logmag_input(13, 4, 40)).to(dev)
    with torch.no_grad():
        a = m([x])[0].clone()
        b = m([x])[0].clone()
        assert torch.equal(a, b)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m([x])
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m([x])[0]
        out.zero_()
        for _ in range(3):
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, a)


def test_weight_update_repacks(dev):
    cfg = dict(F=129, H=16, L=1, D=20, C=2, seed=2, gain=1.0)
    m, sd = build("deep_clustering", cfg, dev)
    x = torch.from_numpy(logmag_input(14, 2, 10)).to(dev)
    with torch.no_grad():
        a = m([x])[0].clone()
        m.rnn.weight_hh_l0.mul_(0.5)
        b = m([x])[0]
    assert not torch.equal(a, b)
    sd["rnn.weight_hh_l0"] = sd["rnn.weight_hh_l0"] * 0.5
    np.testing.assert_allclose(b.cpu().numpy(), TC.deep_clustering_forward(sd, x.cpu().numpy()).numpy(), atol=5e-5)


def test_cpu_tensor_fails_loudly(dev):
    m, _ = build("deep_clustering", dict(F=129, H=8, L=1, D=20, C=2, seed=1, gain=1.0), dev)
    with pytest.raises(RuntimeError, match="no CPU fallback"), torch.no_grad():
        m([torch.zeros(1, 4, 129)])
    with pytest.raises(AssertionError, match="one tensor"), torch.no_grad():
        m([torch.zeros(1, 4, 129), torch.zeros(1)])


# ---------------------------------------------------------------- front end / back end
@pytest.mark.parametrize("n_fft,hop,n,B", [(256, 64, 25536, 4), (512, 128, 16000, 3), (256, 64, 129, 1), (1024, 256, 9000, 2)])
def test_stft_logmag(dev, n_fft, hop, n, B):
    from onssen_amd.features import stft_logmag
    if n < 1000:   # shortest legal signal (reflect padding needs n > n_fft/2)
        wav = np.random.default_rng(n).uniform(-0.5, 0.5, (B, n)).astype(np.float32)
    else:
        wav = np.stack([synth_mixture(30 + b, n) for b in range(B)])
    lm, ri = stft_logmag(torch.from_numpy(wav).to(dev), n_fft, hop)
    lm, ri = lm.cpu().numpy(), ri.cpu().numpy()
    for b in range(B):
        X = O.stft(wav[b], n_fft, hop)
        assert lm[b].shape == X.shape
        got = ri[b, ..., 0] + 1j * ri[b, ..., 1]
        assert np.abs(got - X).max() <= 2e-7 * np.abs(X).max() + 1e-9   # fp64 butterflies, complex64 rounding
        np.testing.assert_allclose(10.0 ** lm[b].astype(np.float64), np.abs(X).astype(np.float64) + 1e-7,
                                   rtol=2e-6, atol=1e-9)
        big = np.abs(X) > 1e-3
        np.testing.assert_allclose(lm[b][big], O.log_magnitude(X)[big], atol=2e-6)


@pytest.mark.parametrize("n_fft,hop,n,length", [(256, 64, 25536, 25536), (256, 64, 5000, 5400), (512, 128, 16000, 15000),
                                                (256, 100, 4000, 4000)])
def test_mask_istft_and_roundtrip(dev, n_fft, hop, n, length):
    from onssen_amd.features import mask_istft, stft_logmag
    B = 2
    wav = np.stack([synth_mixture(50 + b, n) for b in range(B)])
    wd = torch.from_numpy(wav).to(dev)
    _, ri = stft_logmag(wd, n_fft, hop)
    T, F = ri.shape[1], ri.shape[2]
    rng = np.random.default_rng(0)
    m0 = rng.random((B, T, F)).astype(np.float32)
    masks = torch.from_numpy(np.stack([m0, 1 - m0], -1)).to(dev)
    out = mask_istft(ri, masks, hop, length).cpu().numpy()
    for b in range(B):
        X = O.stft(wav[b], n_fft, hop)
        ref = O.mask_istft(X, np.stack([m0[b], 1 - m0[b]]), hop, length)
        np.testing.assert_allclose(out[b], ref, atol=2e-6)
        # ... and within float32 rounding of the variant that overlap-adds in float32 exactly as librosa 0.7 / 0.8 does
        if length <= n:
            ref32 = np.stack([O.istft_f32_ola(X * mm, hop, length) for mm in (m0[b], 1 - m0[b])])
            inner = slice(n_fft, min(n, length) - n_fft)
            scale = np.abs(ref).max()
            assert np.abs(out[b][:, inner] - ref32[:, inner]).max() <= 6 * np.finfo(np.float32).eps * scale
            assert np.abs(out[b][:, inner] - ref[:, inner]).max() <= 3 * np.finfo(np.float32).eps * scale
    # size-independent properties: masks summing to one split the mixture; stft->istft is the identity
    # (over the signal's own support: the last reflected half-window divides by a vanishing window sum)
    y = mask_istft(ri, None, hop, length).cpu().numpy()[:, 0]
    k = min(n, length)
    np.testing.assert_allclose(out.sum(1)[:, :k], y[:, :k], atol=2e-6)
    np.testing.assert_allclose(y[:, :k], wav[:, :k], atol=2e-6)


@pytest.mark.parametrize("B,T", [(32, 400), (12, 400), (1, 1000), (5, 401)])
def test_front_and_back_end_at_every_launch_shape(dev, B, T):
    """The host picks pairs-of-frames per wave (STFT) and frames per workgroup (iSTFT: 8 / 12 / 16) from the grid size: the
    headline batch, a medium one, one long utterance and an odd frame count all give the oracle's numbers."""
    from onssen_amd.features import mask_istft, stft_logmag
    n = (T - 1) * 64
    wav = np.stack([synth_mixture(70 + b, n) for b in range(B)])
    lm, ri = stft_logmag(torch.from_numpy(wav).to(dev), 256, 64)
    assert lm.shape == (B, T, 129)
    rng = np.random.default_rng(B)
    m0 = rng.random((B, T, 129)).astype(np.float32)
    out = mask_istft(ri, torch.from_numpy(np.stack([m0, 1 - m0], -1)).to(dev), 64, n).cpu().numpy()
    lm, ri = lm.cpu().numpy(), ri.cpu().numpy()
    for b in sorted({0, B // 2, B - 1}):
        X = O.stft(wav[b], 256, 64)
        got = ri[b, ..., 0] + 1j * ri[b, ..., 1]
        assert np.abs(got - X).max() <= 2e-7 * np.abs(X).max() + 1e-9
        big = np.abs(X) > 1e-3
        np.testing.assert_allclose(lm[b][big], O.log_magnitude(X)[big], atol=2e-6)
        np.testing.assert_allclose(out[b], O.mask_istft(X, np.stack([m0[b], 1 - m0[b]]), 64, n), atol=2e-6)


def cfg5_inputs(seed, B, n_samples=16000, n_fft=512, hop=128):
    """The input recipe of tools/gen_golden_cfg5.py (seeded synthetic 16 kHz mixtures -> log-magnitude, (Re, Im))."""
    mags, phs = [], []
    for b in range(B):
        X = O.stft(synth_mixture(seed * 100 + b, n_samples=n_samples, sr=16000), n_fft, hop)
        mags.append(O.log_magnitude(X))
        phs.append(O.phase_re_im(X))
    return np.stack(mags).astype(np.float32), np.stack(phs).astype(np.float32)


def test_cfg5_phase_net_full_shape_golden(dev, golden_dir, prec):
    """BASELINE config 5 at its full shape: phase_net with F = 257 (STFT 512/128 at 16 kHz), H = 600,
    L = 4, T = 126 one-second chunks -- the first layer of the phase BLSTM has K = 3F = 771, its head is the residual +
    2-wide L2-norm -- against the reference's strided subsample and per-frame checksums (tools/gen_golden_cfg5.py)."""
    z = np.load(f"{golden_dir}/g2_cfg5_phase_L4.npz")
    m, _ = build("phase_net", z, dev)
    x_mag, x_phase = cfg5_inputs(int(z["x_seed"]), int(z["B"]))
    with torch.no_grad():
        emb, ma, mb, pa, pb = [o.cpu().numpy() for o in m([torch.from_numpy(x_mag).to(dev), torch.from_numpy(x_phase).to(dev)])]
    assert emb.shape == (2, 126, 257, 20) and pa.shape == (2, 126, 257, 2)
    e_sub = emb[:, ::9, ::16, :]
    print(f"[{prec['name']}] cfg5: emb max abs err {np.abs(e_sub - z['emb_sub']).max():.3e}, rel-L2 {rel_l2(e_sub, z['emb_sub']).max():.3e}; "
          f"phase max abs err {max(np.abs(pa[:, ::5, ::4] - z['phase_A_sub']).max(), np.abs(pb[:, ::5, ::4] - z['phase_B_sub']).max()):.3e}")
    np.testing.assert_allclose(e_sub, z["emb_sub"], atol=1e-5, rtol=1e-4)
    assert rel_l2(e_sub, z["emb_sub"]).max() < 1e-4
    np.testing.assert_allclose(emb.astype(np.float64).sum(axis=(2, 3)), z["emb_sum_per_frame"], atol=5e-3)
    np.testing.assert_allclose(ma[:, ::5, :], z["mask_A_sub"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(mb[:, ::5, :], z["mask_B_sub"], atol=1e-5, rtol=1e-4)
    # the phase outputs normalise a 2-vector (re, im) + residual that can be short: round-off is amplified there
    # (the oracle itself needs 3e-5 on the tiny fixture); unit norm is checked exactly
    ptol = 2e-4 if not prec["name"].startswith("f32") else 1e-4
    np.testing.assert_allclose(pa[:, ::5, ::4, :], z["phase_A_sub"], atol=ptol)
    np.testing.assert_allclose(pb[:, ::5, ::4, :], z["phase_B_sub"], atol=ptol)
    np.testing.assert_allclose(np.linalg.norm(pa, axis=-1), 1.0, atol=1e-5)
    np.testing.assert_allclose(pa.astype(np.float64).sum(axis=(2, 3)), z["phase_A_sum_per_frame"], atol=2e-2)
    np.testing.assert_allclose(pb.astype(np.float64).sum(axis=(2, 3)), z["phase_B_sum_per_frame"], atol=2e-2)


@pytest.mark.parametrize("T", [80, 400])
def test_cfg3_chimera_batch_64_matches_aten_oracle(dev, monkeypatch, T):
    """BASELINE config 3 at its bench batch (T = 400 = the bench's chunk length: VERDICT r2 weak 1):
    chimera++ L = 4, H = 600 with B = 64 -- 16-row exchange groups, the first layer's projection fused into the recurrence
    launch, two heads on the x3 image -- against the ATen-on-CPU oracle (itself pinned to the reference by the B = 1
    fixture g2_cfg3_chimera_L4)."""
    monkeypatch.setenv("ONSSEN_PRECISION", "bf16x3")
    monkeypatch.setenv("ONSSEN_XCD", "1")
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    B = 64
    m, sd = build("chimera", dict(F=129, H=600, L=4, D=20, C=2, seed=0, gain=1.0), dev)
    x = logmag_input(13, B, T)
    ref = [r.numpy() for r in TC.chimera_forward(sd, x)]
    with torch.no_grad():
        outs = [o.cpu().numpy() for o in m([torch.from_numpy(x).to(dev)])]
    print(f"cfg3 B=64: emb max abs err {np.abs(outs[0] - ref[0]).max():.3e}, rel-L2 {rel_l2(outs[0], ref[0]).max():.3e}")
    np.testing.assert_allclose(outs[0], ref[0], atol=1e-5, rtol=1e-4)
    assert rel_l2(outs[0], ref[0]).max() < 1e-4
    np.testing.assert_allclose(outs[1], ref[1], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(outs[2], ref[2], atol=1e-5, rtol=1e-4)


def test_feature_kernels_match_reference_fixture(dev, golden_dir):
    """the label kernel (one_hot / |S| / cos) against the REFERENCE's get_one_hot / get_cos_difference
    outputs (tests/golden/g3_features.npz, tools/gen_golden_features.py) on the fixture's STFTs."""
    from onssen_amd.features import training_labels
    z = np.load(f"{golden_dir}/g3_features.npz")
    for tag in ("a", "b"):
        X, S1, S2 = z[f"{tag}_X"], z[f"{tag}_S1"], z[f"{tag}_S2"]
        ri = lambda c: torch.from_numpy(np.stack([c.real, c.imag], -1).astype(np.float32))[None].to(dev)
        feat = torch.from_numpy(z[f"{tag}_log_magnitude"])[None].to(dev)
        for db in (40, 20):
            oh, mm, m1, m2, c1, c2 = training_labels(ri(X), ri(S1), ri(S2), feat, float(db), with_cos=True)
            ref = z[f"{tag}_one_hot_{db}"]
            # argmax ties (|S1| == |S2| to the last bit, e.g. both silent) are the only bins allowed to differ
            diff = (oh[0].cpu().numpy() != ref).any(-1)
            assert np.all(np.abs(np.abs(S1) - np.abs(S2))[diff] <= 1e-6 * (np.abs(S1) + np.abs(S2))[diff] + 1e-12), diff.sum()
            assert diff.mean() < 1e-3
        np.testing.assert_allclose(mm[0].cpu().numpy(), np.abs(X), rtol=2e-6, atol=1e-8)
        big = (np.abs(X) > 1e-3) & (np.abs(S1) > 1e-3) & (np.abs(S2) > 1e-3)
        np.testing.assert_allclose(c1[0].cpu().numpy()[big], z[f"{tag}_cos_s1"][big], atol=1e-4)
        np.testing.assert_allclose(c2[0].cpu().numpy()[big], z[f"{tag}_cos_s2"][big], atol=1e-4)


def test_end_to_end_separation_dc_matches_oracle(dev, monkeypatch):
    """deep-clustering separation end to end -- separate_dc(host_kmeans=True), i.e. upstream's
    sklearn KMeans(2, random_state=0) on the active bins -- against the oracle's mask-apply + iSTFT fed with the SAME
    labels computed here from the oracle's embedding (egs/wsj0-2mix/deep_clustering/evaluate.py:31-45)."""
    from sklearn.cluster import KMeans
    from onssen_amd.separation import separate_dc
    monkeypatch.setenv("ONSSEN_PRECISION", "bf16x3")
    m, sd = build("deep_clustering", dict(F=129, H=48, L=2, D=20, C=2, seed=8, gain=1.5), dev)
    n = 6400
    wav = np.stack([synth_mixture(75 + b, n) for b in range(2)])
    sig = separate_dc(m, torch.from_numpy(wav).to(dev), host_kmeans=True).cpu().numpy()
    for b in range(2):
        X = O.stft(wav[b], 256, 64)
        feat = O.log_magnitude(X)
        emb = O.deep_clustering_forward(sd, feat[None])[0]
        act = O.dc_active_bins(feat)
        lab = KMeans(n_clusters=2, random_state=0, n_init=10).fit_predict(emb[act])
        masks = O.dc_binary_masks(feat, lab)                       # (2, T, F): label 1 -> speaker 0 like evaluate.py:38-41
        ref = O.mask_istft(X, masks, 64, n)
        # a bin whose embedding sits between the clusters may flip with the 1e-5 embedding difference: compare per speaker
        # with the better of the two label orders and allow a few flipped bins' worth of energy
        err = min(np.abs(sig[b] - ref).max(), np.abs(sig[b] - ref[::-1]).max())
        assert err < 5e-3, err
        best = ref if np.abs(sig[b] - ref).max() <= np.abs(sig[b] - ref[::-1]).max() else ref[::-1]
        assert np.mean(np.abs(sig[b] - best) > 5e-5) < 0.02


def test_end_to_end_separation_chimera(dev, prec):
    from onssen_amd.separation import separate_chimera
    m, sd = build("chimera", dict(F=129, H=48, L=2, D=20, C=2, seed=4, gain=1.5), dev)
    wav = np.stack([synth_mixture(70 + b, 6400) for b in range(2)])
    sig = separate_chimera(m, torch.from_numpy(wav).to(dev)).cpu().numpy()
    for b in range(2):
        X = O.stft(wav[b], 256, 64)
        _, a, bb = O.chimera_forward(sd, O.log_magnitude(X)[None])
        ref = O.mask_istft(X, np.stack([a[0], bb[0]]), 64, 6400)
        np.testing.assert_allclose(sig[b], ref, atol=prec["atol"])


def test_label_features_and_synthetic_loader(dev):
    """Rows N3 / H1: label kernels vs the oracle restatement of get_one_hot / get_cos_difference, and the
    synthetic wsj0-2mix loader's yield contract feeding loss_dc."""
    from onssen_amd.data import wsj0_2mix_dataloader
    from onssen_amd.features import stft_logmag, training_labels
    from onssen_amd.loss import loss_dc
    B, n = 3, 9000
    trips = [synth_mixture(80 + b, n, return_sources=True) for b in range(B)]
    wav = torch.from_numpy(np.stack([np.stack(t) for t in trips])).to(dev)
    lm, ri = stft_logmag(wav.view(3 * B, -1))
    T, F = lm.shape[1:]
    lm, ri = lm.view(B, 3, T, F), ri.view(B, 3, T, F, 2)
    oh, mm, m1, m2, c1, c2 = training_labels(ri[:, 0], ri[:, 1], ri[:, 2], lm[:, 0], 40.0, with_cos=True)
    for b in range(B):
        X, S1, S2 = (O.stft(w, 256, 64) for w in trips[b])
        ref = O.one_hot_labels(lm[b, 0].cpu().numpy(), np.abs(S1), np.abs(S2), 40.0)
        assert (oh[b].cpu().numpy() != ref).mean() < 1e-4            # ties / threshold edges at fp32 rounding only
        np.testing.assert_allclose(m1[b].cpu().numpy(), np.abs(S1), rtol=2e-6, atol=1e-8)
        big = (np.abs(X) > 1e-3) & (np.abs(S1) > 1e-3)
        np.testing.assert_allclose(c1[b].cpu().numpy()[big], O.cos_difference(X, S1)[big], atol=1e-4)
    fo = dict(data_path="", batch_size=4, frame_length=100, sampling_rate=8000, window_size=256, hop_size=64, db_threshold=40)
    for name, n_in, n_lab in (("dc", 1, 2), ("chimera", 1, 4), ("chimera++", 1, 6), ("phase", 2, 6)):
        loader = wsj0_2mix_dataloader(name, fo, "tr", "cuda:0")
        inp, lab = next(iter(loader))
        assert len(inp) == n_in and len(lab) == n_lab
        assert inp[0].shape == (4, 100, 129) and lab[0].shape == (4, 100, 129, 2) and lab[1].shape == (4, 100, 129)
    inp, lab = next(iter(wsj0_2mix_dataloader("dc", fo, "cv", "cuda:0")))
    m, _ = build("deep_clustering", dict(F=129, H=16, L=1, D=20, C=2, seed=1, gain=1.0), dev)
    with torch.no_grad():
        loss = loss_dc(m(inp), lab)
    assert loss.shape == (4, 4) and torch.isfinite(loss).all()


def test_dc_cluster_agrees_with_sklearn_and_separates(dev):
    """Row N2: device 2-means vs sklearn KMeans(n_clusters=2, random_state=0) on the same active bins
    (permutation-invariant agreement), then the all-GPU separate_dc against the host-k-means variant."""
    from sklearn.cluster import KMeans
    from onssen_amd.separation import dc_masks, separate_dc
    rng = np.random.default_rng(5)
    B, T, F, D = 2, 400, 129, 20
    cents = rng.standard_normal((B, 2, D)); cents /= np.linalg.norm(cents, axis=-1, keepdims=True)
    lab = rng.integers(0, 2, (B, T, F))
    e = np.take_along_axis(cents[:, None, None], lab[..., None, None], axis=3)[..., 0, :] + 0.2 * rng.standard_normal((B, T, F, D))
    e = (e / np.linalg.norm(e, axis=-1, keepdims=True)).astype(np.float32)
    feat = rng.uniform(-3.0, 1.0, (B, T, F)).astype(np.float32)
    masks = dc_masks(torch.from_numpy(e).to(dev), torch.from_numpy(feat).to(dev)).cpu().numpy()
    for b in range(B):
        act = O.dc_active_bins(feat[b])
        sk = KMeans(n_clusters=2, random_state=0, n_init=10).fit_predict(e[b][act])
        agree = (masks[b][act][:, 0] == sk).mean()
        assert max(agree, 1 - agree) > 0.999 and np.all(masks[b][~act] == 0)
    m, _ = build("deep_clustering", dict(F=129, H=32, L=1, D=20, C=2, seed=6, gain=1.0), dev)
    wav = torch.from_numpy(np.stack([synth_mixture(90 + b, 8000) for b in range(2)])).to(dev)
    a = separate_dc(m, wav)
    assert a.shape == (2, 2, 8000) and torch.isfinite(a).all()
    assert torch.equal(a, separate_dc(m, wav))                       # deterministic


@pytest.mark.parametrize("B,T,H,L", [(32, 400, 600, 2), (5, 37, 600, 2), (33, 21, 300, 3), (70, 9, 128, 2), (16, 50, 64, 1)])
@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_xcd_local_persistent_recurrence(dev, monkeypatch, B, T, H, L, precision):
    """ONSSEN_BLSTM_XCD: one persistent launch per layer with the h exchange inside one XCD's L2 (or the
    placement-independent protocol if the kernel finds a group spread over XCDs); split-bf16 and exact-fp32 forms."""
    from onssen_amd.nn import _core
    monkeypatch.setenv("ONSSEN_XCD", "1")
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    monkeypatch.setenv("ONSSEN_PRECISION", precision)
    cfg = dict(F=129, H=H, L=L, D=20, C=2, seed=3, gain=1.0)
    m, sd = build("deep_clustering", cfg, dev)
    x = logmag_input(11, B, max(T, 3))[:, :T]
    ref = TC.deep_clustering_forward(sd, x).numpy()
    with torch.no_grad():
        emb = m([torch.from_numpy(x).to(dev)])[0].cpu().numpy()
        emb2 = m([torch.from_numpy(x).to(dev)])[0].cpu().numpy()
    print(f"xcd-local B={B} T={T} H={H}: max abs err {np.abs(emb - ref).max():.3e}, placement-independent protocol used: "
          f"{_core._XcdStatus.safe_protocol_seen}")
    np.testing.assert_allclose(emb, ref, atol=5e-5 if precision == "bf16x3" else 1e-5, rtol=1e-4)
    assert rel_l2(emb, ref).max() < 1e-4
    np.testing.assert_array_equal(emb, emb2)
    assert _core._XcdPolicy.persistent_launches > 0


@pytest.mark.parametrize("B,T,H,L,frames", [(32, 400, 768, 2, None), (5, 37, 700, 2, None), (17, 50, 768, 3, None),
                                            (6, 40, 768, 2, [40, 3, 17, 40, 1, 29])])
def test_wide_layers_on_the_persistent_recurrence(dev, monkeypatch, B, T, H, L, frames):
    """Round 4 (VERDICT r3 missing #2): 640 < H <= 768 runs the XCD-local persistent recurrence in split-bf16 -- 32 members of
    24 units, every CU of an XCD -- instead of the launch-per-step form; parity with the oracle as for H = 600, ragged rows
    bit-identical to their batch-1 runs, and the per-step time printed (bound in the test: well under the launch-per-step
    form's ~10 us/step)."""
    from onssen_amd.nn import _core
    monkeypatch.setenv("ONSSEN_XCD", "1")
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    monkeypatch.setenv("ONSSEN_PRECISION", "bf16x3")
    cfg = dict(F=129, H=H, L=L, D=20, C=2, seed=3, gain=1.0)
    m, sd = build("deep_clustering", cfg, dev)
    x = logmag_input(11, B, T)
    xd = torch.from_numpy(x).to(dev)
    n_p = _core._XcdPolicy.persistent_launches
    if frames is not None:
        fr = torch.tensor(frames, dtype=torch.int32, device=dev)
        with torch.no_grad():
            emb = m([xd], frames=fr)[0].cpu().numpy().reshape(B, T, 129, 20)
            for b, n in enumerate(frames):
                one = m([xd[b:b + 1, :n].contiguous()])[0].cpu().numpy().reshape(n, 129, 20)
                np.testing.assert_array_equal(emb[b, :n], one)
                ref = TC.deep_clustering_forward(sd, x[b:b + 1, :n]).numpy().reshape(n, 129, 20)
                np.testing.assert_allclose(one, ref, atol=5e-5, rtol=1e-4)
        assert _core._XcdPolicy.persistent_launches > n_p
        return
    ref = TC.deep_clustering_forward(sd, x).numpy()
    with torch.no_grad():
        emb = m([xd])[0].cpu().numpy()
        emb2 = m([xd])[0].cpu().numpy()
    np.testing.assert_allclose(emb, ref, atol=5e-5, rtol=1e-4)
    assert rel_l2(emb, ref).max() < 1e-4
    np.testing.assert_array_equal(emb, emb2)
    assert _core._XcdPolicy.persistent_launches > n_p
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        e0.record()
        for _ in range(5):
            m([xd])
        e1.record()
    torch.cuda.synchronize()
    per_step = e0.elapsed_time(e1) * 1e3 / (5 * L * T)     # whole forward (projections and head included) per layer-step
    print(f"H={H} B={B} T={T} L={L}: max abs err {np.abs(emb - ref).max():.3e}; whole forward {per_step:.2f} us per layer-step")
    if T >= 400:
        assert per_step < 5.0


@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
@pytest.mark.parametrize("B,T,H,L", [(64, 60, 600, 2), (33, 21, 300, 3)])
def test_xcd_placement_independent_protocol(dev, monkeypatch, B, T, H, L, precision):
    """Test bit 8 of the debug flags rotates the exchange groups across the XCDs: the kernel must notice (XCC ids
    disagree), switch to the write-through / agent-fence protocol, and still match the oracle."""
    from onssen_amd.nn import _core
    monkeypatch.setenv("ONSSEN_XCD", "1")
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    monkeypatch.setenv("ONSSEN_PRECISION", precision)
    monkeypatch.setenv("ONSSEN_ABLATE", "8")
    _core._XcdStatus.safe_protocol_seen = False
    cfg = dict(F=129, H=H, L=L, D=20, C=2, seed=3, gain=1.0)
    m, sd = build("deep_clustering", cfg, dev)
    x = logmag_input(12, B, T)
    ref = TC.deep_clustering_forward(sd, x).numpy()
    with torch.no_grad():
        emb = m([torch.from_numpy(x).to(dev)])[0].cpu().numpy()
        emb2 = m([torch.from_numpy(x).to(dev)])[0].cpu().numpy()
    assert _core._XcdStatus.safe_protocol_seen
    np.testing.assert_allclose(emb, ref, atol=5e-5, rtol=1e-4)
    np.testing.assert_array_equal(emb, emb2)


@pytest.mark.gpu
def test_train_mode_batch_norm_and_normalisation_on_device(dev, monkeypatch):
    """Training-forward glue of deep_clustering (onssen/nn/deep_clustering.py:36-41): BatchNorm1d over the frames and F.normalize
    of the embedding on the HIP kernels (batch_norm_rows, l2_normalize) against nn.BatchNorm1d on the permuted (B, C, T) view and
    F.normalize under float64 autograd: outputs, running statistics after two batches, all gradients."""
    from onssen_amd.nn._train import batch_norm_rows, l2_normalize
    monkeypatch.setenv("ONSSEN_TRAIN_HIP", "1")
    torch.manual_seed(2)
    B, T, C, D = 4, 50, 1200, 20
    bn = torch.nn.BatchNorm1d(C).to(dev).train()
    ref = torch.nn.BatchNorm1d(C).double().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_()
        ref.weight.copy_(bn.weight.cpu().double()); ref.bias.copy_(bn.bias.cpu().double())
    for it in range(2):
        x = torch.randn(B, T, C) * 0.5 + 0.1
        g = torch.randn(B, T, C)
        xg = x.to(dev).requires_grad_(True)
        y = batch_norm_rows(bn, xg)
        (y * g.to(dev)).sum().backward()
        xr = x.double().requires_grad_(True)
        yr = ref(xr.permute(0, 2, 1)).permute(0, 2, 1)
        (yr * g.double()).sum().backward()
        np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), rtol=2e-5, atol=5e-6)
        assert (xg.grad.cpu().double() - xr.grad).abs().max() <= 3e-5 * xr.grad.abs().max()
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), ref.running_mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), ref.running_var.numpy(), rtol=2e-5)
    assert int(bn.num_batches_tracked) == 2
    np.testing.assert_allclose(bn.weight.grad.cpu().numpy(), ref.weight.grad.numpy(), rtol=5e-5, atol=5e-4)
    np.testing.assert_allclose(bn.bias.grad.cpu().numpy(), ref.bias.grad.numpy(), rtol=5e-5, atol=5e-4)
    # normalisation of (B, T*F, D) rows
    e = torch.randn(B, T * 129, D); e[0, 5] = 0.0
    ge = torch.randn(B, T * 129, D)
    eg = e.to(dev).requires_grad_(True)
    n = l2_normalize(eg)
    (n * ge.to(dev)).sum().backward()
    er = e.double().requires_grad_(True)
    nr = torch.nn.functional.normalize(er, p=2, dim=-1)
    (nr * ge.double()).sum().backward()
    np.testing.assert_allclose(n.detach().cpu().numpy(), nr.detach().numpy(), rtol=2e-6, atol=1e-7)
    ok = torch.ones(B, T * 129, dtype=torch.bool); ok[0, 5] = False
    assert (eg.grad.cpu().double() - er.grad)[ok].abs().max() <= 2e-5 * er.grad[ok].abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize("hip", ["1", "0"])
def test_loss_dc_with_gradient_on_device(dev, golden_dir, monkeypatch, hip):
    """N1: loss_dc WITH a gradient on the device -- onssen_loss_dc_f32 + onssen_loss_dc_grad_f32 behind autograd (ONSSEN_LOSS_HIP=0:
    the PyTorch Gram form) -- (a) against the reference's own loss / gradient fixture (G4: value, (B,B) shape quirk, gradient norm
    over all parameters, fc_dc bias gradient) and (b) at the training shape against float64 autograd through the literal
    three-product form of onssen/loss/loss_dc.py:36-44."""
    from onssen_amd import nn as onn
    from onssen_amd.loss import loss_dc
    monkeypatch.setenv("ONSSEN_LOSS_HIP", hip)
    monkeypatch.setenv("ONSSEN_TRAIN_HIP", "1")          # eval-mode network with a graph: the HIP training path (MIOpen's RNN refuses)
    z = np.load(f"{golden_dir}/g4_loss_dc.npz")
    sd = make_state_dict("deep_clustering", 129, int(z["H"]), int(z["L"]), 20, 2, seed=int(z["seed"]))
    m = onn.deep_clustering(129, int(z["H"]), int(z["L"]), 20, dropout=0.0)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.to(dev).eval()
    out = m([torch.from_numpy(z["x"]).to(dev)])
    assert out[0].requires_grad
    loss = loss_dc(out, [torch.from_numpy(z["one_hot"]).to(dev), torch.from_numpy(z["mag"]).to(dev)])
    assert tuple(loss.shape) == (3, 3)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), z["loss"], rtol=2e-5)
    torch.mean(loss).backward()
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters()))
    np.testing.assert_allclose(gn.item(), float(z["grad_norm"]), rtol=2e-4)
    gb, rb = m.fc_dc.bias.grad.cpu().numpy(), z["grad_fc_dc_bias"]      # fp32 sums over 3 x T rows: bound by the largest entry
    assert np.abs(gb - rb).max() <= 5e-4 * np.abs(rb).max(), (np.abs(gb - rb).max(), np.abs(rb).max())
    # (b) 4 x 400 x 129 bins, D = 20
    torch.manual_seed(3)
    B, T, F, D, C = 4, 400, 129, 20, 2
    emb = torch.nn.functional.normalize(torch.randn(B, T, F, D), dim=-1)
    one_hot = torch.nn.functional.one_hot(torch.randint(0, C + 1, (B, T, F)), C + 1)[..., :C].float()
    mag = torch.rand(B, T, F) + 0.01
    e64 = emb.double().requires_grad_(True)
    V = e64.reshape(B, T * F, D); Y = one_hot.double().reshape(B, T * F, C); mg = mag.double().reshape(B, -1)
    tot = mg.sum(1, keepdim=True); w = torch.sqrt(mg / tot).unsqueeze(-1)
    Vm, Ym = V * Y.sum(2, keepdim=True) * w, Y * w
    fro = lambda x: torch.sqrt((x * x).flatten(1).sum(1))
    ref = (fro(Vm.transpose(1, 2) @ Vm) - 2 * fro(Vm.transpose(1, 2) @ Ym) + fro(Ym.transpose(1, 2) @ Ym)) * tot
    g_ref, = torch.autograd.grad(ref.mean(), e64)
    eg = emb.to(dev).requires_grad_(True)
    got = loss_dc([eg], [one_hot.double().to(dev), mag.to(dev)])
    g_got, = torch.autograd.grad(got.mean(), eg)
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=5e-5)
    assert (g_got.cpu().double() - g_ref).abs().max() <= 5e-5 * g_ref.abs().max()


def test_loss_dc_value_on_device_matches_reference_fixture(dev, golden_dir, monkeypatch):
    """N1 (forward): validation-style loss -- HIP forward + onssen_loss_dc_f32 under no_grad -- against the value the
    reference's loss_dc produced (G4 fixture), and against the oracle on a full-size batch."""
    from onssen_amd.loss import loss_dc
    monkeypatch.setenv("ONSSEN_PRECISION", "bf16x3")
    z = np.load(f"{golden_dir}/g4_loss_dc.npz")
    cfg = dict(F=129, H=int(z["H"]), L=int(z["L"]), D=20, C=2, seed=int(z["seed"]), gain=1.0)
    m, _ = build("deep_clustering", cfg, dev)
    with torch.no_grad():
        out = m([torch.from_numpy(z["x"]).to(dev)])
        loss = loss_dc(out, [torch.from_numpy(z["one_hot"]).to(dev), torch.from_numpy(z["mag"]).to(dev)])
    assert tuple(loss.shape) == (3, 3)
    np.testing.assert_allclose(loss.cpu().numpy(), z["loss"], rtol=1e-4)
    # BASELINE-size rows: 4 x (400 x 129) bins, D = 20
    rng = np.random.default_rng(9)
    B, T, F, D = 4, 400, 129, 20
    emb = rng.standard_normal((B, T, F, D)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=-1, keepdims=True)
    lab = rng.integers(0, 3, size=(B, T, F))
    one_hot = np.stack([lab == 0, lab == 1], -1).astype(np.float64)
    mag = (np.abs(rng.standard_normal((B, T, F))) + 1e-3).astype(np.float32)
    with torch.no_grad():
        got = loss_dc([torch.from_numpy(emb).to(dev)], [torch.from_numpy(one_hot).to(dev), torch.from_numpy(mag).to(dev)])
    np.testing.assert_allclose(got.cpu().numpy(), O.loss_dc(emb, one_hot, mag), rtol=1e-4)


@pytest.mark.parametrize("name", ["g5_enhance_H16_L2", "g5_enhance_H32_L1"])
def test_enhance_golden(dev, golden_dir, prec, name):
    """N4: onssen.nn.enhance on the HIP path (BLSTM + BN-folded sigmoid mask head + the two ReLU restoration GEMMs)
    against the reference's outputs."""
    from onssen_amd import nn as onn
    z = np.load(f"{golden_dir}/{name}.npz")
    sd = make_state_dict("enhance", int(z["F"]), int(z["H"]), int(z["L"]), seed=int(z["seed"]), gain=float(z["gain"]))
    m = onn.enhance(int(z["F"]), int(z["H"]), int(z["L"]))
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to(dev).eval()
    with torch.no_grad():
        out, = m([torch.from_numpy(z["x"]).to(dev), torch.from_numpy(z["mag_noisy"]).to(dev)])
    # outputs reach ~7; the mask carries the split-bf16 error, the two small layers are exact fp32
    np.testing.assert_allclose(out.cpu().numpy(), z["out_clean"], atol=prec["atol"] * 4, rtol=1e-4)


def test_batch_sdr_on_device_matches_reference_fixture(dev, golden_dir):
    """N4: onssen_amd.evaluate.batch_SDR_torch (HIP) against the reference's own outputs, and on separated-waveform
    sized inputs against the oracle."""
    from onssen_amd.evaluate import batch_SDR_torch
    z = np.load(f"{golden_dir}/g6_batch_sdr.npz")
    for tag in ("c2", "c3m"):
        mask = torch.from_numpy(z[f"{tag}_mask"]).to(dev) if f"{tag}_mask" in z.files else None
        sdr, perm = batch_SDR_torch(torch.from_numpy(z[f"{tag}_est"]).to(dev), torch.from_numpy(z[f"{tag}_org"]).to(dev),
                                    mask, return_perm=True)
        np.testing.assert_allclose(sdr.cpu().numpy(), z[f"{tag}_sdr"], rtol=1e-4, atol=1e-4)
        np.testing.assert_array_equal(perm.cpu().numpy(), z[f"{tag}_perm"])
    rng = np.random.default_rng(3)
    org = rng.standard_normal((32, 2, 25536)).astype(np.float32)
    est = (org[:, ::-1] + 0.5 * rng.standard_normal(org.shape)).astype(np.float32)
    got = batch_SDR_torch(torch.from_numpy(est.copy()).to(dev), torch.from_numpy(org).to(dev))
    np.testing.assert_allclose(got.cpu().numpy(), O.batch_sdr(est, org)[0], rtol=1e-4, atol=1e-4)


def test_chimera_losses_on_device_match_reference_fixture(dev, golden_dir):
    """N1: loss_chimera_msa / psa under no_grad (HIP loss_dc + HIP mask term on the strided mask views) against the
    reference's values."""
    from onssen_amd.loss import loss_chimera_msa, loss_chimera_psa
    z = np.load(f"{golden_dir}/g7_loss_chimera.npz")
    d = lambda k: torch.from_numpy(z[k]).to(dev)
    masks = d("masks")
    out = [d("emb"), masks[..., 0], masks[..., 1]]
    with torch.no_grad():
        msa = loss_chimera_msa(out, [d("one_hot"), d("mag"), d("s1"), d("s2")])
        psa = loss_chimera_psa(out, [d("one_hot"), d("mag"), d("s1"), d("s2"), d("c1"), d("c2")])
    np.testing.assert_allclose(msa.cpu().numpy(), z["msa"], rtol=1e-4)
    np.testing.assert_allclose(psa.cpu().numpy(), z["psa"], rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["msa", "psa"])
@pytest.mark.parametrize("views", ["network", "separate"])
def test_chimera_losses_with_gradient_on_device(dev, golden_dir, which, views):
    """N1: loss_chimera_msa / psa WITH a gradient on the HIP kernels (loss_dc value + gradient kernels, mask
    term: onssen_loss_mask_f32 picks the assignment, onssen_loss_mask_grad_f32 the gradient) against float64 autograd through
    the literal form of onssen/loss/loss_chimera.py:18-31 / :47-59 + loss_dc.py:24-44 -- value (incl. the (B,B) quirk), the
    embedding's gradient and both masks' gradients; the masks either as the network hands them over (two strided views of one
    (B,T,F,2) tensor: the gradient goes to that tensor in one pass) or as two unrelated tensors.  The reference-generated g7
    fixture pins the value of the same functions (test above)."""
    from onssen_amd.loss import loss_chimera_msa, loss_chimera_psa
    torch.manual_seed(11)
    B, T, F, D, C = 3, 100, 129, 20, 2
    emb = torch.nn.functional.normalize(torch.randn(B, T, F, D), dim=-1)
    raw = torch.rand(B, T, F, C) * 0.98 + 0.01
    lab = torch.randint(0, C + 1, (B, T, F))
    one_hot = torch.nn.functional.one_hot(lab, C + 1)[..., :C].double()
    mag = torch.rand(B, T, F) + 0.05
    s1, s2 = mag * torch.rand(B, T, F), mag * torch.rand(B, T, F)
    c1, c2 = torch.rand(B, T, F) * 2 - 1, torch.rand(B, T, F) * 2 - 1
    raw[1] = torch.stack([s2[1] / mag[1], s1[1] / mag[1]], -1) * 0.9 + 0.02      # utterance 1: the swapped assignment wins (MSA)

    def literal(e, m):                # float64, the reference's formulas
        V = e.reshape(B, T * F, D); Y = one_hot.reshape(B, T * F, C); mg = mag.double().reshape(B, -1)
        tot = mg.sum(1, keepdim=True); w = torch.sqrt(mg / tot).unsqueeze(-1)
        Vm, Ym = V * Y.sum(2, keepdim=True) * w, Y * w
        fro = lambda x: torch.sqrt((x * x).flatten(1).sum(1))
        le = (fro(Vm.transpose(1, 2) @ Vm) - 2 * fro(Vm.transpose(1, 2) @ Ym) + fro(Ym.transpose(1, 2) @ Ym)) * tot
        x = mag.double()
        t1, t2 = s1.double(), s2.double()
        if which == "psa":
            t1, t2 = torch.min(x, torch.relu(t1 * c1.double())), torch.min(x, torch.relu(t2 * c2.double()))
        l1 = lambda a: a.reshape(B, -1).abs().sum(1)
        mA, mB = m[..., 0], m[..., 1]
        lm = torch.min(l1(mA * x - t1) + l1(mB * x - t2), l1(mB * x - t1) + l1(mA * x - t2))
        return le * 0.975 + lm * 0.025
    e64, m64 = emb.double().requires_grad_(True), raw.double().requires_grad_(True)
    ref = literal(e64, m64)
    ge_ref, gm_ref = torch.autograd.grad(ref.mean(), [e64, m64])

    eg = emb.to(dev).requires_grad_(True)
    if views == "network":
        leaf = raw.to(dev).requires_grad_(True)
        mg_ = (leaf * 1.0).reshape(B, T, F * C).reshape(B, T, F, C)        # a non-leaf buffer, sliced like chimera.py:43-45
        mA, mB = mg_[:, :, :, 0], mg_[:, :, :, 1]
        wrt = [eg, leaf]
    else:
        la, lb = raw[..., 0].contiguous().to(dev).requires_grad_(True), raw[..., 1].contiguous().to(dev).requires_grad_(True)
        mA, mB = la, lb
        wrt = [eg, la, lb]
    label = [one_hot.to(dev), mag.to(dev), s1.to(dev), s2.to(dev)] + ([c1.to(dev), c2.to(dev)] if which == "psa" else [])
    got = (loss_chimera_msa if which == "msa" else loss_chimera_psa)([eg, mA, mB], label)
    assert tuple(got.shape) == (B, B)
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=5e-5)
    grads = torch.autograd.grad(got.mean(), wrt)
    assert (grads[0].cpu().double() - ge_ref).abs().max() <= 5e-5 * ge_ref.abs().max()
    gm = grads[1].cpu().double() if views == "network" else torch.stack([grads[1].cpu().double(), grads[2].cpu().double()], -1)
    # |.|'s gradient is a sign: it may differ where the fp32 residual is within rounding of zero
    x = mag.double().unsqueeze(-1)
    far = (gm - gm_ref).abs() <= 1e-6 * gm_ref.abs().max()
    assert far.double().mean() > 0.9995 and torch.isfinite(gm).all()
    assert (gm.abs() <= (0.025 / B) * x * (1 + 1e-6) + 1e-12).all()


# ---------------------------------------------------------------- training path (row N1): HIP forward + backward
@pytest.mark.gpu
# XCD-local persistent backward launch + split-bf16 MFMA gradient GEMMs (default) | launch per step + fp32 library GEMMs
@pytest.mark.parametrize("bwd_xcd,gemm", [("1", "x3"), ("0", "blas")])
@pytest.mark.parametrize("B,T,F,H,L", [(4, 50, 129, 600, 2), (3, 17, 129, 30, 3), (18, 9, 33, 128, 1), (40, 12, 20, 64, 2)])
def test_blstm_training_gradients_match_autograd(dev, monkeypatch, B, T, F, H, L, bwd_xcd, gemm):
    """What `loss.backward()` computes for self.rnn (onssen/utils/train.py:80-84): the HIP training path (saved-state
    XCD forward, backward recurrence kernel, rocBLAS weight-gradient GEMMs) against nn.LSTM autograd in float64 on the
    CPU, dropout off.  Split-bf16 products: every gradient tensor within 3e-4 of its largest entry."""
    from onssen_amd.nn._core import BLSTMParams
    monkeypatch.setenv("ONSSEN_TRAIN_HIP", "1")
    monkeypatch.setenv("ONSSEN_BWD_XCD", bwd_xcd)
    monkeypatch.setenv("ONSSEN_TRAIN_GEMM", gemm)
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    torch.manual_seed(H + L)
    ref = torch.nn.LSTM(F, H, L, batch_first=True, bidirectional=True).double()
    rnn = BLSTMParams(F, H, L, dropout=0.0)
    rnn.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    rnn = rnn.to(dev)
    x = torch.randn(B, T, F, dtype=torch.float64)
    R = torch.randn(B, T, 2 * H, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    yr, _ = ref(xr)
    (yr * R).sum().backward()
    xg = x.float().to(dev).requires_grad_(True)
    yg = rnn.autograd_forward(xg, True)
    (yg * R.float().to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert (yg.detach().cpu().double() - yr.detach()).abs().max() < 2e-5

    bad = []

    def close(a, b, what):
        a, b = a.detach().cpu().double(), b.detach()
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        if not err <= 3e-4 * max(scale, 1e-6):
            bad.append(f"{what}: max err {err:.3e} vs max |ref| {scale:.3e}")
    close(xg.grad, xr.grad, "dx")
    for name, p in rnn.named_parameters():
        close(p.grad, getattr(ref, name).grad, name)
    assert not bad, "; ".join(bad)
    from onssen_amd.nn._core import _XcdStatus
    _XcdStatus.poll(wait=True)       # raises if a persistent launch (forward or backward) aborted


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,F,H,L,forced", [(3, 12, 40, 800, 2, False),      # H > 768: exact-fp32 launch-per-step forward + launch-per-step backward
                                              (5, 30, 40, 768, 2, False),      # 640 < H <= 768: persistent forward (24-unit members) + launch-per-step backward
                                              (18, 9, 33, 700, 1, False),
                                              (5, 20, 129, 600, 2, True),      # H <= 640 inside forced_steps(): what an aborted step is re-run on
                                              (18, 9, 33, 128, 1, True)])
def test_training_without_the_persistent_kernels_stays_on_hip(dev, monkeypatch, B, T, F, H, L, forced):
    """Round 4: wide layers and the re-run of an aborted step train on the launch-per-step HIP recurrences
    (forward with saved state: onssen_lstm_train_forward_form_f32; backward: ONSSEN_LSTM_BWD_STEPS) -- never on the stock ATen /
    MIOpen LSTM, which is replaced by a function that raises here.  Gradients against nn.LSTM autograd in float64 on the CPU."""
    import contextlib
    from onssen_amd.nn._core import BLSTMParams, _XcdPolicy

    def no_aten_lstm(*a, **k):
        raise AssertionError("the GPU training path called torch._VF.lstm")
    torch.manual_seed(H + L)
    ref = torch.nn.LSTM(F, H, L, batch_first=True, bidirectional=True).double()
    rnn = BLSTMParams(F, H, L, dropout=0.0)
    rnn.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    rnn = rnn.to(dev)
    x = torch.randn(B, T, F, dtype=torch.float64)
    R = torch.randn(B, T, 2 * H, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    yr, _ = ref(xr)
    (yr * R).sum().backward()
    monkeypatch.setattr(torch._VF, "lstm", no_aten_lstm, raising=False)
    n_p = _XcdPolicy.persistent_launches
    xg = x.float().to(dev).requires_grad_(True)
    with (_XcdPolicy.forced_steps() if forced else contextlib.nullcontext()):
        yg = rnn.autograd_forward(xg, True)
        (yg * R.float().to(dev)).sum().backward()
    torch.cuda.synchronize()
    if 640 < H <= 768 and not forced:
        assert _XcdPolicy.persistent_launches > n_p          # the forward with saved state ran on the persistent kernel
    else:
        assert _XcdPolicy.persistent_launches == n_p
    assert (yg.detach().cpu().double() - yr.detach()).abs().max() < 2e-5
    bad = []

    def close(a, b, what):
        a, b = a.detach().cpu().double(), b.detach()
        rl2 = ((a - b).norm() / b.norm().clamp_min(1e-300)).item()
        if not rl2 <= 1e-4:
            bad.append(f"{what}: rel-L2 {rl2:.2e}")
    close(xg.grad, xr.grad, "dx")
    for name, p in rnn.named_parameters():
        close(p.grad, getattr(ref, name).grad, name)
    assert not bad, "; ".join(bad)


@pytest.mark.gpu
def test_cfg4_full_shape_gradients_match_fp64_autograd(dev, monkeypatch):
    """BASELINE cfg4's training shape as shipped -- 16 chunks x 400 frames, 3 x BLSTM-600, dropout off -- through the
    default HIP training path (persistent forward with saved state, persistent backward recurrence, split-bf16 gradient
    GEMMs) against nn.LSTM autograd in float64 on the CPU: EVERY gradient tensor by per-tensor rel-L2 (a small tensor
    cannot hide behind another's largest entry) and by the max-norm bound of the short-sequence tests.  400 dependent
    steps of split-bf16 products (~1e-5 relative each) in both directions of time; the measured errors are printed and
    recorded in DESIGN.md."""
    from onssen_amd.nn._core import BLSTMParams, _XcdStatus
    monkeypatch.setenv("ONSSEN_TRAIN_HIP", "1")
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    B, T, F, H, L = 16, 400, 129, 600, 3
    torch.manual_seed(404)
    ref = torch.nn.LSTM(F, H, L, batch_first=True, bidirectional=True).double()
    rnn = BLSTMParams(F, H, L, dropout=0.0)
    rnn.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    rnn = rnn.to(dev)
    x = torch.randn(B, T, F, dtype=torch.float64)
    R = torch.randn(B, T, 2 * H, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    yr, _ = ref(xr)
    (yr * R).sum().backward()
    xg = x.float().to(dev).requires_grad_(True)
    yg = rnn.autograd_forward(xg, True)
    (yg * R.float().to(dev)).sum().backward()
    torch.cuda.synchronize()
    _XcdStatus.poll(wait=True)
    assert (yg.detach().cpu().double() - yr.detach()).abs().max() < 2e-5
    rows, bad = [], []

    def close(a, b, what):
        a, b = a.detach().cpu().double(), b.detach()
        rl2 = ((a - b).norm() / b.norm().clamp_min(1e-300)).item()
        mx = ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()
        rows.append(f"{what}: rel-L2 {rl2:.2e}, max-norm-relative {mx:.2e}")
        if not (rl2 <= 1e-4 and mx <= 3e-4):
            bad.append(rows[-1])
    close(xg.grad, xr.grad, "dx")
    for name, p in rnn.named_parameters():
        close(p.grad, getattr(ref, name).grad, name)
    print("[cfg4 16x400 L3 H600]\n  " + "\n  ".join(rows))
    assert not bad, "; ".join(bad)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,F,H,L", [(16, 30, 129, 600, 3), (3, 11, 20, 30, 2)])
def test_blstm_training_with_dropout_matches_autograd_under_the_same_masks(dev, monkeypatch, B, T, F, H, L):
    """nn.LSTM(dropout=0.3) in training (onssen/nn/deep_clustering.py:15-22): the HIP path applies the inter-layer dropout with
    onssen_dropout_f32 (mask regenerated from the seed in the backward pass).  Reference: single-layer nn.LSTMs in float64 on
    the CPU with the SAME masks (the kernel run on ones with the seeds the path draws) -- outputs and every gradient agree
    as in the dropout-free test."""
    from onssen_amd.hip import get_lib
    from onssen_amd.nn._core import BLSTMParams
    monkeypatch.setenv("ONSSEN_TRAIN_HIP", "1")
    p = 0.3
    torch.manual_seed(H + L)
    ref = torch.nn.LSTM(F, H, L, batch_first=True, bidirectional=True).double()
    rnn = BLSTMParams(F, H, L, dropout=p)
    rnn.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    rnn = rnn.to(dev)
    x = torch.randn(B, T, F, dtype=torch.float64)
    R = torch.randn(B, T, 2 * H, dtype=torch.float64)
    torch.manual_seed(77)
    seeds = [int(torch.randint(0, 2 ** 62, (1,)).item()) for _ in range(L - 1)]
    lib = get_lib()
    ug = 4 * -(-H // 128)
    Hp = lib.lstm_geometry(H, ug)[0]
    masks = []
    for sd in seeds:
        ones = torch.ones(T, B, 2, Hp, device=dev)
        m = torch.empty_like(ones)
        lib.dropout(ones.data_ptr(), ones.numel(), p, sd, m.data_ptr(), torch.cuda.current_stream().cuda_stream)
        masks.append(m[..., :H].reshape(T, B, 2 * H).transpose(0, 1).cpu().double())
        assert abs((masks[-1] != 0).double().mean().item() - (1 - p)) < 5 * (p * (1 - p) / masks[-1].numel()) ** 0.5
    # the reference stack, one layer at a time
    layers = []
    for l in range(L):
        one = torch.nn.LSTM(F if l == 0 else 2 * H, H, 1, batch_first=True, bidirectional=True).double()
        sdl = {}
        for kind in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            for sfx in ("", "_reverse"):
                sdl[f"{kind}_l0{sfx}"] = ref.state_dict()[f"{kind}_l{l}{sfx}"]
        one.load_state_dict(sdl)
        layers.append(one)
    xr = x.clone().requires_grad_(True)
    h = xr
    for l, one in enumerate(layers):
        h, _ = one(h)
        if l < L - 1:
            h = h * masks[l]
    (h * R).sum().backward()
    torch.manual_seed(77)
    xg = x.float().to(dev).requires_grad_(True)
    yg = rnn.autograd_forward(xg, True)
    (yg * R.float().to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert (yg.detach().cpu().double() - h.detach()).abs().max() < 4e-5
    bad = []

    def close(a, b, what):
        a, b = a.detach().cpu().double(), b.detach()
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        if not err <= 3e-4 * max(scale, 1e-6):
            bad.append(f"{what}: max err {err:.3e} vs max |ref| {scale:.3e}")
    close(xg.grad, xr.grad, "dx")
    for name, prm in rnn.named_parameters():
        l = int(name.replace("_reverse", "").rsplit("_l", 1)[1])
        close(prm.grad, getattr(layers[l], name.replace(f"_l{l}", "_l0")).grad, name)      # e.g. weight_hh_l2_reverse -> layer 2's weight_hh_l0_reverse
    assert not bad, "; ".join(bad)


@pytest.mark.gpu
def test_dc_training_step_hip_vs_aten(dev, monkeypatch):
    """One deep-clustering training forward + loss_dc + backward (dropout off so both paths see the same network): the
    HIP BLSTM path and the stock ATen LSTM give the same loss and gradients (2e-3 of each tensor's largest entry: two
    fp32 implementations of a T-step recurrence); with dropout on, the HIP path still trains (finite, loss decreases)."""
    from onssen_amd import nn as onn
    from onssen_amd.loss import loss_dc
    torch.manual_seed(1)
    B, T, Fq = 4, 60, 129
    x = torch.randn(B, T, Fq, device=dev)
    lab = torch.nn.functional.one_hot(torch.randint(0, 2, (B, T, Fq), device=dev), 2).float()
    wt = torch.rand(B, T, Fq, device=dev)
    results = {}
    import contextlib
    from tools.aten_lstm_reference import aten_lstm_reference
    monkeypatch.setenv("ONSSEN_TRAIN_HIP", "1")
    for mode in ("1", "0"):      # "0": the stock ATen / MIOpen LSTM + ATen heads, patched in by the test (the product has no such path)
        with (aten_lstm_reference() if mode == "0" else contextlib.nullcontext()):
            torch.manual_seed(2)
            m = onn.deep_clustering(Fq, 600, 2, 20, dropout=0.0).to(dev).train()
            out = m([x])
            loss = torch.mean(loss_dc(out, [lab, wt]))      # as trainer.train does (onssen/utils/train.py:78)
            loss.backward()
            results[mode] = (loss.item(), {k: p.grad.clone() for k, p in m.named_parameters()})
    assert abs(results["1"][0] - results["0"][0]) <= 1e-4 * abs(results["0"][0])
    for k, g0 in results["0"][1].items():
        g1 = results["1"][1][k]
        assert (g1 - g0).abs().max().item() <= 2e-3 * max(g0.abs().max().item(), 1e-8), k
    monkeypatch.setenv("ONSSEN_TRAIN_HIP", "1")
    torch.manual_seed(3)
    m = onn.deep_clustering(Fq, 600, 2, 20, dropout=0.3).to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = torch.mean(loss_dc(m([x]), [lab, wt]))
        loss.backward()
        assert all(torch.isfinite(p.grad).all() for p in m.parameters())
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(4, 60), (16, 400)])
def test_fused_head_and_loss_of_the_train_step(dev, monkeypatch, B, T):
    """VERDICT r3 missing #3: ``dist.train_step`` holds the labels while the forward runs and takes
    ``deep_clustering.fused_loss_dc`` -- fc_dc + F.normalize + loss_dc as one autograd node whose backward goes from the
    embedding straight to the operands of fc_dc's gradient GEMMs.  Same loss and the same gradients as ``loss_dc(model(x),
    label)`` (two orders of fp32 summation), and train_step really takes it (no separate loss-gradient kernel output exists:
    checked through the function it calls)."""
    from onssen_amd import nn as onn
    from onssen_amd.dist import train_step
    from onssen_amd.loss import loss_dc
    torch.manual_seed(1)
    Fq = 129
    x = torch.randn(B, T, Fq, device=dev)
    lab = torch.nn.functional.one_hot(torch.randint(0, 3, (B, T, Fq), device=dev), 3)[..., :2].double()     # silent bins too
    wt = torch.rand(B, T, Fq, device=dev)
    res = {}
    for mode in ("fused", "plain"):
        torch.manual_seed(2)
        m = onn.deep_clustering(Fq, 600, 2, 20, dropout=0.0).to(dev).train()
        if mode == "fused":
            loss_t = m.fused_loss_dc([x], [lab, wt])
            assert loss_t is not None
        else:
            loss_t = loss_dc(m([x]), [lab, wt])
        loss = torch.mean(loss_t)
        loss.backward()
        res[mode] = (loss.item(), {k: p.grad.clone() for k, p in m.named_parameters()}, m.bn.running_mean.clone())
    assert abs(res["fused"][0] - res["plain"][0]) <= 1e-5 * abs(res["plain"][0])
    assert torch.equal(res["fused"][2], res["plain"][2])
    for k, g0 in res["plain"][1].items():
        g1 = res["fused"][1][k]
        rl2 = ((g1 - g0).norm() / g0.norm().clamp_min(1e-30)).item()
        assert rl2 <= 1e-4, (k, rl2)        # (measured: <= 2.2e-5 at the shipped shape, 400 dependent steps)
    # train_step takes the fused node for this model / loss pair, and the generic route when told not to
    calls = []
    orig = onn.deep_clustering.fused_loss_dc
    monkeypatch.setattr(onn.deep_clustering, "fused_loss_dc", lambda self, i, l: (calls.append(1), orig(self, i, l))[1])
    losses = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("ONSSEN_TRAIN_FUSED_LOSS", flag)
        torch.manual_seed(2)
        m = onn.deep_clustering(Fq, 600, 2, 20, dropout=0.0).to(dev).train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        losses[flag] = [float(train_step(m, opt, loss_dc, [x], [lab, wt])) for _ in range(3)]
    assert len(calls) == 6 and losses["1"][-1] < losses["1"][0]
    np.testing.assert_allclose(losses["1"], losses["0"], rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("H,Kx,T,B", [(600, 1200, 400, 16), (600, 1200, 37, 5), (64, 128, 50, 3), (600, 129, 400, 16)])
def test_weight_gradients_from_row_major_images(dev, H, Kx, T, B):
    """onssen_lstm_wgrad_images_f32 (contraction over the rows of row-major x3 images: gfx950's transposing LDS read, h_prev as
    a row shift) against the transposed-image route (onssen_x3_image_t_f32 + onssen_linear_x3p_batched_split_alt): bit for bit,
    at the shipped training shape; both timed."""
    from onssen_amd.hip import get_lib
    lib = get_lib()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(H + T)
    NP, Hp, K = 4 * H, H, T * B
    dP, y, x = (torch.randn(K, n, device=dev) for n in (2 * NP, 2 * Hp, Kx))
    KB = (K + 31) // 32
    def rows_img(m):
        o = torch.empty(m.shape[0], (m.shape[1] + 31) // 32, 2, 32, device=dev, dtype=torch.int16)
        lib.x3_image(m.data_ptr(), m.shape[1], 0, 1, m.shape[0], m.shape[1], o.data_ptr(), st)
        return o
    zero = torch.zeros(max(Hp + Kx, 8), device=dev)
    def transposed_route():
        a_t = torch.empty(2 * NP, KB, 2, 32, device=dev, dtype=torch.int16)
        lib.x3_image_t(dP.data_ptr(), 2 * NP, 2 * NP, K, 0, a_t.data_ptr(), st)
        w1 = torch.empty(Hp + Kx + Hp, KB, 2, 32, device=dev, dtype=torch.int16)
        lib.x3_image_t(y.data_ptr(), 2 * Hp, Hp, K, -B, w1.data_ptr(), st)
        lib.x3_image_t(x.data_ptr(), Kx, Kx, K, 0, w1[Hp:].data_ptr(), st)
        lib.x3_image_t(y[:, Hp:].data_ptr(), 2 * Hp, Hp, K, B, w1[Hp + Kx:].data_ptr(), st)
        ih, hh = torch.empty(2, 4 * H, Kx, device=dev), torch.empty(2, 4 * H, H, device=dev)
        lib.linear_x3p_batched_split_alt(a_t.data_ptr(), NP * KB * 64, NP, K, w1.data_ptr(), Hp * KB * 64, zero.data_ptr(), Hp + Kx, 4,
                                         hh.data_ptr(), 4 * H * H, H, H * H, Hp, ih.data_ptr(), 4 * H * Kx, Kx, H * Kx, Kx, 2, st)
        return ih, hh
    dp_img, y_img, x_img = rows_img(dP), rows_img(y), rows_img(x)
    def row_major_route():
        ih, hh = torch.empty(2, 4 * H, Kx, device=dev), torch.empty(2, 4 * H, H, device=dev)
        lib.lstm_wgrad_images(dp_img.data_ptr(), y_img.data_ptr(), x_img.data_ptr(), K, B, NP, Hp, Kx, zero.data_ptr(), 4,
                              ih.data_ptr(), 4 * H * Kx, Kx, H * Kx, hh.data_ptr(), 4 * H * H, H, H * H, st)
        return ih, hh
    ih_a, hh_a = transposed_route()
    ih_b, hh_b = row_major_route()
    assert torch.equal(ih_b, ih_a) and torch.equal(hh_b, hh_a)
    hprev = torch.zeros(K, Hp, device=dev, dtype=torch.float64); hprev[B:] = y[:-B, :Hp].double()
    ref = (dP[:, :NP].double().t() @ hprev).reshape(H, 4, Hp).transpose(0, 1).reshape(4 * H, Hp)
    assert ((hh_b[0].double() - ref).norm() / ref.norm()).item() < 1e-5
    times = {}
    for name, fn in (("transposed images + GEMM", transposed_route), ("row-major GEMM", row_major_route)):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[name] = e0.elapsed_time(e1) * 100
    print(f"H={H} Kx={Kx} T={T} B={B}: " + ", ".join(f"{k} {v:.1f} us" for k, v in times.items()))


@pytest.mark.gpu
def test_row_major_weight_gradients_in_the_training_path(dev, monkeypatch):
    """ONSSEN_TRAIN_WGRAD_ROWS (default on): the layers' weight gradients from row-major images -- dP's, and the persistent
    forward's OWN images of its input and output still sitting in its workspace -- against the transposed-image route: the same
    gradients bit for bit, also when a second forward has overwritten the workspace before the first one's backward runs (the
    backward then makes the images itself)."""
    from onssen_amd.nn._core import BLSTMParams
    B, T, F, H, L = 16, 64, 129, 600, 3
    torch.manual_seed(7)
    x1, x2 = torch.randn(B, T, F, device=dev), torch.randn(B, T, F, device=dev)
    R1, R2 = torch.randn(B, T, 2 * H, device=dev), torch.randn(B, T, 2 * H, device=dev)
    grads = {}
    for rows in ("1", "0"):
        monkeypatch.setenv("ONSSEN_TRAIN_WGRAD_ROWS", rows)
        for two in (False, True):
            torch.manual_seed(8)
            rnn = BLSTMParams(F, H, L, dropout=0.0).to(dev)
            y1 = rnn.autograd_forward(x1, True)
            if two:
                y2 = rnn.autograd_forward(x2, True)
                ((y1 * R1).sum() + (y2 * R2).sum()).backward()
            else:
                (y1 * R1).sum().backward()
            grads[(rows, two)] = [p.grad.clone() for p in rnn.parameters()]
    for two in (False, True):
        for a, b in zip(grads[("1", two)], grads[("0", two)]):
            assert torch.equal(a, b)
    assert not torch.equal(grads[("1", False)][0], grads[("1", True)][0])


@pytest.mark.gpu
def test_chimera_training_step_hip_vs_aten(dev, monkeypatch):
    """chimera++ (4 x BLSTM-600, no BatchNorm) + loss_chimera_msa: the HIP training path against the stock ATen LSTM,
    dropout off.  Same bound as the deep-clustering test: 2e-3 of each gradient tensor's largest entry."""
    from onssen_amd import nn as onn
    from onssen_amd.loss import loss_chimera_msa
    torch.manual_seed(5)
    B, T, Fq = 3, 40, 129
    x = torch.randn(B, T, Fq, device=dev)
    lab = [torch.nn.functional.one_hot(torch.randint(0, 2, (B, T, Fq), device=dev), 2).float(),
           torch.rand(B, T, Fq, device=dev) + 0.1, torch.rand(B, T, Fq, device=dev), torch.rand(B, T, Fq, device=dev)]
    results = {}
    import contextlib
    from tools.aten_lstm_reference import aten_lstm_reference
    monkeypatch.setenv("ONSSEN_TRAIN_HIP", "1")
    for mode in ("1", "0"):
        with (aten_lstm_reference() if mode == "0" else contextlib.nullcontext()):
            torch.manual_seed(6)
            m = onn.chimera(Fq, 600, 4, 20, dropout=0.0).to(dev).train()
            loss = torch.mean(loss_chimera_msa(m([x]), lab))
            loss.backward()
            results[mode] = (loss.item(), {k: p.grad.clone() for k, p in m.named_parameters()})
    assert abs(results["1"][0] - results["0"][0]) <= 1e-4 * abs(results["0"][0])
    for k, g0 in results["0"][1].items():
        g1 = results["1"][1][k]
        assert (g1 - g0).abs().max().item() <= 2e-3 * max(g0.abs().max().item(), 1e-8), k
