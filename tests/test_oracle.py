"""The CPU oracle (oracle/np_oracle.py) against the committed golden vectors
(outputs of the reference's own onssen.nn modules, tools/gen_golden.py) and
against torch.stft/istft for the librosa-backed front/back end."""
import glob
import os

import numpy as np
import pytest
import torch

from onssen_amd.synthetic import make_state_dict, synth_mixture
from oracle import np_oracle as O


def rel_l2(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    return np.linalg.norm(a - b, axis=-1) / np.maximum(np.linalg.norm(b, axis=-1), 1e-30)


def load_case(fn):
    z = np.load(fn)
    sd = make_state_dict(str(z["kind"]), int(z["F"]), int(z["H"]), int(z["L"]), int(z["D"]),
                         int(z["C"]), seed=int(z["seed"]), gain=float(z["gain"]))
    return z, sd


@pytest.mark.parametrize("name", ["g1_deep_clustering_H8_L1", "g1_deep_clustering_H32_L2"])
def test_dc_tiny_matches_reference(golden_dir, name):
    z, sd = load_case(f"{golden_dir}/{name}.npz")
    emb = O.deep_clustering_forward(sd, z["x"])
    assert emb.shape == z["out_embedding"].shape
    np.testing.assert_allclose(emb, z["out_embedding"], atol=2e-6, rtol=0)
    assert rel_l2(emb, z["out_embedding"]).max() < 1e-5


@pytest.mark.parametrize("name", ["g1_deep_clustering_H8_L1", "g1_deep_clustering_H32_L2"])
def test_rounded_restatement_is_the_same_network(golden_dir, name):
    """``deep_clustering_forward_rounded`` (the restatement of the opt-in bf16 mode the GPU tests pin that mode with) with an
    identity rounding IS the reference network -- BatchNorm folded, the lone-column form of the first layer included --,
    and with bf16 rounding it is bf16-grade close to the reference's golden vectors."""
    z, sd = load_case(f"{golden_dir}/{name}.npz")
    for tail in (False, True):
        emb = O.deep_clustering_forward_rounded(sd, z["x"], rnd=lambda v: v, exact_tail0=tail)
        np.testing.assert_allclose(emb, z["out_embedding"], atol=3e-6, rtol=0)
    emb = O.deep_clustering_forward_rounded(sd, z["x"])
    assert 1e-4 < rel_l2(emb, z["out_embedding"]).max() < 1e-1      # bf16-grade (tiny configs with over-scaled weights)
    r = O.bf16_round(np.float32([1.0 + 2.0 ** -9, 1.0 + 3 * 2.0 ** -9, -3.1415927, 1e-30]))
    assert r[0] == 1.0 and r[1] == np.float32(1.0 + 2.0 ** -7) and (r.view(np.uint32) & 0xFFFF == 0).all()   # ties to even


def test_chimera_tiny_matches_reference(golden_dir):
    z, sd = load_case(f"{golden_dir}/g1_chimera_H32_L2.npz")
    e, a, b = O.chimera_forward(sd, z["x"])
    np.testing.assert_allclose(e, z["out_embedding"], atol=2e-6)
    np.testing.assert_allclose(a, z["out_mask_A"], atol=2e-6)
    np.testing.assert_allclose(b, z["out_mask_B"], atol=2e-6)


def test_phase_net_tiny_matches_reference(golden_dir):
    z, sd = load_case(f"{golden_dir}/g1_phase_net_H16_L2.npz")
    outs = O.phase_net_forward(sd, z["x"], z["x_phase"])
    for o, n in zip(outs, ["embedding", "mask_A", "mask_B", "phase_A", "phase_B"]):
        # phase_* normalise a 2-vector that can be short: fp32 round-off is amplified there
        np.testing.assert_allclose(o, z["out_" + n], atol=3e-5 if n.startswith("phase") else 3e-6, err_msg=n)


def _logmag_input(seed, B, T):
    return np.stack([O.log_magnitude(O.stft(synth_mixture(seed * 100 + b, (T - 1) * 64), 256, 64))
                     for b in range(B)])


def test_dc_full_size_matches_reference_subsample(golden_dir):
    """cfg1 (DC 2xBLSTM-600) at full width: the oracle vs the reference's
    strided output subsample and per-frame checksums."""
    z, sd = load_case(f"{golden_dir}/g2_cfg1_dc_L2.npz")
    x = _logmag_input(int(z["x_seed"]), int(z["B"]), int(z["T"]))
    emb = O.deep_clustering_forward(sd, x)
    np.testing.assert_allclose(emb[:, ::40, ::16, :], z["emb_sub"], atol=1e-5)
    assert rel_l2(emb[:, ::40, ::16, :], z["emb_sub"]).max() < 1e-4
    np.testing.assert_allclose(emb.astype(np.float64).sum(axis=(2, 3)), z["emb_sum_per_frame"],
                               atol=2e-3)


def test_stft_against_torch():
    sig = synth_mixture(3, 25536)
    X = O.stft(sig, 256, 64)
    assert X.shape == (400, 129) and X.dtype == np.complex64
    Xt = torch.stft(torch.from_numpy(sig).double(), 256, 64, window=torch.hann_window(256, periodic=True).double(),
                    center=True, pad_mode="reflect", return_complex=True).numpy().T
    assert np.abs(X - Xt).max() < 1e-5 * np.abs(Xt).max()
    lm = O.log_magnitude(X)
    assert lm.dtype == np.float32
    # log10 amplifies the complex64 rounding of near-zero bins: compare where |X| is not tiny
    big = np.abs(Xt) > 1e-3
    np.testing.assert_allclose(lm[big], np.log10(np.abs(Xt) + 1e-7)[big], atol=1e-4)
    np.testing.assert_allclose(10.0 ** lm.astype(np.float64), np.abs(Xt) + 1e-7, atol=1e-5)
    ph = O.phase_re_im(X)
    assert ph.shape == (400, 129, 2) and np.array_equal(ph[..., 1], X.imag)


@pytest.mark.parametrize("n_fft,hop,n", [(256, 64, 25536), (256, 64, 9000), (512, 128, 16000)])
def test_istft_against_torch_and_roundtrip(n_fft, hop, n):
    sig = synth_mixture(9, n)
    X = O.stft(sig, n_fft, hop)
    y = O.istft(X, hop, n)
    yt = torch.istft(torch.from_numpy(X.T.astype(np.complex128)), n_fft, hop,
                     window=torch.hann_window(n_fft, periodic=True).double(), center=True, length=n).numpy()
    assert np.abs(y - yt).max() < 1e-6
    assert np.abs(y - sig).max() < 1e-5       # COLA: stft -> istft reproduces the signal
    # masks that sum to one split the signal additively
    rng = np.random.default_rng(0)
    m = rng.random(X.shape)
    parts = O.mask_istft(X, np.stack([m, 1 - m]), hop, n)
    assert np.abs(parts.sum(0) - y).max() < 1e-6


@pytest.mark.parametrize("n_fft,hop,n", [(256, 64, 25536), (512, 128, 16000), (256, 100, 4000)])
def test_istft_float32_overlap_add_variant(n_fft, hop, n):
    """``istft_f32_ola`` follows librosa 0.7 / 0.8's dtype rule literally (float32 output buffer and window
    sum, float64 frames): it must agree with the float64-accumulating restatement to a few float32 ulps of the signal scale --
    the band inside which the HIP kernel's output is then required to lie (tests/test_gpu_parity.py)."""
    sig = synth_mixture(9, n)
    X = O.stft(sig, n_fft, hop)
    rng = np.random.default_rng(1)
    Xm = (X * rng.random(X.shape)).astype(np.complex64)
    y64, y32 = O.istft(Xm, hop, n), O.istft_f32_ola(Xm, hop, n)
    assert y32.dtype == np.float32 and y32.shape == y64.shape
    scale = np.abs(y64).max()
    inner = slice(n_fft, n - n_fft)            # (the edges divide by a small window sum: compare where it is ~1.5)
    assert np.abs(y32[inner] - y64[inner]).max() <= 4 * np.finfo(np.float32).eps * scale
    assert np.abs(y32 - y64).max() <= 1e-5 * scale


def test_istft_longer_than_signal_pads_with_zeros():
    X = O.stft(synth_mixture(1, 2000), 256, 64)
    y = O.istft(X, 64, 2600)
    assert y.shape == (2600,) and np.all(y[X.shape[0] * 64 + 128:] == 0)


def test_torch_cpu_restatement_matches_reference_golden(golden_dir):
    """oracle/torch_cpu.py (the timed cpu_baseline 'port') against the
    reference's own outputs."""
    from oracle import torch_cpu as TC
    z, sd = load_case(f"{golden_dir}/g1_deep_clustering_H32_L2.npz")
    np.testing.assert_allclose(TC.deep_clustering_forward(sd, z["x"]).numpy(), z["out_embedding"], atol=1e-6)
    z, sd = load_case(f"{golden_dir}/g1_chimera_H32_L2.npz")
    e, a, b = TC.chimera_forward(sd, z["x"])
    np.testing.assert_allclose(e.numpy(), z["out_embedding"], atol=1e-6)
    np.testing.assert_allclose(b.numpy(), z["out_mask_B"], atol=1e-6)
    z, sd = load_case(f"{golden_dir}/g1_phase_net_H16_L2.npz")
    outs = TC.phase_net_forward(sd, z["x"], z["x_phase"])
    np.testing.assert_allclose(outs[3].numpy(), z["out_phase_A"], atol=1e-6)
    np.testing.assert_allclose(outs[4].numpy(), z["out_phase_B"], atol=1e-6)


def test_loss_dc_restatement_matches_reference_fixture(golden_dir):
    """N1: oracle/np_oracle.loss_dc against the reference's own loss_dc (tools/gen_golden.py G4): value and the
    (B, B) broadcast quirk, on the reference network's embedding reproduced by the oracle forward."""
    from onssen_amd.synthetic import make_state_dict
    z = np.load(f"{golden_dir}/g4_loss_dc.npz")
    sd = make_state_dict("deep_clustering", 129, int(z["H"]), int(z["L"]), 20, 2, seed=int(z["seed"]))
    emb = O.deep_clustering_forward(sd, z["x"])
    loss = O.loss_dc(emb, z["one_hot"], z["mag"])
    assert loss.shape == z["loss"].shape == (3, 3)
    np.testing.assert_allclose(loss, z["loss"], rtol=1e-4)
    np.testing.assert_allclose(loss.mean(), float(z["loss_mean"]), rtol=1e-4)


@pytest.mark.parametrize("name", ["g5_enhance_H16_L2", "g5_enhance_H32_L1"])
def test_enhance_restatement_and_module_match_reference(golden_dir, name):
    """N4: the oracle restatement of onssen.nn.enhance, and the drop-in module's autograd path with the reference's
    parameter names loaded strictly, against the reference's own outputs (tools/gen_golden_enhance.py)."""
    import torch
    from onssen_amd import nn as onn
    from onssen_amd.synthetic import make_state_dict
    z = np.load(f"{golden_dir}/{name}.npz")
    sd = make_state_dict("enhance", int(z["F"]), int(z["H"]), int(z["L"]), seed=int(z["seed"]), gain=float(z["gain"]))
    out = O.enhance_forward(sd, z["x"], z["mag_noisy"])
    np.testing.assert_allclose(out, z["out_clean"], atol=2e-5, rtol=1e-4)
    m = onn.enhance(int(z["F"]), int(z["H"]), int(z["L"]))
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m.eval()
    with pytest.raises(AssertionError, match="two tensors"):
        m([torch.from_numpy(z["x"])])
    with torch.enable_grad():
        got, = m([torch.from_numpy(z["x"]), torch.from_numpy(z["mag_noisy"])])      # CPU -> autograd path
    np.testing.assert_allclose(got.detach().numpy(), z["out_clean"], atol=2e-5, rtol=1e-4)


def test_batch_sdr_restatement_matches_reference_fixture(golden_dir):
    """N4: oracle/np_oracle.batch_sdr against the reference's batch_SDR_torch (tools/gen_golden_sdr.py)."""
    z = np.load(f"{golden_dir}/g6_batch_sdr.npz")
    for tag in ("c2", "c3m"):
        sdr, perm = O.batch_sdr(z[f"{tag}_est"], z[f"{tag}_org"], z[f"{tag}_mask"] if f"{tag}_mask" in z.files else None)
        np.testing.assert_allclose(sdr, z[f"{tag}_sdr"], rtol=1e-4, atol=1e-4)
        np.testing.assert_array_equal(perm, z[f"{tag}_perm"])


# ----------------------------------------------------------------------------- G3: label / feature helpers pinned by the reference
@pytest.mark.parametrize("tag", ["a", "b"])
def test_feature_helpers_match_reference_fixture(golden_dir, tag):
    """get_log_magnitude / get_phase / get_cos_difference / get_one_hot of the reference
    (onssen/data/feature_utils.py:49-95, run by tools/gen_golden_features.py) against the oracle restatements --
    bit for bit: they are the same NumPy expressions."""
    z = np.load(f"{golden_dir}/g3_features.npz")
    X, S1, S2 = z[f"{tag}_X"], z[f"{tag}_S1"], z[f"{tag}_S2"]
    feat = O.log_magnitude(X)
    assert feat.dtype == z[f"{tag}_log_magnitude"].dtype == np.float32
    np.testing.assert_array_equal(feat, z[f"{tag}_log_magnitude"])
    np.testing.assert_array_equal(O.log_magnitude(X, 1e-3), z[f"{tag}_log_magnitude_eps3"])
    np.testing.assert_array_equal(O.phase_re_im(X), z[f"{tag}_phase"])
    np.testing.assert_array_equal(O.cos_difference(X, S1), z[f"{tag}_cos_s1"])
    np.testing.assert_array_equal(O.cos_difference(X, S2), z[f"{tag}_cos_s2"])
    for db in (40, 20):
        oh = O.one_hot_labels(feat, np.abs(S1), np.abs(S2), db)
        assert oh.dtype == z[f"{tag}_one_hot_{db}"].dtype == np.float64          # the reference yields float64 labels
        np.testing.assert_array_equal(oh, z[f"{tag}_one_hot_{db}"])
    assert 0.0 < z[f"{tag}_one_hot_40"].sum(-1).mean() < 1.0                     # silent bins exist, active bins exist


@pytest.mark.filterwarnings("ignore:NOLA")
@pytest.mark.parametrize("n_fft,hop,n", [(256, 64, 25536), (512, 128, 16000), (256, 64, 5000)])
def test_stft_istft_second_cross_check_scipy(n_fft, hop, n):
    """The STFT / iSTFT legs are unpinned by the reference (librosa is not installable here).  Second independent
    cross-check next to torch.stft / torch.istft: scipy.signal (scipy 1.15) with librosa's conventions spelled out --
    periodic Hann, reflect padding of n_fft/2, no scaling on the forward transform."""
    from scipy import signal
    sig = synth_mixture(3, n)
    X = O.stft(sig, n_fft, hop)                                   # (T, F) complex64
    w = signal.get_window("hann", n_fft, fftbins=True)
    padded = np.pad(sig.astype(np.float64), n_fft // 2, mode="reflect")
    _, _, Z = signal.stft(padded, window=w, nperseg=n_fft, noverlap=n_fft - hop, boundary=None, padded=False,
                          return_onesided=True, scaling="spectrum")
    Z = (Z * w.sum()).T                                           # undo scipy's 1/sum(w) scaling
    assert Z.shape == X.shape
    np.testing.assert_allclose(X, Z.astype(np.complex64), atol=2e-5 * np.abs(Z).max())
    # inverse: scipy's istft (overlap-add / window sum-of-squares normalisation, NOLA) on the same spectrum
    y = O.istft(X, hop, n)
    _, ys = signal.istft((X.T.astype(np.complex128)) / w.sum(), window=w, nperseg=n_fft, noverlap=n_fft - hop, input_onesided=True,
                         boundary=None, scaling="spectrum")
    ys = ys[n_fft // 2:n_fft // 2 + n]
    m = min(len(ys), n)
    np.testing.assert_allclose(y[:m], ys[:m], atol=2e-6)
    np.testing.assert_allclose(y[n_fft:m - n_fft], sig[n_fft:m - n_fft], atol=2e-6)      # and the round trip itself
