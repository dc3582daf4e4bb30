"""Ragged batches of whole utterances on the GPU (round 4): the reference's evaluation runs one utterance per forward
(onssen/utils/test.py:29-41, batch-1 loader onssen/data/wsj0_2mix.py:231-245); here K utterances of different lengths share
every launch, and each utterance's results must be BIT FOR BIT those of its own batch-1 run (and match the oracle)."""
import numpy as np
import pytest
import torch

from onssen_amd.synthetic import make_state_dict, synth_mixture
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from onssen_amd.hip import get_lib
    get_lib()
    return torch.device("cuda:0")


def build(kind, dev, H=600, L=2, F=129, seed=3):
    from onssen_amd import nn as onn
    sd = make_state_dict(kind, F, H, L, 20, 2, seed=seed, gain=1.0)
    m = {"deep_clustering": onn.deep_clustering, "chimera": onn.chimera}[kind](F, H, L, 20)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.to(dev).eval(), sd


def ragged_features(frames, F, dev, seed=0, poison=True):
    g = torch.Generator().manual_seed(seed)
    T = max(frames)
    x = torch.full((len(frames), T, F), float("nan") if poison else 0.0)
    for b, Tb in enumerate(frames):
        x[b, :Tb] = torch.randn(Tb, F, generator=g) * 1.5 - 2.0
    return x.to(dev)


@pytest.mark.parametrize("K", [3, 16, 20, 40])       # one 4-row group | the chip full of 4-row groups | 8-row groups | two launches of 8-row groups
def test_dc_ragged_forward_is_the_batch_of_one_bit_for_bit(dev, monkeypatch, K):
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    m, sd = build("deep_clustering", dev)
    rng = np.random.default_rng(K)
    frames = [int(t) for t in rng.integers(60, 140, K)]
    frames[0] = 140
    x = ragged_features(frames, 129, dev, seed=K)              # the padding holds NaNs: nothing of it may spread
    with torch.no_grad():
        emb, = m([x], frames=frames)
        assert torch.isfinite(emb).all()
        for b in sorted({0, 1, K // 2, K - 1}):
            one, = m([x[b:b + 1, :frames[b]].contiguous()])
            assert torch.equal(emb[b, :frames[b]], one[0]), f"row {b}"
    for b in (1, K - 1):                                        # ... and the oracle (fp32 contract of the default mode)
        ref = O.deep_clustering_forward(sd, x[b:b + 1, :frames[b]].cpu().numpy())[0]
        got = emb[b, :frames[b]].cpu().numpy()
        np.testing.assert_allclose(got, ref, atol=1e-5, rtol=1e-4)


def test_chimera_ragged_forward(dev, monkeypatch):
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    m, sd = build("chimera", dev, L=3)
    frames = [90, 33, 71, 90, 12, 64, 50]
    x = ragged_features(frames, 129, dev, seed=5)
    with torch.no_grad():
        e, a, b_ = m([x], frames=torch.tensor(frames, dtype=torch.int32, device=dev))     # device tensor accepted as it is
        for r in (0, 4, 6):
            e1, a1, b1 = m([x[r:r + 1, :frames[r]].contiguous()])
            assert torch.equal(e[r, :frames[r]], e1[0]) and torch.equal(a[r, :frames[r]], a1[0]) and torch.equal(b_[r, :frames[r]], b1[0])
    ref = O.chimera_forward(sd, x[4:5, :12].cpu().numpy())
    np.testing.assert_allclose(a[4, :12].cpu().numpy(), ref[1][0], atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("mode", ["f32", "steps"])
def test_ragged_on_the_launch_per_step_recurrence(dev, monkeypatch, mode):
    """Exact fp32 (its persistent kernel has no ragged instantiation) and the re-run path of an aborted call.  These forms
    run their GEMMs on kernels that pick a different path for very few rows (a batch-1 utterance of 20 frames is 20 rows), so
    a row equals its own batch-1 run to fp32 rounding here, not bit for bit."""
    from onssen_amd.nn._core import _XcdPolicy
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    if mode == "f32":
        monkeypatch.setenv("ONSSEN_PRECISION", "f32")
    m, sd = build("deep_clustering", dev, H=96, L=2)
    frames = [20, 7, 13, 20, 1] + [11] * 14
    x = ragged_features(frames, 129, dev, seed=9)
    import contextlib
    with torch.no_grad(), (_XcdPolicy.forced_steps() if mode == "steps" else contextlib.nullcontext()):
        emb, = m([x], frames=frames)
        for b in (0, 1, 4, 18):
            one, = m([x[b:b + 1, :frames[b]].contiguous()])
            assert (emb[b, :frames[b]] - one[0]).abs().max().item() < (2e-5 if mode == "steps" else 2e-6)
    ref = O.deep_clustering_forward(sd, x[1:2, :7].cpu().numpy())[0]
    np.testing.assert_allclose(emb[1, :7].cpu().numpy(), ref, atol=5e-5 if mode == "steps" else 1e-5, rtol=1e-4)


def test_ragged_refused_in_bf16_mode_and_in_training(dev, monkeypatch):
    m, _ = build("deep_clustering", dev, H=64)
    x = ragged_features([5, 3], 129, dev, poison=False)
    monkeypatch.setenv("ONSSEN_PRECISION", "bf16")
    with torch.no_grad(), pytest.raises(RuntimeError, match="bf16"):
        m([x], frames=[5, 3])
    monkeypatch.setenv("ONSSEN_PRECISION", "bf16x3")
    with pytest.raises(RuntimeError, match="inference-path"):
        m.train()([x], frames=[5, 3])
    with torch.no_grad(), pytest.raises(ValueError, match="frames"):
        m.eval()([x], frames=[5, 9])


def test_separate_dc_ragged_end_to_end(dev, monkeypatch):
    """waveforms of different lengths -> STFT -> network -> threshold + 2-means -> masks -> iSTFT in ONE set of launches:
    every utterance's two estimates are bit for bit those of its own batch-1 call, zeros after its own length; and the STFT /
    iSTFT halves agree with the oracle."""
    from onssen_amd.separation import separate_dc
    from onssen_amd.features import stft_logmag
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    m, _ = build("deep_clustering", dev)
    ns = [64 * 150 + 17, 64 * 90, 64 * 149 + 63, 64 * 40 + 5, 64 * 120, 64 * 77 + 1, 64 * 150 + 17, 64 * 33, 64 * 101]
    n = max(ns)
    wav = torch.full((len(ns), n), float("nan"))
    for b, nb in enumerate(ns):
        wav[b, :nb] = torch.from_numpy(synth_mixture(70 + b, nb))
    wav = wav.to(dev)
    out = separate_dc(m, wav, lengths=ns)
    assert out.shape == (len(ns), 2, n) and torch.isfinite(out).all()
    for b in (0, 3, 5, 8):
        one = separate_dc(m, wav[b:b + 1, :ns[b]].contiguous())
        assert torch.equal(out[b, :, :ns[b]], one[0]), f"utterance {b}"
        assert (out[b, :, ns[b]:] == 0).all()
    lm, ri = stft_logmag(wav, lengths=ns)
    X = O.stft(wav[3, :ns[3]].cpu().numpy(), 256, 64)
    Tb = X.shape[0]
    np.testing.assert_allclose((ri[3, :Tb, :, 0] + 1j * ri[3, :Tb, :, 1]).cpu().numpy(), X, atol=1e-6 * np.abs(X).max())
    # both estimates add up to the masked mixture: the active bins' masks are complementary
    assert float((out[3, 0, :ns[3]] + out[3, 1, :ns[3]]).abs().max()) > 0


@pytest.mark.parametrize("kind", ["dc", "chimera"])
def test_eval_batched_equals_the_one_by_one_loop(dev, monkeypatch, kind):
    """tester.eval(batch=K): the reference's evaluation loop over K utterances per forward gives the SAME mean SI-SDR (every
    utterance's SDR is bit-identical; the mean is the same sum in the same order)."""
    monkeypatch.setenv("ONSSEN_SYNTHETIC_DATA", "1")
    from onssen_amd.data import wsj0_2mix_dataloader
    from onssen_amd.evaluate import tester_chimera, tester_dc
    fo = dict(batch_size=1, frame_length=100, sampling_rate=8000, window_size=256, hop_size=64, db_threshold=40)
    m, _ = build("deep_clustering" if kind == "dc" else "chimera", dev, H=300, L=2)
    loader = list(wsj0_2mix_dataloader(kind, fo, "tt", dev))
    assert len({l[1][2].shape[-1] for l in loader}) > 3            # utterances of different lengths
    t = (tester_dc if kind == "dc" else tester_chimera)(dict(model=m, test_loader=loader, device=str(dev), model_name=kind))
    one_by_one = t.eval(batch=1)
    for K in (3, 8, 16):
        got = t.eval(batch=K)
        assert abs(got - one_by_one) <= 1e-9 * max(1.0, abs(one_by_one)), (K, got, one_by_one)     # (fp64 sums grouped differently)
    got = t.eval(batch=3, bucket=2)                                # 6 utterances read ahead, sorted by length, two batches of 3
    assert abs(got - one_by_one) <= 1e-9 * max(1.0, abs(one_by_one))
    if kind == "dc":
        # round 6c: batch = 2 .. 16 of a two-layer network goes through separation.DCRaggedPipeline (the calls above did); the plain
        # loop gives the same mean, and so does a stream whose last chunk is smaller (drained, then run by the plain loop or a new pipeline)
        from onssen_amd.separation import DCRaggedPipeline
        calls = []
        real = DCRaggedPipeline.push_features
        monkeypatch.setattr(DCRaggedPipeline, "push_features", lambda self, *a, **k: (calls.append(1), real(self, *a, **k))[1])
        got = t.eval(batch=8)
        assert len(calls) == -(-len(loader) // 8) and abs(got - one_by_one) <= 1e-9 * max(1.0, abs(one_by_one))
        calls.clear()
        got = t.eval(batch=8, pipeline=False)
        assert not calls and abs(got - one_by_one) <= 1e-9 * max(1.0, abs(one_by_one))
        got = t.eval(batch=7, bucket=3)
        assert calls and abs(got - one_by_one) <= 1e-9 * max(1.0, abs(one_by_one))
        monkeypatch.setattr(DCRaggedPipeline, "push_features", real)
    # every utterance's own SDR, bit for bit
    from onssen_amd.evaluate import batch_SDR_torch
    with torch.no_grad():
        inp, lab, (frames, lengths) = t.collate(loader[:5])
        est, ref = t.get_est_sig(inp, lab, m(inp, frames=frames), frames=frames, lengths=lengths)
        sdr = batch_SDR_torch(est, ref, lengths=lengths)
        for k, (i1, l1) in enumerate(loader[:5]):
            e1, r1 = t.get_est_sig(i1, l1, m(i1))
            assert torch.equal(batch_SDR_torch(e1, r1), sdr[k:k + 1]), k
    # the collated batch itself: padded shapes, extents on the device
    inp, lab, (frames, lengths) = t.collate(loader[:3])
    assert inp[0].shape[0] == 3 and inp[0].shape[1] == int(frames.max()) and lab[2].shape[-1] == int(lengths.max())
    assert frames.tolist() == [l[0][0].shape[1] for l in loader[:3]]


@pytest.mark.parametrize("B,T,ragged", [(32, 400, False), (5, 77, False), (16, 140, True)])
def test_dc_masks_without_the_embedding_round_trip(dev, monkeypatch, B, T, ragged):
    """Round 4: ``dc_masks_from_features`` (threshold -> target map, fc_dc GEMM storing only the active bins' rows into the
    compacted array, clustering on it) gives the masks of ``dc_masks(model([logmag])[0], logmag)`` bit for bit -- at the
    headline shape, a small one, and a ragged batch -- and separate_dc (which takes that route) equals the explicit pipeline."""
    from onssen_amd.features import mask_istft, stft_logmag
    from onssen_amd.separation import dc_masks, dc_masks_from_features, separate_dc
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    m, _ = build("deep_clustering", dev)
    n = 64 * (T - 1) + 13
    rng = np.random.default_rng(B)
    ns = [int(v) for v in rng.integers(64 * 60, n, B)] if ragged else None
    if ragged:
        ns[0] = n
    wav = torch.from_numpy(np.stack([synth_mixture(300 + b % 8, n) for b in range(B)])).to(dev)
    with torch.no_grad():
        lengths = torch.tensor(ns, dtype=torch.int32, device=dev) if ragged else None
        frames = (1 + lengths // 64).to(torch.int32) if ragged else None
        logmag, ri = stft_logmag(wav, lengths=lengths)
        got = dc_masks_from_features(m, logmag, frames=frames)
        assert got is not None
        emb, = m([logmag]) if not ragged else m([logmag], frames=frames)
        want = dc_masks(emb, logmag, frames=frames)
        assert torch.equal(got, want)
        act = logmag >= (logmag.amax(dim=(1, 2), keepdim=True) - 2.0) if not ragged else None
        if act is not None:
            assert torch.equal(got.sum(-1) == 1, act)                    # exactly the bins within 40 dB of the loudest are labelled
        out = separate_dc(m, wav, lengths=lengths)
        assert torch.equal(out, mask_istft(ri, want, 64, wav.shape[-1], frames=frames, lengths=lengths))
        monkeypatch.setenv("ONSSEN_DC_COMPACT", "0")                     # the switch: the explicit pipeline
        assert dc_masks_from_features(m, logmag, frames=frames) is None
        assert torch.equal(separate_dc(m, wav, lengths=lengths), out)


def test_separate_chimera_ragged(dev, monkeypatch):
    """chimera++: masks straight from the network, K utterances of different lengths per call -- every row equals its own
    batch-1 call bit for bit, zeros after its own length."""
    from onssen_amd.separation import separate_chimera
    monkeypatch.setenv("ONSSEN_CHECK", "1")
    m, _ = build("chimera", dev, H=300, L=2)
    ns = [64 * 100 + 9, 64 * 37, 64 * 99 + 63, 64 * 100 + 9, 64 * 64]
    wav = torch.full((len(ns), max(ns)), float("nan"))
    for b, nb in enumerate(ns):
        wav[b, :nb] = torch.from_numpy(synth_mixture(170 + b, nb))
    wav = wav.to(dev)
    out = separate_chimera(m, wav, lengths=ns)
    assert torch.isfinite(out).all()
    for b in (0, 1, 4):
        one = separate_chimera(m, wav[b:b + 1, :ns[b]].contiguous())
        assert torch.equal(out[b, :, :ns[b]], one[0]) and (out[b, :, ns[b]:] == 0).all()
