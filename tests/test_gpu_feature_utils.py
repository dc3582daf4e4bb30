"""SURVEY section 8(b), data front-end surface: the reference's feature-helper NAMES (onssen/data/feature_utils.py:5-21,49-95) as
exported by onssen_amd.data.feature_utils, fed with the reference fixture tests/golden/g3_features.npz (tools/gen_golden_features.py:
the reference's own get_log_magnitude / get_phase / get_cos_difference / get_one_hot outputs on committed STFTs) and, for get_stft
(librosa upstream: parity unpinned), with wav files written here and the oracle's restatement."""
import numpy as np
import pytest
import torch

gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def fu():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    from onssen_amd.data import feature_utils
    return feature_utils


def test_names_and_signatures_are_the_reference_s():
    import inspect
    from onssen_amd import data
    from onssen_amd.data import feature_utils as m
    assert list(inspect.signature(m.get_stft).parameters) == ["fn", "sampling_rate", "window_size", "hop_size"]
    sig = inspect.signature(m.get_log_magnitude)
    assert list(sig.parameters) == ["stft", "epsilon"] and sig.parameters["epsilon"].default == 1e-7
    assert list(inspect.signature(m.get_phase).parameters) == ["stft"]
    assert list(inspect.signature(m.get_cos_difference).parameters) == ["stft_1", "stft_2"]
    assert list(inspect.signature(m.get_one_hot).parameters) == ["feature_mix", "mag_s1", "mag_s2", "db_threshold"]
    for name in ("get_stft", "get_log_magnitude", "get_phase", "get_cos_difference", "get_one_hot"):
        assert getattr(data, name) is getattr(m, name)


@gpu
def test_helpers_match_the_reference_fixture(fu, golden_dir):
    z = np.load(f"{golden_dir}/g3_features.npz")
    for tag in ("a", "b"):
        X, S1, S2 = z[f"{tag}_X"], z[f"{tag}_S1"], z[f"{tag}_S2"]
        feat = fu.get_log_magnitude(X)
        assert feat.dtype == np.float32 and feat.shape == X.shape
        np.testing.assert_allclose(10.0 ** feat.astype(np.float64), 10.0 ** z[f"{tag}_log_magnitude"].astype(np.float64), rtol=2e-6, atol=1e-9)
        err = np.abs(feat - z[f"{tag}_log_magnitude"]).max()
        feat3 = fu.get_log_magnitude(X, 1e-3)
        np.testing.assert_allclose(10.0 ** feat3.astype(np.float64), 10.0 ** z[f"{tag}_log_magnitude_eps3"].astype(np.float64), rtol=2e-6)
        ph = fu.get_phase(X)
        assert ph.dtype == np.float32
        np.testing.assert_array_equal(ph, z[f"{tag}_phase"])
        big = (np.abs(X) > 1e-3) & (np.abs(S1) > 1e-3) & (np.abs(S2) > 1e-3)
        for S, key in ((S1, "cos_s1"), (S2, "cos_s2")):
            c = fu.get_cos_difference(X, S)
            assert c.dtype == np.float32 and c.shape == X.shape
            np.testing.assert_allclose(c[big], z[f"{tag}_{key}"][big], atol=1e-5)
        for db in (40, 20):
            # the fixture's feature array in: bit for bit the reference's labels (the reference computes them from the same arrays)
            y = fu.get_one_hot(z[f"{tag}_log_magnitude"], np.abs(S1), np.abs(S2), db)
            assert y.dtype == np.float64 and y.shape == X.shape + (2,)
            np.testing.assert_array_equal(y, z[f"{tag}_one_hot_{db}"])
        print(f"g3 {tag}: get_log_magnitude max abs err {err:.2e} (log10 units)")


@gpu
def test_get_stft_reads_the_file_and_matches_the_restatement(fu, tmp_path):
    from onssen_amd.data import write_wav
    from onssen_amd.synthetic import synth_mixture
    from oracle import np_oracle as O
    for rate, n_fft, hop, n in ((8000, 256, 64, 64 * 57 + 13), (16000, 512, 128, 16000)):
        sig = synth_mixture(91, n, rate).astype(np.float32)
        fn = str(tmp_path / f"u{rate}.wav")
        write_wav(fn, sig, rate, subtype="FLOAT")
        X = fu.get_stft(fn, rate, n_fft, hop)
        ref = O.stft(sig, n_fft, hop)
        assert X.dtype == np.complex64 and X.shape == ref.shape == (1 + n // hop, n_fft // 2 + 1)
        np.testing.assert_allclose(X, ref, atol=2e-6 * np.abs(ref).max(), rtol=0)
        # ... and the chain a reference-side caller writes (wsj0_2mix.py:114-135): feature = get_log_magnitude(get_stft(...))
        np.testing.assert_allclose(10.0 ** fu.get_log_magnitude(X).astype(np.float64), 10.0 ** O.log_magnitude(ref).astype(np.float64),
                                   rtol=1e-5, atol=1e-8)
    # 16-bit PCM file at another rate: the resample branch (feature_utils.py:17-20) -> 8 kHz frame count
    sig16 = synth_mixture(92, 16000, 16000).astype(np.float32)
    fn = str(tmp_path / "u16.wav")
    write_wav(fn, sig16, 16000)
    X = fu.get_stft(fn, 8000, 256, 64)
    assert X.shape == (1 + 8000 // 64, 129) and np.isfinite(X.view(np.float32)).all()


@gpu
def test_bad_inputs_raise(fu):
    with pytest.raises(TypeError):
        fu.get_log_magnitude(np.zeros((4, 5), np.float32))
    with pytest.raises(ValueError):
        fu.get_cos_difference(np.zeros((4, 5), np.complex64), np.zeros((4, 6), np.complex64))
    with pytest.raises(ValueError):
        fu.get_one_hot(np.zeros((4, 5), np.float32), np.zeros((4, 5), np.float32), np.zeros((5, 4), np.float32), 40)


def test_no_cpu_fallback():
    """Without a ROCm device the helpers raise (this test is the CPU suite's half; on the GPU box it is skipped)."""
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    from onssen_amd.data import feature_utils as m
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.get_log_magnitude(np.ones((3, 4), np.complex64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.get_one_hot(np.zeros((4, 5), np.float32), np.zeros((4, 5), np.float32), np.zeros((4, 5), np.float32), 40)
