"""Row N4 tail / H1: wav I/O and the wsj0-2mix loader over real files (onssen/data/wsj0_2mix.py:26-37,78-79,103-158,216-228;
feature_utils.get_stft:5-21).  The corpus here is a handful of synthetic utterances written as RIFF files in the corpus layout."""
import os

import numpy as np
import pytest
import torch

from onssen_amd.data import Wsj02mixFiles, read_wav, write_wav, wsj0_2mix_dataloader
from onssen_amd.synthetic import synth_mixture
from oracle import np_oracle as O

FO = dict(batch_size=2, frame_length=40, sampling_rate=8000, window_size=256, hop_size=64, db_threshold=40)


def make_corpus(root, partition, lengths, rate=8000, subtype="PCM_16"):
    for sub in ("mix", "s1", "s2"):
        os.makedirs(os.path.join(root, "wav8k", "min", partition, sub), exist_ok=True)
    sigs = []
    for i, n in enumerate(lengths):
        trip = synth_mixture(500 + i, n, rate, return_sources=True)
        # keep the triple additive after 16-bit quantisation of the sources
        q = [np.rint(t * 0.5 * 32768) / 32768 for t in trip[1:]]
        trip = [(q[0] + q[1]).astype(np.float32), q[0].astype(np.float32), q[1].astype(np.float32)]
        for sub, sig in zip(("mix", "s1", "s2"), trip):
            write_wav(os.path.join(root, "wav8k", "min", partition, sub, f"utt{i:02d}.wav"), sig, rate, subtype)
        sigs.append(trip)
    return sigs


def test_read_wav_formats(tmp_path):
    from scipy.io import wavfile
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(1000) * 0.2).astype(np.float32)
    write_wav(tmp_path / "a.wav", x, 8000)                       # PCM16: quantised to 2^-15
    y, r = read_wav(tmp_path / "a.wav")
    assert r == 8000 and y.dtype == np.float32 and np.abs(y - x).max() <= 2.0 ** -16 + 1e-9
    write_wav(tmp_path / "b.wav", x, 16000, "FLOAT")
    y, r = read_wav(tmp_path / "b.wav")
    assert r == 16000 and np.array_equal(y, x)
    st = np.stack([x, -0.5 * x], 1)                               # two channels -> their mean (librosa mono=True)
    wavfile.write(tmp_path / "c.wav", 8000, st)
    y, _ = read_wav(tmp_path / "c.wav")
    np.testing.assert_allclose(y, 0.25 * x, atol=1e-7)
    wavfile.write(tmp_path / "d.wav", 8000, (x * 2 ** 31).astype(np.int32))
    y, _ = read_wav(tmp_path / "d.wav")
    np.testing.assert_allclose(y, x, atol=1e-6)
    wavfile.write(tmp_path / "e.wav", 8000, np.clip(np.rint(x * 128 + 128), 0, 255).astype(np.uint8))
    y, _ = read_wav(tmp_path / "e.wav")
    assert np.abs(y - x).max() <= 1 / 128


def test_factory_picks_files_or_synthetic(tmp_path, monkeypatch):
    monkeypatch.delenv("ONSSEN_SYNTHETIC_DATA", raising=False)
    make_corpus(str(tmp_path), "tr", [3000, 2500, 2800])
    fo = dict(FO, data_path=str(tmp_path))
    dl = wsj0_2mix_dataloader("dc", fo, "tr", device="cpu")
    assert isinstance(dl, Wsj02mixFiles) and len(dl) == 2 and len(dl.file_list) == 3
    with pytest.raises(FileNotFoundError):                                                          # data_path given, no cv files: never
        wsj0_2mix_dataloader("dc", fo, "cv", device="cpu")                                          # silently synthetic
    monkeypatch.setenv("ONSSEN_SYNTHETIC_DATA", "1")
    assert not isinstance(wsj0_2mix_dataloader("dc", fo, "cv", device="cpu"), Wsj02mixFiles)       # ... unless asked for
    monkeypatch.delenv("ONSSEN_SYNTHETIC_DATA")
    assert not isinstance(wsj0_2mix_dataloader("dc", dict(FO, data_path="synthetic"), "tr", device="cpu"), Wsj02mixFiles)
    assert not isinstance(wsj0_2mix_dataloader("dc", FO, "tr", device="cpu"), Wsj02mixFiles)       # no data_path
    with pytest.raises(ValueError):
        Wsj02mixFiles("conv-tasnet", fo, "tr")


@pytest.mark.gpu
@pytest.mark.parametrize("model_name", ["dc", "chimera++", "phase"])
def test_training_batches_from_files_match_the_oracle(tmp_path, model_name):
    """Batches over files: list layout per model, one utterance shorter than frame_length (repeated before the crop), the
    crop the seeded generator draws, log-magnitude / labels against the NumPy restatement of the reference's helpers."""
    dev = torch.device("cuda:0")
    L = FO["frame_length"]
    lengths = [64 * 70 + 13, 64 * 25, 64 * 55 + 7]          # 71, 26 (<= L: repeated twice), 56 frames
    sigs = make_corpus(str(tmp_path), "tr", lengths)
    fo = dict(FO, data_path=str(tmp_path))
    dl = Wsj02mixFiles(model_name, fo, "tr", device=dev, shuffle=False, seed=11)
    batches = list(dl)
    assert len(batches) == 2 and batches[0][0][0].shape == (2, L, 129) and batches[1][0][0].shape == (1, L, 129)
    nlab = {"dc": 2, "chimera++": 6, "phase": 6}[model_name]
    assert all(len(lab) == nlab for _, lab in batches) and batches[0][1][0].dtype == torch.float64
    # the prefetching producer draws the epoch's crops from a child generator spawned on the calling thread (the parent is never
    # touched off-thread): epoch 0 of seed 11 = the first child
    rng = np.random.default_rng(11).spawn(1)[0]
    k = 0
    for inp, lab in batches:
        for b in range(inp[0].shape[0]):
            trip = sigs[k]; k += 1
            specs = [O.stft(s, 256, 64) for s in trip]
            T = specs[0].shape[0]
            if T <= L:
                specs = [np.concatenate([s] * (L // T + 1), 0) for s in specs]
            start = int(rng.integers(0, specs[0].shape[0] - L))
            X, S1, S2 = (s[start:start + L] for s in specs)
            feat = O.log_magnitude(X)
            np.testing.assert_allclose(inp[0][b].cpu().numpy(), feat, atol=2e-5)
            got = lab[0][b].cpu().numpy()
            ref = O.one_hot_labels(feat, np.abs(S1), np.abs(S2), 40.0)
            assert (got != ref).mean() < 2e-4            # bins within rounding of the threshold / of a tie may flip
            np.testing.assert_allclose(lab[1][b].cpu().numpy(), np.abs(X), rtol=2e-5, atol=1e-6)
            if model_name == "chimera++":      # the phase of a bin with (almost) no energy is rounding noise: compare where both have some
                ok = np.minimum(np.abs(X), np.abs(S1)) > 1e-4
                assert ok.mean() > 0.5
                np.testing.assert_allclose(lab[4][b].cpu().numpy()[ok], O.cos_difference(X, S1)[ok], atol=2e-3)
            if model_name == "phase":
                np.testing.assert_allclose(inp[1][b].cpu().numpy(), np.stack([X.real, X.imag], -1), atol=2e-5)
                np.testing.assert_allclose(lab[5][b].cpu().numpy(), np.stack([S2.real, S2.imag], -1), atol=2e-5)


@pytest.mark.gpu
def test_eval_partition_from_files_and_tester(tmp_path):
    """Partition "tt": whole utterances, batch 1, [Re, Im, sig_ref] with the signals padded by 32 - n % 32 (get_sigs); the
    batches drive onssen_amd.evaluate.tester like the reference's test loop drives its tester."""
    dev = torch.device("cuda:0")
    lengths = [4000, 5017]
    sigs = make_corpus(str(tmp_path), "tt", lengths)
    fo = dict(FO, data_path=str(tmp_path))
    dl = wsj0_2mix_dataloader("dc", fo, "tt", device=dev)
    assert isinstance(dl, Wsj02mixFiles) and len(dl) == 2
    for (inp, lab), n, trip in zip(dl, lengths, sigs):
        npad = n + 32 - n % 32
        T = 1 + n // 64                      # the STFT of the file as it is (get_stft(fn)); only sig_ref is padded (get_sigs)
        assert inp[0].shape == (1, T, 129) and lab[0].shape == (1, T, 129) and lab[2].shape == (1, 2, npad)
        np.testing.assert_array_equal(lab[2][0, 0, :n].cpu().numpy(), trip[1])
        assert not lab[2][0, :, n:].any()
        X = O.stft(trip[0], 256, 64)
        np.testing.assert_allclose(lab[0][0].cpu().numpy(), X.real, atol=2e-5)
    from onssen_amd import nn as onn
    from onssen_amd.evaluate import tester_dc
    torch.manual_seed(0)
    model = onn.deep_clustering(129, 32, 1, 8).to(dev).eval()
    t = tester_dc(dict(model_name="dc", model=model, test_loader=dl, device="cuda:0"), hop_size=64)
    sdr = t.eval()
    assert np.isfinite(sdr)


def test_native_batch_reader_matches_read_wav_bit_for_bit(tmp_path):
    """onssen_wav_read_batch_f32 / onssen_wav_info (csrc/wav_io.inc; host code, runs without a GPU) against ``read_wav`` (scipy)
    on every sample format the loader accepts, a missing file and a row that is too short."""
    import struct
    from scipy.io import wavfile
    from onssen_amd import _abi
    from onssen_amd.hip import get_lib
    lib = get_lib()
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(5000) * 0.2).astype(np.float32)
    p = lambda n: str(tmp_path / n)
    write_wav(p("pcm16.wav"), x, 8000)
    write_wav(p("f32.wav"), x[:3001], 16000, "FLOAT")
    wavfile.write(p("stereo.wav"), 8000, np.stack([x, -0.5 * x], 1))
    wavfile.write(p("pcm32.wav"), 8000, (x * 2 ** 31).astype(np.int32))
    wavfile.write(p("u8.wav"), 8000, np.clip(np.rint(x * 128 + 128), 0, 255).astype(np.uint8))
    wavfile.write(p("f64.wav"), 8000, x.astype(np.float64))
    wavfile.write(p("three.wav"), 8000, np.stack([(x * 2e4).astype(np.int16), (x * 1e4).astype(np.int16), (x * 3e3).astype(np.int16)], 1))
    v = (np.clip(x, -1, 1) * (2 ** 23 - 1)).astype(np.int32)
    raw = np.stack([v & 255, (v >> 8) & 255, (v >> 16) & 255], 1).astype(np.uint8).tobytes()
    with open(p("pcm24.wav"), "wb") as f:       # with an odd-sized LIST chunk in front of the data, like files in the wild
        f.write(b"RIFF" + struct.pack("<I", 36 + 8 + 6 + len(raw)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, 8000, 24000, 3, 24)
                + b"LIST" + struct.pack("<I", 5) + b"abcde\x00" + b"data" + struct.pack("<I", len(raw)) + raw)
    names = ["pcm16.wav", "f32.wav", "stereo.wav", "pcm32.wav", "u8.wav", "f64.wav", "three.wav", "pcm24.wav", "missing.wav"]
    paths = [p(n) for n in names]
    n, stride = len(paths), 5056
    out = torch.full((n, stride), float("nan"))
    fr, rt, st = (torch.zeros(n, dtype=torch.int32) for _ in range(3))
    rc = lib.wav_read_batch(paths, out.data_ptr(), stride, fr.data_ptr(), rt.data_ptr(), st.data_ptr(), 3)
    assert rc == -19 and st.tolist() == [0] * 8 + [-16] and fr[8] == 0
    for i, path in enumerate(paths[:-1]):
        ref, rate = read_wav(path)
        assert rt[i] == rate and fr[i] == len(ref), names[i]
        assert np.array_equal(out[i, :len(ref)].numpy(), ref), names[i]
        assert lib.wav_info(path)[:2] == (len(ref), rate)
    assert lib.wav_info(paths[2])[2:] == (2, -32) and lib.wav_info(paths[7])[2:] == (1, 24)
    rc = lib.wav_read_batch(paths[:1], out.data_ptr(), 100, fr.data_ptr(), rt.data_ptr(), st.data_ptr(), 1)       # the row is too short
    assert rc == 0 and st[0] == _abi.WAV_TRUNCATED and fr[0] == 100
    (tmp_path / "junk.wav").write_bytes(b"not a wave file at all")
    with pytest.raises(_abi.OnssenError, match="RIFF"):
        lib.wav_info(p("junk.wav"))


def test_host_side_of_the_training_batches(tmp_path):
    """The producer's half of a training batch without a GPU: signals (one file at 16 kHz: resampled like _load does), per-
    utterance sample counts and the seeded crops in file order -- the sequence the per-utterance loop of rounds 1-4 drew."""
    from onssen_amd.data.wsj0_2mix import _load
    lengths = [64 * 70 + 13, 64 * 25, 64 * 55 + 7]
    make_corpus(str(tmp_path), "tr", lengths)
    mix16 = synth_mixture(77, 9000, 16000)                        # utt01's mixture at another rate (s1 / s2 stay at 8 kHz)
    write_wav(os.path.join(str(tmp_path), "wav8k", "min", "tr", "mix", "utt01.wav"), mix16, 16000)
    fo = dict(FO, data_path=str(tmp_path))
    dl = Wsj02mixFiles("dc", fo, "tr", device="cpu", shuffle=False, seed=11)
    items = [(slot.wav.clone(), n_utt.copy(), starts.copy()) for slot, n_utt, starts in dl.host_batches()]
    assert [len(it[1]) for it in items] == [2, 1]
    rng = np.random.default_rng(11)
    k = 0
    for wav, n_utt, starts in items:
        for b in range(len(n_utt)):
            fn = dl.file_list[k]; k += 1
            sigs = [_load(f, 8000) for f in dl._sources(fn)]
            n = min(len(s) for s in sigs)
            assert n_utt[b] == n
            for j in range(3):
                np.testing.assert_array_equal(wav[3 * b + j, :n].numpy(), sigs[j][:n])
            T = 1 + n // 64
            Tr = T * (40 // T + 1) if T <= 40 else T
            assert starts[b] == int(rng.integers(0, Tr - 40))


def test_an_abandoned_epoch_does_not_strand_the_producer_or_perturb_the_next_epoch(tmp_path, monkeypatch):
    """the producer thread's sentinel / exception puts honour ``stop`` (a consumer that leaves the epoch early with a full
    queue must not leave a thread blocked in ``put``), the thread is joined, and because it draws from a child generator the next
    epoch's crop sequence does not depend on how far the abandoned one got.  Host side only (no GPU: ``_prefetched`` items)."""
    import threading
    monkeypatch.setenv("ONSSEN_LOADER_PREFETCH", "1")
    monkeypatch.setenv("ONSSEN_LOADER_WORKERS", "2")
    lengths = [64 * 50 + k for k in range(6)]
    make_corpus(str(tmp_path), "tr", lengths)
    fo = dict(FO, data_path=str(tmp_path), batch_size=1)

    def second_epoch_starts(abandon_after):
        dl = Wsj02mixFiles("dc", fo, "tr", device="cpu", shuffle=False, seed=5)
        order = np.arange(len(dl.file_list))
        it = dl._prefetched(order)
        for _ in range(abandon_after):
            next(it)
        it.close()                                   # the consumer walks away: queue full, producer mid-epoch
        assert not any(t.name == "onssen-wsj0-2mix-loader" and t.is_alive() for t in threading.enumerate())
        return [int(starts[0]) for _, _, starts in dl._prefetched(order)]
    a, b = second_epoch_starts(1), second_epoch_starts(4)
    assert a == b and len(a) == 6


def test_header_frame_counts_are_clamped_to_the_file(tmp_path):
    """a header that states more data than the file holds (streamed WAVs say 0xFFFFFFFF) must not size a multi-GB row."""
    import struct
    make_corpus(str(tmp_path), "tr", [64 * 30])
    fn = os.path.join(str(tmp_path), "wav8k", "min", "tr", "mix", "utt00.wav")
    raw = bytearray(open(fn, "rb").read())
    i = raw.find(b"data")
    raw[i + 4:i + 8] = struct.pack("<I", 0xFFFFFFFF)
    open(fn, "wb").write(raw)
    dl = Wsj02mixFiles("dc", dict(FO, data_path=str(tmp_path)), "tr", device="cpu", shuffle=False, seed=1)
    fr, rate = dl._frames_of(fn)
    assert rate == 8000 and 0 < fr <= os.path.getsize(fn) // 2
