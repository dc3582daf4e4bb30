"""Ragged batches of whole utterances (round 4; the shape the reference evaluates one utterance at a time,
onssen/utils/test.py:29-41 with the batch-1 loader of onssen/data/wsj0_2mix.py:231-245) through the host-side emulation
of the product's HIP source: every ``onssen_*_ragged_f32`` entry point must give, inside each row's own extent, BIT FOR BIT
what the uniform entry point gives for that utterance alone -- and must not let the padding (poisoned with NaNs here)
reach anything."""
import numpy as np
import pytest

from onssen_amd import _abi
from onssen_amd.synthetic import make_state_dict, synth_mixture
from oracle import np_oracle as O
from tests.emu_build import load_emu
from tests.test_emu_kernels import P, _shm, _two_cluster_embeddings, aligned_f32, rand


@pytest.fixture(scope="module")
def lib():
    return load_emu()


def i32(a):
    return np.ascontiguousarray(np.asarray(a, np.int32))


@pytest.mark.parametrize("n_fft,hop,ns", [(256, 64, [700, 515, 641, 129]), (512, 128, [1400, 900])])
def test_stft_logmag_ragged(lib, n_fft, hop, ns):
    B, n_max = len(ns), max(ns)
    T, F = 1 + n_max // hop, n_fft // 2 + 1
    wav = np.full((B, n_max), np.nan, np.float32)
    for b, n in enumerate(ns):
        wav[b, :n] = synth_mixture(5 + b, n)
    lm = np.full((B, T, F), np.nan, np.float32)
    ri = np.full((B, T, F, 2), np.nan, np.float32)
    ns_d = i32(ns)                                                     # (named: the array must outlive the call)
    lib.stft_logmag(P(wav), B, n_max, n_max, n_fft, hop, 1e-7, P(lm), P(ri), None, n_per_utt=P(ns_d))
    for b, n in enumerate(ns):
        Tb = 1 + n // hop
        one = np.ascontiguousarray(wav[b, :n])
        lm1 = np.full((1, Tb, F), np.nan, np.float32)
        ri1 = np.full((1, Tb, F, 2), np.nan, np.float32)
        lib.stft_logmag(P(one), 1, n, n, n_fft, hop, 1e-7, P(lm1), P(ri1), None)
        np.testing.assert_array_equal(lm[b, :Tb], lm1[0])
        np.testing.assert_array_equal(ri[b, :Tb], ri1[0])
        X = O.stft(one, n_fft, hop)                                    # ... and the oracle
        np.testing.assert_allclose(ri[b, :Tb, :, 0] + 1j * ri[b, :Tb, :, 1], X, atol=1e-6 * np.abs(X).max())
        # the frames past it: silence (to rounding: a silent frame that shares its complex transform with the row's last frame
        # gets that transform's ~1e-16 asymmetry)
        assert np.abs(ri[b, Tb:]).max(initial=0) < 1e-12 and np.allclose(lm[b, Tb:], -7.0, atol=1e-5)


@pytest.mark.parametrize("n_fft,hop", [(256, 64), (512, 128)])
def test_mask_istft_ragged(lib, n_fft, hop):
    rng = np.random.default_rng(6)
    ns = [2000, 1337, 1800]
    lens = [2016, 1344, 1700]                                        # sig_ref is padded to a multiple of 32 upstream; one row trimmed
    B, Cn, F = len(ns), 2, n_fft // 2 + 1
    specs = [O.stft(synth_mixture(20 + b, n), n_fft, hop) for b, n in enumerate(ns)]
    frames = [s.shape[0] for s in specs]
    T, length = max(frames), max(lens)
    ri = np.full((B, T, F, 2), np.nan, np.float32)
    masks = np.full((B, T, F, Cn), np.nan, np.float32)
    for b, s in enumerate(specs):
        ri[b, :frames[b]] = np.stack([s.real, s.imag], -1)
        masks[b, :frames[b]] = rng.random((frames[b], F, Cn))
    out = np.full((B, Cn, length), np.nan, np.float32)
    fr_d, len_d = i32(frames), i32(lens)
    lib.mask_istft(P(ri), P(masks), T * F * Cn, 1, F * Cn, Cn, B, Cn, T, n_fft, hop, length, P(out), None,
                   frames=P(fr_d), lengths=P(len_d))
    for b in range(B):
        ri1, m1 = np.ascontiguousarray(ri[b:b + 1, :frames[b]]), np.ascontiguousarray(masks[b:b + 1, :frames[b]])
        o1 = np.full((1, Cn, lens[b]), np.nan, np.float32)
        lib.mask_istft(P(ri1), P(m1), frames[b] * F * Cn, 1, F * Cn, Cn, 1, Cn, frames[b], n_fft, hop, lens[b], P(o1), None)
        np.testing.assert_array_equal(out[b, :, :lens[b]], o1[0])
        assert np.all(out[b, :, lens[b]:] == 0)
        ref = O.mask_istft(specs[b], m1[0].transpose(2, 0, 1), hop, lens[b])
        np.testing.assert_allclose(out[b, :, :lens[b]], ref, atol=2e-6)


@pytest.mark.parametrize("persistent", [True, False])
def test_dc_cluster_ragged(lib, monkeypatch, persistent):
    """Threshold + 2-means on a ragged batch: per utterance the same masks, bit for bit, as the uniform call on that utterance
    alone (same compaction, same initialisation, same summation orders); the padding -- NaN embeddings, a LOUDER feature than
    anything real -- neither becomes active nor moves the threshold."""
    monkeypatch.setenv("ONSSEN_EMU_FORK", "1" if persistent else "0")
    monkeypatch.setenv("ONSSEN_EMU_SCRAMBLE_XCC", "0")
    lib.dll.onssen_xcd_spin_limit(40000000)
    rng = np.random.default_rng(21)
    B, T, F, D = 2, 14, 33, 20
    frames = [14, 9]
    emb0, feat0, _ = _two_cluster_embeddings(rng, B, T, F, D)
    emb, feat = _shm(emb0.shape), _shm(feat0.shape)
    emb[...] = emb0; feat[...] = feat0
    for b in range(B):
        emb[b, frames[b]:] = np.nan
        feat[b, frames[b]:] = 50.0
    fr = _shm((B,), dtype=np.int32); fr[...] = frames
    nb = lib.dll.onssen_dc_cluster_workspace_bytes(B, T, F, D)
    ws = _shm((nb // 4 + 64,))
    masks = _shm((B, T, F, 2), fill=np.nan)
    flags = 0 if persistent else _abi.DC_CLUSTER_LAUNCH_PER_ITERATION
    lib.dc_cluster(P(emb), P(feat), B, T, F, D, 40.0, 6, P(masks), P(ws), nb, None, flags=flags, frames=P(fr))
    assert ws.view(np.uint32)[lib.dll.onssen_dc_cluster_status_offset(B, D) // 4] == 0
    for b in range(B):
        Tb = frames[b]
        e1, f1 = _shm((1, Tb, F, D)), _shm((1, Tb, F))
        e1[...] = emb0[b:b + 1, :Tb]; f1[...] = feat0[b:b + 1, :Tb]
        nb1 = lib.dll.onssen_dc_cluster_workspace_bytes(1, Tb, F, D)
        ws1 = _shm((nb1 // 4 + 64,))
        m1 = _shm((1, Tb, F, 2), fill=np.nan)
        lib.dc_cluster(P(e1), P(f1), 1, Tb, F, D, 40.0, 6, P(m1), P(ws1), nb1, None, flags=flags)
        np.testing.assert_array_equal(np.array(masks[b, :Tb]), np.array(m1[0]))
        assert np.all(np.array(masks[b, Tb:]) == 0)
        act = O.dc_active_bins(feat0[b, :Tb])
        assert np.all(np.array(masks[b, :Tb])[act].sum(-1) == 1) and np.all(np.array(masks[b, :Tb])[~act] == 0)


def test_batch_sdr_ragged(lib):
    rng = np.random.default_rng(21)
    B, C = 3, 2
    lens = [1500, 901, 1216]
    n = max(lens)
    org = np.full((B, C, n), np.nan, np.float32)
    est = np.full((B, C, n), np.nan, np.float32)
    for b, nb in enumerate(lens):
        org[b, :, :nb] = rand(rng, C, nb)
        est[b, :, :nb] = org[b, ::-1, :nb] * 0.7 + 0.4 * rand(rng, C, nb) + 0.1
    sdr = np.full(B, np.nan, np.float32); perm = np.full(B, -1, np.int32)
    ws = aligned_f32(lib.batch_sdr_workspace_bytes(B) // 4 + 64)
    len_d = i32(lens)
    lib.batch_sdr(P(est), P(org), None, B, C, n, P(sdr), P(perm), P(ws), ws.nbytes, None, lengths=P(len_d))
    for b, nb in enumerate(lens):
        e1, o1 = np.ascontiguousarray(est[b:b + 1, :, :nb]), np.ascontiguousarray(org[b:b + 1, :, :nb])
        s1 = np.full(1, np.nan, np.float32); p1 = np.full(1, -1, np.int32)
        lib.batch_sdr(P(e1), P(o1), None, 1, C, nb, P(s1), P(p1), P(ws), ws.nbytes, None)
        assert sdr[b] == s1[0] and perm[b] == p1[0]
        ref_sdr, ref_perm = O.batch_sdr(e1, o1)
        np.testing.assert_allclose(sdr[b], ref_sdr[0], rtol=1e-4, atol=1e-4)
        assert perm[b] == ref_perm[0]


def _pack_x3(lib, sd, F, H, L, ug):
    """Operands of the ONSSEN_BLSTM_XCD | ONSSEN_BLSTM_BF16X3 form in shared memory (as test_blstm_xcd_local_persistent)."""
    Hp, NP, KQ, we = lib.lstm_geometry(H, ug)
    _, _, we3 = lib.lstm_geometry_x3(H, ug)
    wih3, whh3, bias = [], [], []
    for l in range(L):
        K = F if l == 0 else 2 * Hp
        Kp = (F + 3) // 4 * 4 if l == 0 else 2 * Hp
        a, c, b3 = _shm((2, NP, Kp)), _shm((2, NP)), _shm((2, we3), dtype=np.uint16)
        scratch = _shm((we,))
        for d, sfx in enumerate(("", "_reverse")):
            srcs = []
            for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                v = sd[f"rnn.{n}_l{l}{sfx}"]
                sv = _shm(v.shape); sv[...] = v
                srcs.append(sv)
            lib.lstm_pack(P(srcs[0]), P(srcs[1]), P(srcs[2]), P(srcs[3]), srcs[0].shape[1], 0 if l == 0 else 1, H, ug,
                          P(a[d]), P(scratch), P(c[d]), None)
            lib.lstm_pack_whh_bf16x3(P(srcs[1]), H, ug, P(b3[d]), None)
        pl = _shm((2 * NP, (K + 31) // 32, 2, 32), dtype=np.uint16)
        lib.x3_image(P(a), Kp, 0, 1, 2 * NP, K, P(pl), None)
        wih3.append(pl), whh3.append(b3), bias.append(c)
    return Hp, NP, wih3, whh3, bias


@pytest.mark.parametrize("H,ug,frames", [(24, 8, [4, 2, 4, 1, 3, 4, 2, 3, 4]),   # 9 rows: three 4-row groups
                                         (12, 4, [3] * 9 + [2] * 9)])            # 18 rows: 8-row groups (stacked)
def test_blstm_ragged_persistent_is_the_batch_of_one_bit_for_bit(lib, monkeypatch, H, ug, frames):
    """onssen_blstm_forward_ragged_f32 in the XCD-local persistent form (forked workgroups): inside its own frames every row
    equals the uniform call on that utterance alone bit for bit (and the oracle to 2e-5); outside them the output is 0 --
    although the padding of the input holds NaNs."""
    monkeypatch.setenv("ONSSEN_EMU_FORK", "1")
    monkeypatch.setenv("ONSSEN_EMU_SCRAMBLE_XCC", "0")
    lib.dll.onssen_xcd_spin_limit(40000000)
    F, L = 9, 2
    B, T = len(frames), max(frames)
    sd = make_state_dict("chimera", F, H, L, 4, 2, seed=H + ug, gain=2.0)
    rng = np.random.default_rng(3)
    x0 = rand(rng, B, T, F)
    x = _shm((B, T, F), fill=np.nan)
    for b, Tb in enumerate(frames):
        x[b, :Tb] = x0[b, :Tb]
    Hp, NP, wih3, whh3, bias = _pack_x3(lib, sd, F, H, L, ug)
    fr = _shm((B,), dtype=np.int32); fr[...] = frames
    flags = _abi.BLSTM_BF16X3 | _abi.BLSTM_XCD
    ws = _shm((lib.blstm_workspace_bytes(B, T, F, H, L, ug) // 4 + 64,))
    y = _shm((T, B, 2, Hp), fill=np.nan)
    lib.blstm_forward(P(x), T * F, F, B, T, F, H, L, ug, [P(a) for a in wih3], [P(a) for a in whh3], [P(a) for a in bias],
                      P(y), P(ws), ws.nbytes, flags, None, frames=P(fr))
    status = ws.view(np.uint32)
    assert status[280] == 0 and status[282] == 0, f"aborted / non-finite ({status[280]}, {status[282]})"
    y = np.array(y)
    for b in sorted(set(range(B)) & {0, 1, 2, B - 1}):      # a few rows against their own batch-1 runs (each forks 8 NU processes)
        Tb = frames[b]
        x1 = _shm((1, Tb, F)); x1[...] = x0[b:b + 1, :Tb]
        ws1 = _shm((lib.blstm_workspace_bytes(1, Tb, F, H, L, ug) // 4 + 64,))
        y1 = _shm((Tb, 1, 2, Hp), fill=np.nan)
        lib.blstm_forward(P(x1), Tb * F, F, 1, Tb, F, H, L, ug, [P(a) for a in wih3], [P(a) for a in whh3], [P(a) for a in bias],
                          P(y1), P(ws1), ws1.nbytes, flags, None)
        np.testing.assert_array_equal(y[:Tb, b], np.array(y1)[:, 0])
    for b, Tb in enumerate(frames):
        ref = O.blstm_stack(x0[b:b + 1, :Tb], sd, "rnn.", L)[0]                      # (Tb, 2H)
        got = np.concatenate([y[:Tb, b, 0, :H], y[:Tb, b, 1, :H]], -1)
        assert np.abs(got - ref).max() < 2e-5
        assert np.all(y[Tb:, b] == 0)


@pytest.mark.parametrize("x3", [True, False])
def test_blstm_ragged_launch_per_step(lib, x3):
    """The launch-per-step recurrence (the re-run of an aborted call, ONSSEN_XCD=0, exact fp32 with H > 640) takes ragged
    batches too: split-bf16 and exact fp32, 20 rows (two row tiles)."""
    from tests.test_emu_kernels import _pack_lstm
    F, H, L, ug = 9, 12, 2, 4
    frames = [5, 2, 4, 1, 5, 3] + [4] * 14
    B, T = len(frames), max(frames)
    sd = make_state_dict("chimera", F, H, L, 4, 2, seed=7, gain=2.0)
    rng = np.random.default_rng(3)
    x0 = rand(rng, B, T, F)
    x = np.full((B, T, F), np.nan, np.float32)
    for b, Tb in enumerate(frames):
        x[b, :Tb] = x0[b, :Tb]
    Hp, NP, wih, whh, bias = _pack_lstm(lib, sd, "rnn.", F, H, L, ug)
    flags = 0
    if x3:
        _, _, we3 = lib.lstm_geometry_x3(H, ug)
        w3, h3 = [], []
        for l in range(L):
            K = F if l == 0 else 2 * Hp
            ld = (K + 31) // 32 * 32
            a3 = np.zeros((2, 2 * NP, ld), np.uint16)
            lib.linear_pack_bf16x3(P(wih[l]), 2 * NP, K, wih[l].shape[2], ld, P(a3), None)
            b3 = np.zeros((2, we3), np.uint16)
            for d, sfx in enumerate(("", "_reverse")):
                lib.lstm_pack_whh_bf16x3(P(np.ascontiguousarray(sd[f"rnn.weight_hh_l{l}{sfx}"])), H, ug, P(b3[d]), None)
            w3.append(a3), h3.append(b3)
        wih, whh, flags = w3, h3, _abi.BLSTM_BF16X3
    ws = aligned_f32(lib.blstm_workspace_bytes(B, T, F, H, L, ug) // 4 + 64)
    y = np.full((T, B, 2, Hp), np.nan, np.float32)
    fr_d = i32(frames)
    lib.blstm_forward(P(x), T * F, F, B, T, F, H, L, ug, [P(a) for a in wih], [P(a) for a in whh], [P(a) for a in bias],
                      P(y), P(ws), ws.nbytes, flags, None, frames=P(fr_d))
    for b, Tb in enumerate(frames):
        ref = O.blstm_stack(x0[b:b + 1, :Tb], sd, "rnn.", L)[0]
        got = np.concatenate([y[:Tb, b, 0, :H], y[:Tb, b, 1, :H]], -1)
        assert np.abs(got - ref).max() < (2e-5 if x3 else 2e-6)
        assert np.all(y[Tb:, b] == 0)
        if b < 3:
            x1 = np.ascontiguousarray(x0[b:b + 1, :Tb])
            ws1 = aligned_f32(lib.blstm_workspace_bytes(1, Tb, F, H, L, ug) // 4 + 64)
            y1 = np.full((Tb, 1, 2, Hp), np.nan, np.float32)
            lib.blstm_forward(P(x1), Tb * F, F, 1, Tb, F, H, L, ug, [P(a) for a in wih], [P(a) for a in whh], [P(a) for a in bias],
                              P(y1), P(ws1), ws1.nbytes, flags, None)
            np.testing.assert_array_equal(y[:Tb, b], y1[:, 0])


def test_ragged_forms_the_persistent_kernel_does_not_have_are_refused(lib):
    """ONSSEN_BLSTM_FUSE_IN0 / ONSSEN_BLSTM_BF16 / exact fp32 with ONSSEN_BLSTM_XCD have no ragged instantiation: the call
    returns ONSSEN_E_ARG instead of running the uniform kernel on a ragged batch."""
    fr = i32([2, 1])
    ws = aligned_f32(lib.blstm_workspace_bytes(2, 2, 9, 8, 1, 4) // 4 + 64)
    dummy = aligned_f32(4096)
    for flags in (_abi.BLSTM_XCD, _abi.BLSTM_XCD | _abi.BLSTM_BF16X3 | _abi.BLSTM_FUSE_IN0,
                  _abi.BLSTM_XCD | _abi.BLSTM_BF16X3 | _abi.BLSTM_BF16):
        with pytest.raises(_abi.OnssenError, match="code -1"):
            lib.blstm_forward(P(dummy), 18, 9, 2, 2, 9, 8, 1, 4, [P(dummy)], [P(dummy)], [P(dummy)], P(dummy), P(ws), ws.nbytes,
                              flags, None, frames=P(fr))


@pytest.mark.parametrize("ragged", [False, True])
def test_dc_compacted_pipeline_equals_the_full_embedding_pipeline(lib, monkeypatch, ragged):
    """Round 4, the embedding never round-trips HBM: onssen_dc_index_f32 (threshold -> target map) + onssen_linear_x3p_compact
    (the fc_dc GEMM stores only the active bins' rows, straight into the compacted array) + onssen_dc_cluster_compact_f32
    against onssen_linear_x3p(L2NORM) + onssen_dc_cluster_f32: the same compacted rows and the same masks, bit for bit."""
    monkeypatch.setenv("ONSSEN_EMU_FORK", "1")
    monkeypatch.setenv("ONSSEN_EMU_SCRAMBLE_XCC", "0")
    lib.dll.onssen_xcd_spin_limit(40000000)
    rng = np.random.default_rng(5)
    B, T, F, D, K = 2, 11, 33, 20, 40                  # (every emulated utterance costs 8 forked workgroups of 512 threads, twice)
    N, M = F * D, T * B
    frames = [11, 7] if ragged else None
    # a head whose embeddings fall into two clusters: W maps two planted directions of the activations to two centroids
    lab = rng.integers(0, 2, (B, T, F))
    cen = rand(rng, 2, D)
    x = _shm((B, T, K)); x[...] = 0.2 * rand(rng, B, T, K)
    x[..., 0] = 1.0
    W = _shm((N, K)); W[...] = 0.05 * rand(rng, N, K)
    bias = _shm((N,)); bias[...] = 0
    # per-bin offset along the constant input column: bins of frame-independent "speaker" pattern (varies with f only) ...
    pat = rng.integers(0, 2, F)
    W[:, 0] = cen[pat].reshape(-1)
    feat = _shm((B, T, F)); feat[...] = (-3.0 + 2.9 * rng.random((B, T, F))).astype(np.float32)
    feat[:, :, ::5] = -6.0                                         # silent bins
    fr = None
    if ragged:
        fr = _shm((B,), dtype=np.int32); fr[...] = frames
        for b in range(B):
            feat[b, frames[b]:] = 30.0                             # louder than anything real: must not matter
    KB = (K + 31) // 32
    a_img, w_img = _shm((M, KB, 2, 32), dtype=np.uint16), _shm((N, KB, 2, 32), dtype=np.uint16)
    lib.x3_image(P(x), K, T * K, B, M, K, P(a_img), None)
    lib.x3_image(P(W), K, 0, 1, N, K, P(w_img), None)
    # (A) the full embedding, then the round-3 back end
    emb = _shm((B, T, F, D), fill=np.nan)
    lib.linear_x3p(P(a_img), M, K, P(w_img), P(bias), N, _abi.EPI_L2NORM, D, 1e-12, P(emb), B, N, T * N, None)
    nbA = lib.dll.onssen_dc_cluster_workspace_bytes(B, T, F, D)
    wsA = _shm((nbA // 4 + 64,))
    mA = _shm((B, T, F, 2), fill=np.nan)
    lib.dc_cluster(P(emb), P(feat), B, T, F, D, 40.0, 8, P(mA), P(wsA), nbA, None, frames=P(fr) if ragged else None)
    # (B) index -> compacting GEMM -> cluster
    nbB, comp_off, dest_off = lib.dc_compact_layout(B, T, F, D)
    wsB = _shm((nbB // 4 + 64,))
    wsB.view(np.uint32)[comp_off // 4:] = 0x7fc00000            # NaNs where nothing may be read before it is written
    mB = _shm((B, T, F, 2), fill=np.nan)
    base = wsB.ctypes.data
    lib.dc_index(P(feat), B, T, F, D, 40.0, base, nbB, None, frames=P(fr) if ragged else None)
    dest = wsB.view(np.int32)[dest_off // 4:dest_off // 4 + B * T * F].reshape(B, T * F)
    for b in range(B):
        Tb = frames[b] if ragged else T
        act = np.zeros((T, F), bool)
        act[:Tb] = O.dc_active_bins(np.array(feat[b, :Tb]))
        want = np.where(act.reshape(-1), np.cumsum(act.reshape(-1)) - 1, -1)
        np.testing.assert_array_equal(dest[b], want)
    lib.linear_x3p_compact(P(a_img), M, K, P(w_img), P(bias), N, D, 1e-12, base + dest_off, T * F, F, base + comp_off, B,
                           T * F * D, False, None)
    comp = wsB[comp_off // 4:comp_off // 4 + B * T * F * D].reshape(B, T * F, D)
    for b in range(B):
        n = int((dest[b] >= 0).sum())
        np.testing.assert_array_equal(comp[b, :n], np.array(emb[b]).reshape(-1, D)[dest[b] >= 0])      # same bits, compacted
    lib.dc_cluster_compact(B, T, F, D, 8, P(mB), base, nbB, None)
    assert wsB.view(np.uint32)[lib.dll.onssen_dc_cluster_status_offset(B, D) // 4] == 0
    np.testing.assert_array_equal(np.array(mB), np.array(mA))
    assert np.array(mB)[..., 0].sum() > 0 and np.array(mB)[..., 1].sum() > 0         # both clusters used
