"""Host-side kernel tests: the product's HIP source compiled against the mock
HIP runtime (tests/emu) and driven through the same C ABI / ctypes binding as
the GPU library, checked against the NumPy oracle.  Sizes are tiny (every
work-item is an OS thread)."""
import os

import numpy as np
import pytest

from onssen_amd import _abi
from onssen_amd.synthetic import make_state_dict, synth_mixture
from oracle import np_oracle as O
from tests.emu_build import load_emu


@pytest.fixture(scope="module")
def lib():
    return load_emu()


def P(a):
    return a.ctypes.data if a is not None else None


def aligned_f32(n, align=256):
    """float32 scratch whose address is `align`-byte aligned (workspaces must be 256-byte aligned)."""
    raw = np.zeros(n + align // 4, np.float32)
    off = (-raw.ctypes.data % align) // 4
    return raw[off:off + n]


def rand(rng, *shape):
    return rng.standard_normal(shape).astype(np.float32)


@pytest.mark.parametrize("mode,group", [(_abi.EPI_BIAS, 0), (_abi.EPI_L2NORM, 20), (_abi.EPI_SIGMOID, 0),
                                        (_abi.EPI_L2NORM, 2)])
def test_linear_epilogues_and_row_maps(lib, mode, group):
    rng = np.random.default_rng(1)
    Bb, Tt, K, N = 3, 47, 37, 100          # M = 141 (2 row blocks, ragged), N = 100 (2 col blocks, ragged)
    x = rand(rng, Bb, Tt, K)               # batch-major A, odd K -> scalar load path
    ldw = 40
    W = np.zeros((N, ldw), np.float32)
    W[:, :K] = rand(rng, N, K)
    bias = rand(rng, N)
    resid = rand(rng, Bb, Tt, N) if group == 2 else None
    out = np.full((Bb, Tt, N), np.nan, np.float32)
    # logical rows are time-major (m = t*B + b): A and C live batch-major
    lib.linear(P(x), Tt * 0 + K, Tt * K, Bb, Tt * Bb, K, P(W), ldw, P(bias), N, mode, group, 1e-12,
               P(resid), P(out), N, Tt * N, None)
    ref = x.astype(np.float64) @ W[:, :K].T.astype(np.float64) + bias
    if mode == _abi.EPI_L2NORM:
        if resid is not None:
            ref = ref + resid
        r = ref.reshape(Bb, Tt, N // group, group)
        ref = (r / np.maximum(np.linalg.norm(r, axis=-1, keepdims=True), 1e-12)).reshape(Bb, Tt, N)
    elif mode == _abi.EPI_SIGMOID:
        ref = 1 / (1 + np.exp(-ref))
    assert not np.isnan(out).any()
    np.testing.assert_allclose(out, ref, atol=2e-5)


def test_linear_vector_path_time_major(lib):
    rng = np.random.default_rng(2)
    M, K, N = 130, 48, 160
    A = rand(rng, M, K)
    W = rand(rng, N, K)
    bias = rand(rng, N)
    out = np.zeros((M, N), np.float32)
    lib.linear(P(A), 4 * K, K, 4, M, K, P(W), K, P(bias), N, _abi.EPI_BIAS, 0, 0.0, None, P(out), 4 * N, N, None)
    np.testing.assert_allclose(out, A.astype(np.float64) @ W.T + bias, atol=2e-5)


def _pack_lstm(lib, sd, prefix, in_dim, H, L, ug):
    Hp, NP, KQ, we = lib.lstm_geometry(H, ug)
    wih, whh, bias = [], [], []
    for l in range(L):
        bidir = 0 if l == 0 else 1
        Kp = (in_dim + 3) // 4 * 4 if l == 0 else 2 * Hp
        a = np.zeros((2, NP, Kp), np.float32)
        b = np.zeros((2, we), np.float32)
        c = np.zeros((2, NP), np.float32)
        for d, sfx in enumerate(("", "_reverse")):
            g = lambda n: np.ascontiguousarray(sd[f"{prefix}{n}_l{l}{sfx}"])
            w_ih, w_hh, b_ih, b_hh = g("weight_ih"), g("weight_hh"), g("bias_ih"), g("bias_hh")
            lib.lstm_pack(P(w_ih), P(w_hh), P(b_ih), P(b_hh), w_ih.shape[1], bidir, H, ug, P(a[d]), P(b[d]),
                          P(c[d]), None)
        wih.append(a); whh.append(b); bias.append(c)
    return Hp, NP, wih, whh, bias


@pytest.mark.parametrize("H,ug,B", [(8, 8, 3), (12, 8, 2), (12, 12, 17), (20, 20, 2), (8, 4, 2)])
def test_blstm_stack_matches_oracle(lib, H, ug, B):
    F, L, T = 9, 2, 4
    sd = make_state_dict("chimera", F, H, L, 4, 2, seed=H + ug, gain=2.0)
    rng = np.random.default_rng(3)
    x = rand(rng, B, T, F)
    Hp, NP, wih, whh, bias = _pack_lstm(lib, sd, "rnn.", F, H, L, ug)
    ws = aligned_f32(lib.blstm_workspace_bytes(B, T, F, H, L, ug) // 4 + 64)
    y = np.full((T, B, 2, Hp), np.nan, np.float32)
    lib.blstm_forward(P(x), T * F, F, B, T, F, H, L, ug, [P(a) for a in wih], [P(a) for a in whh],
                      [P(a) for a in bias], P(y), P(ws), ws.nbytes, 0, None)
    ref = O.blstm_stack(x, sd, "rnn.", L)                      # (B, T, 2H)
    got = np.concatenate([y[:, :, 0, :H], y[:, :, 1, :H]], -1).transpose(1, 0, 2)
    np.testing.assert_allclose(got, ref, atol=2e-6)
    assert np.all(y[:, :, :, H:] == 0)                         # padded units stay exactly zero


def test_head_pack_folds_batchnorm(lib):
    H, Hp, N = 6, 8, 10
    rng = np.random.default_rng(4)
    w, b = rand(rng, N, 2 * H), rand(rng, N)
    g, beta, mean = rand(rng, 2 * H), rand(rng, 2 * H), rand(rng, 2 * H)
    var = rng.uniform(0.1, 1, 2 * H).astype(np.float32)
    wp, bp = np.full((N, 2 * Hp), np.nan, np.float32), np.zeros(N, np.float32)
    lib.head_pack(P(w), P(b), N, H, Hp, P(g), P(beta), P(mean), P(var), 1e-5, P(wp), P(bp), None)
    r = rand(rng, 5, 2 * H)
    ref = ((r - mean) / np.sqrt(var + 1e-5) * g + beta) @ w.T + b
    rp = np.zeros((5, 2 * Hp), np.float32)
    rp[:, :H], rp[:, Hp:Hp + H] = r[:, :H], r[:, H:]
    np.testing.assert_allclose(rp @ wp.T + bp, ref, atol=1e-5)
    wp2, bp2 = np.zeros_like(wp), np.zeros_like(bp)
    lib.head_pack(P(w), P(b), N, H, Hp, None, None, None, None, 0.0, P(wp2), P(bp2), None)
    np.testing.assert_array_equal(bp2, b)
    np.testing.assert_array_equal(wp2[:, :H], w[:, :H])


@pytest.mark.parametrize("n_fft,hop,n", [(256, 64, 700), (512, 128, 1400)])
def test_stft_logmag_matches_oracle(lib, n_fft, hop, n):
    B = 2
    wav = np.stack([synth_mixture(5 + b, n) for b in range(B)])
    T, F = 1 + n // hop, n_fft // 2 + 1
    lm = np.full((B, T, F), np.nan, np.float32)
    ri = np.full((B, T, F, 2), np.nan, np.float32)
    lib.stft_logmag(P(wav), B, n, n, n_fft, hop, 1e-7, P(lm), P(ri), None)
    for b in range(B):
        X = O.stft(wav[b], n_fft, hop)
        np.testing.assert_allclose(ri[b, ..., 0] + 1j * ri[b, ..., 1], X, atol=1e-6 * np.abs(X).max())
        np.testing.assert_allclose(10.0 ** lm[b].astype(np.float64), np.abs(X) + 1e-7, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("n_fft,hop,n,length", [(256, 64, 2000, 2000), (256, 64, 2000, 2300), (512, 128, 3000, 2900),
                                                (256, 100, 1500, 1500)])
def test_mask_istft_matches_oracle(lib, n_fft, hop, n, length):
    rng = np.random.default_rng(6)
    B, Cn = 2, 2
    F = n_fft // 2 + 1
    specs = [O.stft(synth_mixture(20 + b, n), n_fft, hop) for b in range(B)]
    T = specs[0].shape[0]
    ri = np.stack([np.stack([s.real, s.imag], -1) for s in specs]).astype(np.float32)
    masks = rng.random((B, T, F, Cn)).astype(np.float32)       # interleaved like fc_mi's output
    out = np.full((B, Cn, length), np.nan, np.float32)
    lib.mask_istft(P(ri), P(masks), T * F * Cn, 1, F * Cn, Cn, B, Cn, T, n_fft, hop, length, P(out), None)
    for b in range(B):
        ref = O.mask_istft(specs[b], masks[b].transpose(2, 0, 1), hop, length)
        np.testing.assert_allclose(out[b], ref, atol=2e-6)
    out1 = np.zeros((B, 1, length), np.float32)
    lib.mask_istft(P(ri), None, 0, 0, 0, 0, B, 1, T, n_fft, hop, length, P(out1), None)
    np.testing.assert_allclose(out1[0, 0], O.istft(specs[0], hop, length), atol=2e-6)
    if hop == 64 and length == 2000:       # odd speaker count: speakers go through the FFT in pairs, the last one alone
        m3 = rng.random((B, 3, T, F)).astype(np.float32)
        out3 = np.full((B, 3, length), np.nan, np.float32)
        lib.mask_istft(P(ri), P(m3), 3 * T * F, T * F, F, 1, B, 3, T, n_fft, hop, length, P(out3), None)
        np.testing.assert_allclose(out3[1], O.mask_istft(specs[1], m3[1], hop, length), atol=2e-6)


def _shm(shape, fill=0.0, dtype=np.float32):
    """'Device' buffer in MAP_SHARED memory: visible to the forked workgroup processes."""
    import mmap
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    a = np.frombuffer(mmap.mmap(-1, max(n, 16)), dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    a[...] = fill
    return a


@pytest.mark.parametrize("H,ug,B", [(40, 8, 3), (12, 12, 17), (70, 4, 2)])
def test_blstm_split_bf16_recurrence(lib, H, ug, B):
    """ONSSEN_BLSTM_BF16X3: h W_hh^T as three bf16 MFMAs per fp32 product; ~1e-5 of the fp32 oracle."""
    F, L, T = 9, 2, 6
    sd = make_state_dict("chimera", F, H, L, 4, 2, seed=H + ug, gain=2.0)
    rng = np.random.default_rng(3)
    x = rand(rng, B, T, F)
    Hp, NP, wih, whh, bias = _pack_lstm(lib, sd, "rnn.", F, H, L, ug)
    _, _, we3 = lib.lstm_geometry_x3(H, ug)
    whh3, wih3 = [], []
    for l in range(L):
        b3 = np.zeros((2, we3), np.uint16)
        for d, sfx in enumerate(("", "_reverse")):
            w_hh = np.ascontiguousarray(sd[f"rnn.weight_hh_l{l}{sfx}"])
            lib.lstm_pack_whh_bf16x3(P(w_hh), H, ug, P(b3[d]), None)
        whh3.append(b3)
        K = F if l == 0 else 2 * Hp
        ld = (K + 31) // 32 * 32
        pl = np.zeros((2, 2 * NP, ld), np.uint16)           # hi plane, lo plane of the [2*NP][K] projection matrix
        lib.linear_pack_bf16x3(P(wih[l]), 2 * NP, K, wih[l].shape[-1], ld, P(pl), None)
        wih3.append(pl)
    ws = aligned_f32(lib.blstm_workspace_bytes(B, T, F, H, L, ug) // 4 + 64)
    y = np.full((T, B, 2, Hp), np.nan, np.float32)
    lib.blstm_forward(P(x), T * F, F, B, T, F, H, L, ug, [P(a) for a in wih3], [P(a) for a in whh3],
                      [P(a) for a in bias], P(y), P(ws), ws.nbytes, _abi.BLSTM_BF16X3, None)
    ref = O.blstm_stack(x, sd, "rnn.", L)
    got = np.concatenate([y[:, :, 0, :H], y[:, :, 1, :H]], -1).transpose(1, 0, 2)
    err = np.abs(got - ref).max()
    assert err < 2e-5, err
    assert np.all(y[:, :, :, H:] == 0)


@pytest.mark.parametrize("mode,group,K", [(_abi.EPI_BIAS, 0, 37), (_abi.EPI_L2NORM, 20, 64), (_abi.EPI_SIGMOID, 0, 40),
                                          (_abi.EPI_L2NORM, 2, 33)])
def test_linear_split_bf16(lib, mode, group, K):
    """onssen_linear_bf16x3 (256x160x32 tile): ragged M/N/K, both A load paths, all epilogues."""
    rng = np.random.default_rng(7)
    Bb, Tt, N = 3, 91, 180           # M = 273 (2 row blocks), N = 180 (2 column blocks)
    x = rand(rng, Bb, Tt, K)
    W = rand(rng, N, K)
    bias = rand(rng, N)
    ld = (K + 31) // 32 * 32
    planes = np.zeros((2, N, ld), np.uint16)
    lib.linear_pack_bf16x3(P(W), N, K, K, ld, P(planes), None)
    hi = (planes[0].astype(np.uint32) << 16).view(np.float32)
    lo = (planes[1].astype(np.uint32) << 16).view(np.float32)
    assert np.abs(hi[:, :K] + lo[:, :K] - W).max() <= 2 ** -16 * np.abs(W).max() and not planes[:, :, K:].any()
    resid = rand(rng, Bb, Tt, N) if group == 2 else None
    out = np.full((Bb, Tt, N), np.nan, np.float32)
    lib.linear_bf16x3(P(x), K, Tt * K, Bb, Tt * Bb, K, P(planes), ld, P(bias), N, mode, group, 1e-12, P(resid), P(out),
                      N, Tt * N, None)
    ref = x.astype(np.float64) @ W.T.astype(np.float64) + bias
    if mode == _abi.EPI_L2NORM:
        if resid is not None:
            ref = ref + resid
        r = ref.reshape(Bb, Tt, N // group, group)
        ref = (r / np.maximum(np.linalg.norm(r, axis=-1, keepdims=True), 1e-12)).reshape(Bb, Tt, N)
    elif mode == _abi.EPI_SIGMOID:
        ref = 1 / (1 + np.exp(-ref))
    assert not np.isnan(out).any()
    # |x|,|w| ~ N(0,1): ~1e-5 relative of sqrt(K)-sized sums; normalising a short 2-vector amplifies it
    np.testing.assert_allclose(out, ref, atol=2e-3 if group == 2 else 3e-4, rtol=1e-4)


def _half_grid(rows, cols):
    """The checkerboard half of rows x cols (every row value and every column value still occurs, every pair of the full grid
    differs from a kept one in at most one factor); ONSSEN_EMU_FULL=1 runs the whole grid.  The host-side emulation runs every
    work-item as an OS thread: the full grids of the two widest kernels cost a minute and a half of the CPU suite, and the GPU
    suite runs the same kernels at full size."""
    import os
    full = os.environ.get("ONSSEN_EMU_FULL") == "1"
    return [r + c for i, r in enumerate(rows) for j, c in enumerate(cols) if full or (i + j) % 2 == 0]


@pytest.mark.parametrize("tile,mode,group,K", _half_grid(
    [("0",), ("256",), ("320",), ("256/128",), ("320/128",)],     # kernel / width[/height] of the x3q tile
    [(_abi.EPI_BIAS, 0, 37), (_abi.EPI_L2NORM, 20, 64), (_abi.EPI_SIGMOID, 0, 40), (_abi.EPI_L2NORM, 40, 33)]))
def test_linear_x3_images(lib, mode, group, K, tile, monkeypatch):
    """onssen_x3_image_f32 + onssen_linear_x3p (pre-split operands; 256x160 register-staged tile, 256x256 / 256x320
    LDS-DMA tiles; register epilogue): ragged M/N/K, strided A rows and C rows, all epilogues."""
    monkeypatch.setenv("ONSSEN_X3Q", tile.split("/")[0])
    monkeypatch.setenv("ONSSEN_X3Q_BM", tile.split("/")[1] if "/" in tile else "256")
    rng = np.random.default_rng(8)
    Bb, Tt, N = 3, 91, 440           # M = 273 (2 row blocks), N = 440 (3 / 2 / 2 column blocks, the last one ragged)
    x = rand(rng, Bb, Tt, K)
    W = rand(rng, N, K)
    bias = rand(rng, N)
    KB = (K + 31) // 32
    M = Bb * Tt
    a_img = np.full((M, KB, 2, 32), 0x7fc0, np.uint16)     # NaN patterns: the kernel must overwrite all of it
    w_img = np.full((N, KB, 2, 32), 0x7fc0, np.uint16)
    # activations time-major like the recurrent stack's rows (m = t*B + b) read from a (B, T, K) tensor
    lib.x3_image(P(x), K, Tt * K, Bb, M, K, P(a_img), None)
    lib.x3_image(P(W), K, 0, 1, N, K, P(w_img), None)
    f = lambda u: (u.astype(np.uint32) << 16).view(np.float32)
    got_w = (f(w_img[:, :, 0]) + f(w_img[:, :, 1])).reshape(N, KB * 32)
    assert np.abs(got_w[:, :K] - W).max() <= 2 ** -16 * np.abs(W).max() and not got_w[:, K:].any()
    xt = x.transpose(1, 0, 2).reshape(M, K)
    got_a = (f(a_img[:, :, 0]) + f(a_img[:, :, 1])).reshape(M, KB * 32)
    assert np.abs(got_a[:, :K] - xt).max() <= 2 ** -16 * np.abs(xt).max()
    out = np.full((Bb, Tt, N), np.nan, np.float32)
    lib.linear_x3p(P(a_img), M, K, P(w_img), P(bias), N, mode, group, 1e-12, P(out), Bb, N, Tt * N, None)
    ref = x.astype(np.float64) @ W.T.astype(np.float64) + bias
    if mode == _abi.EPI_L2NORM:
        r = ref.reshape(Bb, Tt, N // group, group)
        ref = (r / np.maximum(np.linalg.norm(r, axis=-1, keepdims=True), 1e-12)).reshape(Bb, Tt, N)
    elif mode == _abi.EPI_SIGMOID:
        ref = 1 / (1 + np.exp(-ref))
    assert not np.isnan(out).any()
    np.testing.assert_allclose(out, ref, atol=3e-4, rtol=1e-4)


@pytest.mark.parametrize("tile,K,bf16", [("320", 70, False), ("256", 37, False), ("320/128", 33, False), ("256/128", 96, False), ("320", 40, True)])
def test_linear_x3_images_bias_mode_on_32x32_tiles(lib, tile, K, bf16, monkeypatch):
    """linear_x3r_kernel (round 6c): onssen_linear_x3p's bias mode on v_mfma_f32_32x32x16_bf16 tiles (ONSSEN_X3R=1), all four tile
    shapes, both term counts: ragged M / N / K, strided C rows, against fp64 and -- to the last-bit budget of a re-ordered fp32 sum --
    against the 16 x 16 kernel."""
    monkeypatch.setenv("ONSSEN_X3Q", tile.split("/")[0])
    monkeypatch.setenv("ONSSEN_X3Q_BM", tile.split("/")[1] if "/" in tile else "256")
    rng = np.random.default_rng(18)
    Bb, Tt, N = 3, 91, 440
    x, W, bias = rand(rng, Bb, Tt, K), rand(rng, N, K), rand(rng, N)
    KB, M = (K + 31) // 32, Bb * Tt
    a_img = np.full((M, KB, 2, 32), 0x7fc0, np.uint16)
    w_img = np.full((N, KB, 2, 32), 0x7fc0, np.uint16)
    lib.x3_image(P(x), K, Tt * K, Bb, M, K, P(a_img), None)
    lib.x3_image(P(W), K, 0, 1, N, K, P(w_img), None)
    mode = _abi.EPI_BIAS | (_abi.EPI_BF16 if bf16 else 0)
    outs = {}
    for r in ("1", "0"):
        monkeypatch.setenv("ONSSEN_X3R", r)
        out = np.full((Bb, Tt, N + 4), np.nan, np.float32)               # (C rows 4 floats apart from dense: strides, not a contiguous block)
        lib.linear_x3p(P(a_img), M, K, P(w_img), P(bias), N, mode, 0, 1e-12, P(out), Bb, N + 4, Tt * (N + 4), None)
        assert not np.isnan(out[:, :, :N]).any() and np.isnan(out[:, :, N:]).all()
        outs[r] = out[:, :, :N]
    ref = x.astype(np.float64) @ W.T.astype(np.float64) + bias
    if bf16:
        xr, Wr = O.bf16_round(x), O.bf16_round(W)
        np.testing.assert_allclose(outs["1"], xr.astype(np.float64) @ Wr.T.astype(np.float64) + bias, atol=2e-5, rtol=1e-5)
    else:
        np.testing.assert_allclose(outs["1"], ref, atol=3e-4, rtol=1e-4)
    np.testing.assert_allclose(outs["1"], outs["0"], atol=2e-5, rtol=2e-6)


@pytest.mark.parametrize("group,N,tile", [(2, 514, "256"), (2, 38, "128"), (20, 440, "256")])
def test_linear_x3_images_with_residual(lib, group, N, tile, monkeypatch):
    """onssen_linear_x3p_resid: phase_net's head epilogue (onssen/nn/phase_network.py:58-66) on the pre-split-operand GEMM --
    (A W^T + b + residual) normalised over pairs (re, im) or wider groups; rows b and b + R of a time step add the SAME
    residual row (both speakers of the shared phase BLSTM in one launch); ragged N (514 = 2 F), strided C rows."""
    monkeypatch.setenv("ONSSEN_X3Q_BM", tile)
    rng = np.random.default_rng(9)
    R, Tt, K = 3, 47, 75              # 2 R = 6 batch rows per time step: M = 282 (two 256-row blocks / three 128-row ones)
    Bb = 2 * R
    x = rand(rng, Bb, Tt, K)
    W, bias = rand(rng, N, K), rand(rng, N)
    resid = rand(rng, R, Tt, N)
    KB, M = (K + 31) // 32, Bb * Tt
    a_img, w_img = np.zeros((M, KB, 2, 32), np.uint16), np.zeros((N, KB, 2, 32), np.uint16)
    lib.x3_image(P(x), K, Tt * K, Bb, M, K, P(a_img), None)          # rows m = t*Bb + b, like the recurrence's output image
    lib.x3_image(P(W), K, 0, 1, N, K, P(w_img), None)
    out = np.full((Bb, Tt, N), np.nan, np.float32)
    lib.linear_x3p_resid(P(a_img), M, K, P(w_img), P(bias), N, group, 1e-12, P(resid), R, P(out), Bb, N, Tt * N, False, None)
    ref = x.astype(np.float64) @ W.T.astype(np.float64) + bias + np.concatenate([resid, resid], 0)
    r = ref.reshape(Bb, Tt, N // group, group)
    nrm = np.maximum(np.linalg.norm(r, axis=-1, keepdims=True), 1e-12)
    assert not np.isnan(out).any()
    # a direction is only as well defined as its vector is long: the GEMM's own error (3e-4 on values of order 10) over the norm
    err = np.abs(out.reshape(r.shape) - r / nrm) * nrm
    assert err.max() <= 3e-4, err.max()
    np.testing.assert_allclose(np.linalg.norm(out.reshape(Bb, Tt, N // group, group), axis=-1), 1.0, atol=1e-5)
    # the plain entry keeps refusing pairs (they exist in the residual entry's kernel only)
    assert lib.dll.onssen_linear_x3p(P(a_img), M, K, P(w_img), P(bias), N, _abi.EPI_L2NORM, 2, 1e-12, P(out), Bb, N, Tt * N, None) != 0 \
        if group == 2 else True


@pytest.mark.parametrize("tile_rows,bf16", [(60, False), (300, False), (60, True)])
def test_linear_x3_images_two_heads_in_one_launch(lib, tile_rows, bf16):
    """onssen_linear_x3p_pair (chimera: fc_dc + L2 norm over D | fc_mi + sigmoid, onssen/nn/chimera.py:37-45): the rows of both
    layers form ONE B operand; columns < n_split are normalised per group into C, the rest pass through the logistic into
    C2 -- bit for bit what the two separate launches give."""
    rng = np.random.default_rng(10)
    Bb, Tt, K, group = 2, tile_rows // 2, 75, 20
    Na, Nb = 13 * group, 26                       # 260 + 26 = 286 columns: one 320-wide tile, the split inside its first wave tile
    x = rand(rng, Bb, Tt, K)
    Wa, Wb, ba, bb = rand(rng, Na, K), rand(rng, Nb, K), rand(rng, Na), rand(rng, Nb)
    KB, M = (K + 31) // 32, Bb * Tt
    a_img = np.zeros((M, KB, 2, 32), np.uint16)
    lib.x3_image(P(x), K, Tt * K, Bb, M, K, P(a_img), None)
    wa_img, wb_img = np.zeros((Na, KB, 2, 32), np.uint16), np.zeros((Nb, KB, 2, 32), np.uint16)
    lib.x3_image(P(Wa), K, 0, 1, Na, K, P(wa_img), None)
    lib.x3_image(P(Wb), K, 0, 1, Nb, K, P(wb_img), None)
    w_img, bias = np.ascontiguousarray(np.concatenate([wa_img, wb_img], 0)), np.concatenate([ba, bb])
    out_a, out_b = np.full((Bb, Tt, Na), np.nan, np.float32), np.full((Bb, Tt, Nb), np.nan, np.float32)
    lib.linear_x3p_pair(P(a_img), M, K, P(w_img), P(bias), Na + Nb, Na, group, 1e-12, P(out_a), Bb, Na, Tt * Na, P(out_b), Nb, Tt * Nb,
                        bf16, None)
    ref_a, ref_b = np.full_like(out_a, np.nan), np.full_like(out_b, np.nan)
    fl = _abi.EPI_BF16 if bf16 else 0              # (opt-in plain-bf16 products: the hi halves of the images only)
    lib.linear_x3p(P(a_img), M, K, P(wa_img), P(ba), Na, _abi.EPI_L2NORM | fl, group, 1e-12, P(ref_a), Bb, Na, Tt * Na, None)
    lib.linear_x3p(P(a_img), M, K, P(wb_img), P(bb), Nb, _abi.EPI_SIGMOID | fl, 0, 0.0, P(ref_b), Bb, Nb, Tt * Nb, None)
    assert not np.isnan(out_a).any() and not np.isnan(out_b).any()
    np.testing.assert_array_equal(out_a, ref_a)
    np.testing.assert_array_equal(out_b, ref_b)
    full = x.astype(np.float64) @ Wb.T.astype(np.float64) + bb
    np.testing.assert_allclose(out_b, 1 / (1 + np.exp(-full)), atol=2e-2 if bf16 else 1e-4)


@pytest.mark.parametrize("K,M", [(70, 150), (64, 64), (33, 97)])
def test_x3_image_both_equals_the_two_single_image_kernels(lib, K, M):
    """onssen_x3_image_both_f32: the row-major and the transposed x3 image of a [K][M] matrix from one pass -- the same bits as
    onssen_x3_image_f32 and onssen_x3_image_t_f32 (k_shift 0), padding included."""
    rng = np.random.default_rng(K + M)
    ld = M + 3
    src = rand(rng, K, ld)
    KB, MB = (K + 31) // 32, (M + 31) // 32
    rows_a, t_a = np.full((K, MB, 2, 32), 0x7fc0, np.uint16), np.full((M, KB, 2, 32), 0x7fc0, np.uint16)
    rows_b, t_b = np.full_like(rows_a, 0x7fc0), np.full_like(t_a, 0x7fc0)
    lib.x3_image_both(P(src), ld, M, K, P(rows_a), P(t_a), None)
    lib.x3_image(P(src), ld, 0, 1, K, M, P(rows_b), None)
    lib.x3_image_t(P(src), ld, M, K, 0, P(t_b), None)
    np.testing.assert_array_equal(rows_a, rows_b)
    np.testing.assert_array_equal(t_a, t_b)
    # ... and with the column sums of every 32-row block (the bias gradient's operand): same images, sums to fp32 rounding
    rows_c, t_c, part = np.full_like(rows_a, 0x7fc0), np.full_like(t_a, 0x7fc0), np.full((KB, M), np.nan, np.float32)
    lib.x3_image_both_colsum(P(src), ld, M, K, P(rows_c), P(t_c), P(part), None)
    np.testing.assert_array_equal(rows_c, rows_a)
    np.testing.assert_array_equal(t_c, t_a)
    blocks = np.stack([src[32 * kb:32 * kb + 32, :M].astype(np.float64).sum(0) for kb in range(KB)])
    np.testing.assert_allclose(part, blocks, rtol=1e-5, atol=1e-5)


def test_label_features_match_oracle(lib):
    B, n = 2, 1500
    trips = [synth_mixture(60 + b, n, return_sources=True) for b in range(B)]
    specs = [[O.stft(w, 256, 64) for w in t] for t in trips]
    T, F = specs[0][0].shape
    ri = lambda i: np.ascontiguousarray(np.stack([np.stack([s[i].real, s[i].imag], -1) for s in specs]).astype(np.float32))
    mix, s1, s2 = ri(0), ri(1), ri(2)
    feat = np.stack([O.log_magnitude(s[0]) for s in specs])
    outs = {k: np.full((B, T, F) + ((2,) if k == "one_hot" else ()), np.nan, np.float32)
            for k in ("one_hot", "mag_mix", "mag_s1", "mag_s2", "cos_s1", "cos_s2")}
    umax = np.zeros(B, np.float32)
    lib.labels(P(mix), P(s1), P(s2), P(feat), B, T, F, 40.0, P(umax), *[P(outs[k]) for k in
               ("one_hot", "mag_mix", "mag_s1", "mag_s2", "cos_s1", "cos_s2")], None)
    for b in range(B):
        X, S1, S2 = specs[b]
        np.testing.assert_array_equal(outs["one_hot"][b], O.one_hot_labels(feat[b], np.abs(S1), np.abs(S2), 40.0))
        np.testing.assert_allclose(outs["mag_mix"][b], np.abs(X), rtol=1e-6)
        np.testing.assert_allclose(outs["cos_s1"][b], O.cos_difference(X, S1), atol=2e-6)
        np.testing.assert_allclose(outs["cos_s2"][b], O.cos_difference(X, S2), atol=2e-6)
    assert 0.05 < outs["one_hot"].sum(-1).mean() < 0.98       # some bins active, some silent


def _two_cluster_embeddings(rng, B, T, F, D, noise=0.15):
    cents = rng.standard_normal((B, 2, D))
    cents /= np.linalg.norm(cents, axis=-1, keepdims=True)
    lab = rng.integers(0, 2, (B, T, F))
    e = np.take_along_axis(cents[:, None, None], lab[..., None, None], axis=3)[..., 0, :]
    e = e + noise * rng.standard_normal(e.shape)
    e /= np.linalg.norm(e, axis=-1, keepdims=True)
    feat = rng.uniform(-3.0, 1.0, (B, T, F)).astype(np.float32)
    return e.astype(np.float32), feat, lab


@pytest.mark.parametrize("D", [20, 12])
def test_dc_cluster_masks(lib, D):
    """Launch-per-iteration form of onssen_dc_cluster_f32 (no inter-workgroup waits) against the planted clusters."""
    rng = np.random.default_rng(11)
    B, T, F = 2, 20, 33
    emb, feat, lab = _two_cluster_embeddings(rng, B, T, F, D)
    nb = lib.dll.onssen_dc_cluster_workspace_bytes(B, T, F, D)
    ws = aligned_f32(nb // 4 + 4)
    masks = np.full((B, T, F, 2), np.nan, np.float32)
    lib.dc_cluster(P(emb), P(feat), B, T, F, D, 40.0, 10, P(masks), P(ws), nb, None, flags=_abi.DC_CLUSTER_LAUNCH_PER_ITERATION)
    for b in range(B):
        act = O.dc_active_bins(feat[b])
        assert np.all(masks[b][~act] == 0) and np.all(masks[b][act].sum(-1) == 1)
        agree = (masks[b][act][:, 0] == lab[b][act]).mean()
        assert max(agree, 1 - agree) > 0.995        # cluster numbering is arbitrary


def _shm(shape, fill=0.0, dtype=np.float32):
    """'Device' buffer in MAP_SHARED memory, 256-byte aligned: visible to forked workgroup processes."""
    import mmap
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    a = np.frombuffer(mmap.mmap(-1, max(n, 4096)), dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    a[...] = fill
    return a


# (more than 8 utterances per launch -- the second block of the workgroup -> utterance map -- run on the GPU at B = 32: forking
#  9 x 8 workgroups of 512 threads takes the emulation a minute)
@pytest.mark.parametrize("B,T,F,D,scramble", [(2, 20, 33, 20, "0"), (2, 9, 40, 12, "0"), (1, 20, 33, 20, "1")])
def test_dc_cluster_persistent_lloyd(lib, monkeypatch, B, T, F, D, scramble):
    """Default form of onssen_dc_cluster_f32: count + order-preserving compaction of the active bins, then ALL Lloyd
    iterations in one persistent launch whose 8 workgroups per utterance meet at a counter (forked workgroups over shared
    memory here).  Against the planted clusters, against the launch-per-iteration form, and the compacted rows themselves."""
    monkeypatch.setenv("ONSSEN_EMU_FORK", "1")
    monkeypatch.setenv("ONSSEN_EMU_SCRAMBLE_XCC", scramble)     # "1": the parts report different XCDs -> placement-independent accesses
    lib.dll.onssen_xcd_spin_limit(40000000)   # emulated workgroups are OS processes: be patient
    rng = np.random.default_rng(21)
    emb0, feat0, lab = _two_cluster_embeddings(rng, B, T, F, D)
    emb, feat = _shm(emb0.shape), _shm(feat0.shape)
    emb[...] = emb0; feat[...] = feat0
    nb = lib.dll.onssen_dc_cluster_workspace_bytes(B, T, F, D)
    ws = _shm((nb // 4 + 64,))
    masks = _shm((B, T, F, 2), fill=np.nan)
    iters = 8
    lib.dc_cluster(P(emb), P(feat), B, T, F, D, 40.0, iters, P(masks), P(ws), nb, None)
    so = lib.dll.onssen_dc_cluster_status_offset(B, D)
    assert ws.view(np.uint32)[so // 4] == 0
    # the compacted copy: active rows of every utterance in bin order
    off = (so + 256 + 255) // 256 * 256 // 4
    comp = ws[off:off + B * T * F * D].reshape(B, T * F, D)
    iw = ws.view(np.int32)[(so // 4) - B * 72:so // 4].reshape(B, 72)
    assert all(bool(iw[b, 66] & 0x20000) == (scramble == "1") and (iw[b, 66] & 0xffff) >= 1 for b in range(B))
    for b in range(B):
        act = O.dc_active_bins(feat0[b]).reshape(-1)
        assert iw[b, 64] == act.sum()
        np.testing.assert_array_equal(comp[b, :act.sum()], emb0[b].reshape(-1, D)[act])
    monkeypatch.setenv("ONSSEN_EMU_FORK", "0")
    ws2 = aligned_f32(nb // 4 + 4)
    ref = np.full((B, T, F, 2), np.nan, np.float32)
    lib.dc_cluster(P(emb0), P(feat0), B, T, F, D, 40.0, iters, P(ref), P(ws2), nb, None, flags=_abi.DC_CLUSTER_LAUNCH_PER_ITERATION)
    for b in range(B):
        act = O.dc_active_bins(feat0[b])
        assert np.all(masks[b][~act] == 0) and np.all(masks[b][act].sum(-1) == 1)
        agree = (masks[b][act][:, 0] == lab[b][act]).mean()
        assert max(agree, 1 - agree) > 0.995
        same = (masks[b] == ref[b]).all(-1).mean()
        assert same > 0.995                      # same initialisation, same fixed point (summation orders differ)


def _poison_handoff(ws, B, T, Hp, NP, L):
    """NaNs with the exchange's tag bits set in the h hand-off area of a BLSTM workspace (header | G | ybuf | c | HERE:
    onssen_hip.hip, blstm_ws_layout): the persistent kernels clear their slots -- K padding included -- themselves, no host
    memset stands behind them."""
    a256 = lambda n: (n + 255) // 256 * 256
    off = _abi.BLSTM_WS_HEADER + a256(T * B * 2 * NP * 4) + (a256(T * B * 2 * Hp * 4) if L > 1 else 0) + a256(2 * B * Hp * 4)
    n = a256(2 * 2 * ((B + 3) // 4) * ((Hp + 31) // 32) * 2048)
    ws.view(np.uint32)[off // 4:(off + n) // 4] = 0x7fc07fc0


# fuse 2: FUSE_IN0 + FUSE_TAIL, F = 33; bf16 1: ONSSEN_BLSTM_BF16 (opt-in plain bf16 products)
@pytest.mark.parametrize("H,ug,B,T,scramble,fuse,bf16", _half_grid(
    [(8, 4, 3, 4), (24, 8, 17, 3), (32, 4, 2, 6)],
    [("0", 0, 0), ("1", 0, 0), ("0", 1, 0), ("0", 2, 0), ("0", 0, 1), ("0", 1, 1)]))
def test_blstm_xcd_local_persistent(lib, monkeypatch, H, ug, B, T, scramble, fuse, bf16):
    """ONSSEN_BLSTM_XCD: one persistent launch per layer, h exchanged inside the launch.  The mock runtime runs
    every workgroup concurrently (forked) over shared memory; scramble=1 makes the members of a group report
    different XCC ids, which must select the placement-independent protocol (status word 281)."""
    monkeypatch.setenv("ONSSEN_EMU_FORK", "1")
    monkeypatch.setenv("ONSSEN_EMU_SCRAMBLE_XCC", scramble)
    lib.dll.onssen_xcd_spin_limit(40000000)   # emulated workgroups are OS processes: be patient
    F, L = (33 if fuse == 2 else 9), 2
    sd = make_state_dict("chimera", F, H, L, 4, 2, seed=H + ug, gain=2.0)
    rng = np.random.default_rng(3)
    x = _shm((B, T, F)); x[...] = rand(rng, B, T, F)
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
        pl = _shm((2 * NP, (K + 31) // 32, 2, 32), dtype=np.uint16)      # x3 image of the [2*NP][K] projection matrix
        lib.x3_image(P(a), Kp, 0, 1, 2 * NP, K, P(pl), None)
        if fuse and l == 0:   # ONSSEN_BLSTM_FUSE_IN0: B-fragment image of the two W_ih, x W_ih^T inside the recurrence launch
            kc = (K + 31) // 32
            pl = _shm((2, (Hp // ug) * kc * (ug // 4) * 1024), dtype=np.uint16)
            for d, sfx in enumerate(("", "_reverse")):
                wsrc = _shm(sd[f"rnn.weight_ih_l0{sfx}"].shape); wsrc[...] = sd[f"rnn.weight_ih_l0{sfx}"]
                lib.lstm_pack_wih_bf16x3(P(wsrc), K, H, ug, P(pl[d]), None)
            if fuse == 2:         # the bias, then the last column of the packed W_ih
                ct = _shm((4, NP)); ct[:2] = c; ct[2:] = a[:, :, K - 1]
                c = ct
        wih3.append(pl), whh3.append(b3), bias.append(c)
    ws = _shm((lib.blstm_workspace_bytes(B, T, F, H, L, ug) // 4 + 64,))
    _poison_handoff(ws, B, T, Hp, NP, L)
    y = _shm((T, B, 2, Hp), fill=np.nan)
    lib.blstm_forward(P(x), T * F, F, B, T, F, H, L, ug, [P(a) for a in wih3], [P(a) for a in whh3],
                      [P(a) for a in bias], P(y), P(ws), ws.nbytes,
                      _abi.BLSTM_BF16X3 | _abi.BLSTM_XCD | (_abi.BLSTM_FUSE_IN0 if fuse else 0) |
                      (_abi.BLSTM_FUSE_TAIL if fuse == 2 else 0) | (_abi.BLSTM_BF16 if bf16 else 0), None)
    status = ws.view(np.uint32)
    assert status[280] == 0, f"launch aborted (code {status[280]})"
    assert status[281] == (1 if scramble == "1" else 0)
    ref = O.blstm_stack(np.array(x), sd, "rnn.", L)
    got = np.concatenate([y[:, :, 0, :H], y[:, :, 1, :H]], -1).transpose(1, 0, 2)
    if bf16:
        # the mode's own arithmetic restated: operands of both products rounded to bf16, everything else fp32.  A value
        # that sits on a rounding boundary may fall the other way (one bf16 ulp of one h): 3e-3 abs; and the mode
        # must really be bf16-grade, i.e. measurably away from the fp32 reference yet inside 5e-2 of it
        ref16 = O.blstm_stack(np.array(x), sd, "rnn.", L, rnd=O.bf16_round)
        assert np.abs(got - ref16).max() < 3e-3 and np.abs(got - ref16).mean() < 1e-4
        assert 1e-5 < np.abs(got - ref).max() < 5e-2
        return
    assert np.abs(got - ref).max() < 2e-5
    assert np.all(np.array(y)[:, :, :, H:] == 0)


def _x3_decode(img, rows, K):
    """fp32 values of an x3 image [rows][ceil(K/32)][2][32] (bf16 hi | lo halves): hi + lo."""
    v = img.reshape(rows, -1, 2, 32).astype(np.uint32) << 16
    f = v.view(np.float32)
    return (f[:, :, 0, :] + f[:, :, 1, :]).reshape(rows, -1)[:, :K]


@pytest.mark.parametrize("H,ug,B,T,scramble", [(8, 4, 3, 4, "0"), (24, 8, 17, 3, "0"), (16, 4, 9, 5, "1")])
def test_blstm_pipe2_two_layers_pipelined_over_calls(lib, monkeypatch, H, ug, B, T, scramble):
    """onssen_blstm_pipe2_forward_f32 (round 6): call n runs layer 1 of batch n-1 beside layer 0 of batch n in ONE persistent
    launch (each on its own groups); after call n the workspace holds the x3 image of the stack's output for batch n-1.
    Three different batches + one draining call against the oracle's two-layer stack; the image bits equal those of the
    sequential persistent form's y (same arithmetic per group: 8-row stacked up to B = 16, 16-row unstacked above)."""
    monkeypatch.setenv("ONSSEN_EMU_FORK", "1")
    monkeypatch.setenv("ONSSEN_EMU_SCRAMBLE_XCC", scramble)
    lib.dll.onssen_xcd_spin_limit(40000000)
    F, L = 9, 2
    sd = make_state_dict("chimera", F, H, L, 4, 2, seed=H + ug, gain=2.0)
    rng = np.random.default_rng(5)
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
    nb = lib.blstm_pipe2_workspace_bytes(B, T, F, H, ug)
    assert nb > 0 and lib.blstm_pipe2_workspace_bytes(33, T, F, H, ug) == 0      # B <= 32
    ws = _shm((nb // 4 + 64,))
    off, KB = lib.blstm_pipe2_y_image(B, T, F, H, ug)
    flags = _abi.BLSTM_BF16X3 | _abi.BLSTM_XCD
    xs = [rand(rng, B, T, F) for _ in range(3)]
    x = _shm((B, T, F))
    got = []
    for n in range(4):
        x[...] = xs[min(n, 2)]
        lib.blstm_pipe2_forward(P(x), T * F, F, B, T, F, H, ug, [P(a) for a in wih3], [P(a) for a in whh3], [P(a) for a in bias],
                                P(ws), ws.nbytes, flags, None)
        status = ws.view(np.uint32)
        assert status[280] == 0, f"launch aborted (code {status[280]})"
        assert status[281] == (1 if scramble == "1" else 0)
        img = ws.view(np.uint16)[off // 2:off // 2 + T * B * KB * 64]
        got.append(_x3_decode(np.array(img), T * B, 2 * Hp).reshape(T, B, 2, Hp))
    assert np.all(np.isfinite(got[0]))                # call 0: the answer to an all-zero layer-1 projection
    for n in range(3):
        ref = O.blstm_stack(xs[n], sd, "rnn.", L)
        y = got[n + 1]
        out = np.concatenate([y[:, :, 0, :H], y[:, :, 1, :H]], -1).transpose(1, 0, 2)
        assert np.abs(out - ref).max() < 2e-5, n
        assert np.all(y[:, :, :, H:] == 0)
    # the sequential persistent form on the same rows: identical bits.  Up to 16 rows both run stacked tiles (4- or 8-row groups: a
    # tile column never sees its neighbours); above, the pair runs 16-row groups, which the sequential form uses from 33 rows on
    B2 = B if B <= 16 else 34
    x2 = _shm((B2, T, F)); x2[...] = rand(rng, B2, T, F); x2[:B] = xs[1]
    ws2 = _shm((lib.blstm_workspace_bytes(B2, T, F, H, L, ug) // 4 + 64,))
    y2 = _shm((T, B2, 2, Hp), fill=np.nan)
    lib.blstm_forward(P(x2), T * F, F, B2, T, F, H, L, ug, [P(a) for a in wih3], [P(a) for a in whh3], [P(a) for a in bias], P(y2), P(ws2),
                      ws2.nbytes, flags, None)
    off2, _ = lib.blstm_y_image(B2, T, F, H, L, ug)
    img2 = _x3_decode(np.array(ws2.view(np.uint16)[off2 // 2:off2 // 2 + T * B2 * KB * 64]), T * B2, 2 * Hp).reshape(T, B2, 2, Hp)
    np.testing.assert_array_equal(got[2], img2[:, :B])
    # wrong flags are refused (the fused first layer, plain bf16 products, the launch-per-step form)
    for bad in (flags | _abi.BLSTM_FUSE_IN0, flags | _abi.BLSTM_BF16, _abi.BLSTM_BF16X3):
        with pytest.raises(_abi.OnssenError):
            lib.blstm_pipe2_forward(P(x), T * F, F, B, T, F, H, ug, [P(a) for a in wih3], [P(a) for a in whh3], [P(a) for a in bias],
                                    P(ws), ws.nbytes, bad, None)


@pytest.mark.parametrize("H,ug,B,scramble", [(8, 4, 3, "0"), (24, 8, 9, "0"), (16, 4, 5, "1")])
def test_blstm_pipe2_ragged_batches_pipelined_over_calls(lib, monkeypatch, H, ug, B, scramble):
    """onssen_blstm_pipe2_forward_ragged_f32 (round 6c): the pair launch over a stream of RAGGED batches -- each call brings a batch
    padded to its OWN longest row (T, frames), the launch's other half still runs the batch before (T_prev, frames_prev); one workspace
    laid out for T_cap.  Every row at its own frames against the oracle's batch-1 run of that row, zeros behind them; the image bits
    equal those of onssen_blstm_forward_ragged_f32 on the same batch (stacked tiles in both)."""
    monkeypatch.setenv("ONSSEN_EMU_FORK", "1")
    monkeypatch.setenv("ONSSEN_EMU_SCRAMBLE_XCC", scramble)
    lib.dll.onssen_xcd_spin_limit(40000000)
    F, L, T_cap = 9, 2, 6
    sd = make_state_dict("chimera", F, H, L, 4, 2, seed=H + ug, gain=2.0)
    rng = np.random.default_rng(7)
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
    nb = lib.blstm_pipe2_workspace_bytes(B, T_cap, F, H, ug)
    ws = _shm((nb // 4 + 64,))
    off, KB = lib.blstm_pipe2_y_image(B, T_cap, F, H, ug)
    flags = _abi.BLSTM_BF16X3 | _abi.BLSTM_XCD
    # three batches of different padded lengths (the second is the longest, the third the shortest), every row with its own length
    Ts = [4, 6, 3]
    frs = [np.minimum(rng.integers(1, T + 1, B), T).astype(np.int32) for T in Ts]
    for f, T in zip(frs, Ts):
        f[rng.integers(0, B)] = T                                     # (a batch is padded to its longest row)
    xs = [rand(rng, B, T, F) for T in Ts]
    wih_p, whh_p, bias_p = [P(a) for a in wih3], [P(a) for a in whh3], [P(a) for a in bias]
    got = []
    prev = None
    for n in range(4):
        k = min(n, 2)
        T = Ts[k]
        x = _shm((B, T, F)); x[...] = xs[k]
        fr = _shm((B,), dtype=np.int32); fr[...] = frs[k]
        if prev is None:
            prev = (T, fr)
        lib.blstm_pipe2_forward_ragged(P(x), T * F, F, B, T_cap, T, P(fr), prev[0], P(prev[1]), F, H, ug, wih_p, whh_p, bias_p,
                                       P(ws), ws.nbytes, flags, None)
        status = ws.view(np.uint32)
        assert status[280] == 0, f"launch aborted (code {status[280]})"
        Tp = prev[0]
        img = np.array(ws.view(np.uint16)[off // 2:off // 2 + Tp * B * KB * 64])
        got.append(_x3_decode(img, Tp * B, 2 * Hp).reshape(Tp, B, 2, Hp))
        prev = (T, fr)
    for n in range(3):
        y, T = got[n + 1], Ts[n]
        out = np.concatenate([y[:, :, 0, :H], y[:, :, 1, :H]], -1).transpose(1, 0, 2)
        for b in range(B):
            m = int(frs[n][b])
            ref = O.blstm_stack(xs[n][b:b + 1, :m], sd, "rnn.", L)
            assert np.abs(out[b, :m] - ref[0]).max() < 2e-5, (n, b)
            assert np.all(out[b, m:] == 0), (n, b)
        assert np.all(y[:, :, :, H:] == 0)
    # the sequential ragged form on the same batch: identical bits
    n = 1
    T = Ts[n]
    x2 = _shm((B, T, F)); x2[...] = xs[n]
    fr2 = _shm((B,), dtype=np.int32); fr2[...] = frs[n]
    ws2 = _shm((lib.blstm_workspace_bytes(B, T, F, H, L, ug) // 4 + 64,))
    y2 = _shm((T, B, 2, Hp), fill=np.nan)
    lib.blstm_forward(P(x2), T * F, F, B, T, F, H, L, ug, wih_p, whh_p, bias_p, P(y2), P(ws2), ws2.nbytes, flags, None, frames=P(fr2))
    off2, _ = lib.blstm_y_image(B, T, F, H, L, ug)
    img2 = _x3_decode(np.array(ws2.view(np.uint16)[off2 // 2:off2 // 2 + T * B * KB * 64]), T * B, 2 * Hp).reshape(T, B, 2, Hp)
    np.testing.assert_array_equal(got[n + 1], img2)
    # refused: a longer batch than the workspace was laid out for, more than 16 ragged rows, one batch without its frames
    x = _shm((B, T_cap + 1, F))
    with pytest.raises(_abi.OnssenError):
        lib.blstm_pipe2_forward_ragged(P(x), (T_cap + 1) * F, F, B, T_cap, T_cap + 1, P(fr2), T, P(fr2), F, H, ug, wih_p, whh_p, bias_p,
                                       P(ws), ws.nbytes, flags, None)
    with pytest.raises(_abi.OnssenError):
        lib.blstm_pipe2_forward_ragged(P(x2), T * F, F, B, T_cap, T, None, T, P(fr2), F, H, ug, wih_p, whh_p, bias_p, P(ws), ws.nbytes,
                                       flags, None)


@pytest.mark.parametrize("H,B,T,ragged", [(48, 3, 4, False), (40, 17, 3, False), (48, 5, 4, True)])
def test_blstm_xcd_24_unit_groups(lib, monkeypatch, H, B, T, ragged):
    """Round 4: 24 hidden units per member (640 < H <= 768 on the device: 32 members = every CU of an XCD) -- split-bf16 only,
    neither the fused first projection nor the training state; the other precision modes and those flags are refused."""
    monkeypatch.setenv("ONSSEN_EMU_FORK", "1")
    monkeypatch.setenv("ONSSEN_EMU_SCRAMBLE_XCC", "0")
    lib.dll.onssen_xcd_spin_limit(40000000)
    F, L, ug = 9, 2, 24
    sd = make_state_dict("chimera", F, H, L, 4, 2, seed=H + ug, gain=2.0)
    rng = np.random.default_rng(5)
    x = _shm((B, T, F)); x[...] = rand(rng, B, T, F)
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
    ws = _shm((lib.blstm_workspace_bytes(B, T, F, H, L, ug) // 4 + 64,))
    _poison_handoff(ws, B, T, Hp, NP, L)
    y = _shm((T, B, 2, Hp), fill=np.nan)
    args = (P(x), T * F, F, B, T, F, H, L, ug, [P(a) for a in wih3], [P(a) for a in whh3], [P(a) for a in bias], P(y), P(ws),
            ws.nbytes)
    for bad in (_abi.BLSTM_XCD, _abi.BLSTM_BF16X3 | _abi.BLSTM_XCD | _abi.BLSTM_FUSE_IN0,
                _abi.BLSTM_BF16X3 | _abi.BLSTM_XCD | _abi.BLSTM_BF16):
        with pytest.raises(_abi.OnssenError):
            lib.blstm_forward(*args, bad, None)
    frames = None
    if ragged:
        frames = _shm((B,), dtype=np.int32); frames[...] = [4, 1, 3, 2, 4]
        lib.blstm_forward(*args, _abi.BLSTM_BF16X3 | _abi.BLSTM_XCD, None, frames=P(frames))
    else:
        lib.blstm_forward(*args, _abi.BLSTM_BF16X3 | _abi.BLSTM_XCD, None)
    status = ws.view(np.uint32)
    assert status[280] == 0, f"launch aborted (code {status[280]})"
    got = np.concatenate([y[:, :, 0, :H], y[:, :, 1, :H]], -1).transpose(1, 0, 2)
    if ragged:
        for b in range(B):
            n = int(frames[b])
            ref = O.blstm_stack(np.array(x[b:b + 1, :n]), sd, "rnn.", L)
            assert np.abs(got[b, :n] - ref[0]).max() < 2e-5
            assert np.all(got[b, n:] == 0)
        return
    ref = O.blstm_stack(np.array(x), sd, "rnn.", L)
    assert np.abs(got - ref).max() < 2e-5
    assert np.all(np.array(y)[:, :, :, H:] == 0)


@pytest.mark.parametrize("H,ug,B,T,scramble", [(8, 4, 3, 4, "0"), (24, 8, 17, 3, "0"), (40, 20, 5, 5, "0"), (24, 8, 6, 3, "1")])
def test_blstm_xcd_exact_fp32(lib, monkeypatch, H, ug, B, T, scramble):
    """ONSSEN_BLSTM_XCD WITHOUT ONSSEN_BLSTM_BF16X3 (round 3): the persistent recurrence in exact fp32 -- fp32 fragment
    images of onssen_lstm_pack_f32, v_mfma_f32_16x16x4_f32, h handed on as fp32 words tagged in bit 30, fp32 rows between
    the layers -- against the oracle at fp32 tolerance; scrambled XCC ids select the placement-independent accesses."""
    monkeypatch.setenv("ONSSEN_EMU_FORK", "1")
    monkeypatch.setenv("ONSSEN_EMU_SCRAMBLE_XCC", scramble)
    lib.dll.onssen_xcd_spin_limit(40000000)
    F, L = 9, 2
    sd = make_state_dict("chimera", F, H, L, 4, 2, seed=H + ug + 1, gain=2.0)
    rng = np.random.default_rng(4)
    x = _shm((B, T, F)); x[...] = rand(rng, B, T, F)
    Hp, NP, KQ, we = lib.lstm_geometry(H, ug)
    wih, whh, bias = [], [], []
    for l in range(L):
        Kp = (F + 3) // 4 * 4 if l == 0 else 2 * Hp
        a, b, c = _shm((2, NP, Kp)), _shm((2, we)), _shm((2, NP))
        for d, sfx in enumerate(("", "_reverse")):
            srcs = []
            for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                v = sd[f"rnn.{n}_l{l}{sfx}"]
                sv = _shm(v.shape); sv[...] = v
                srcs.append(sv)
            lib.lstm_pack(P(srcs[0]), P(srcs[1]), P(srcs[2]), P(srcs[3]), srcs[0].shape[1], 0 if l == 0 else 1, H, ug,
                          P(a[d]), P(b[d]), P(c[d]), None)
        wih.append(a), whh.append(b), bias.append(c)
    ws = _shm((lib.blstm_workspace_bytes(B, T, F, H, L, ug) // 4 + 64,))
    _poison_handoff(ws, B, T, Hp, NP, L)
    y = _shm((T, B, 2, Hp), fill=np.nan)
    lib.blstm_forward(P(x), T * F, F, B, T, F, H, L, ug, [P(a) for a in wih], [P(a) for a in whh], [P(a) for a in bias],
                      P(y), P(ws), ws.nbytes, _abi.BLSTM_XCD, None)
    status = ws.view(np.uint32)
    assert status[280] == 0, f"launch aborted (code {status[280]})"
    assert status[281] == (1 if scramble == "1" else 0) and status[282] == 0
    ref = O.blstm_stack(np.array(x), sd, "rnn.", L)
    got = np.concatenate([y[:, :, 0, :H], y[:, :, 1, :H]], -1).transpose(1, 0, 2)
    assert np.abs(got - ref).max() < 3e-6
    assert np.all(np.array(y)[:, :, :, H:] == 0)


# (1, 300, 30, 4): D + C = 34 > 32 -- the LDS form, two tasks per thread; the others: the MFMA form (4000 bins: several strips per wave)
@pytest.mark.parametrize("B,TF,D,C", [(2, 150, 20, 2), (1, 700, 6, 3), (1, 300, 30, 4), (1, 4000, 20, 2), (1, 515, 28, 4)])
def test_loss_dc_value(lib, B, TF, D, C):
    """onssen_loss_dc_f32 against the NumPy restatement of loss_dc (Frobenius norms of the weighted affinity blocks)."""
    rng = np.random.default_rng(5)
    emb = rand(rng, B, TF, D)
    emb /= np.linalg.norm(emb, axis=-1, keepdims=True)
    lab = rng.integers(0, C + 1, size=(B, TF))                      # C = silent bin
    one_hot = np.zeros((B, TF, C), np.float32)
    for c in range(C):
        one_hot[..., c] = lab == c
    mag = np.abs(rand(rng, B, TF)) + 0.01
    per_utt = np.full(B, np.nan, np.float32)
    total = np.full(B, np.nan, np.float32)
    ws = aligned_f32(lib.loss_dc_workspace_bytes(B) // 4 + 64)
    lib.loss_dc(P(emb), P(one_hot), P(mag), B, TF, D, C, P(per_utt), P(total), P(ws), ws.nbytes, None)
    ref = O.loss_dc_per_utt(emb, one_hot, mag)
    np.testing.assert_allclose(total, mag.sum(1), rtol=1e-5)
    np.testing.assert_allclose(per_utt, ref, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("B,TF,D,C", [(2, 300, 20, 2), (1, 700, 6, 3), (1, 200, 30, 4)])
def test_loss_dc_gradient(lib, B, TF, D, C):
    """onssen_loss_dc_grad_f32 against float64 autograd through the LITERAL form of onssen/loss/loss_dc.py:24-44 (three products,
    Frobenius norms, detached weights), for an arbitrary upstream gradient per utterance."""
    import torch
    rng = np.random.default_rng(6)
    emb = rand(rng, B, TF, D)
    emb /= np.linalg.norm(emb, axis=-1, keepdims=True)
    lab = rng.integers(0, C + 1, size=(B, TF))
    one_hot = np.zeros((B, TF, C), np.float32)
    for c in range(C):
        one_hot[..., c] = lab == c
    mag = np.abs(rand(rng, B, TF)) + 0.01
    g = rand(rng, B)
    per_utt, total = np.full(B, np.nan, np.float32), np.full(B, np.nan, np.float32)
    ws = aligned_f32(lib.loss_dc_workspace_bytes(B) // 4 + 64)
    lib.loss_dc(P(emb), P(one_hot), P(mag), B, TF, D, C, P(per_utt), P(total), P(ws), ws.nbytes, None)
    d_emb = np.full((B, TF, D), np.nan, np.float32)
    lib.loss_dc_grad(P(emb), P(one_hot), P(mag), B, TF, D, C, P(g), P(d_emb), P(ws), ws.nbytes, None)
    V = torch.from_numpy(emb).double().requires_grad_(True)
    Y, m = torch.from_numpy(one_hot).double(), torch.from_numpy(mag).double()
    w = torch.sqrt(m / m.sum(1, keepdim=True)).unsqueeze(-1)
    Vm, Ym = V * Y.sum(2, keepdim=True) * w, Y * w
    fro = lambda x: torch.sqrt((x * x).flatten(1).sum(1))
    val = fro(Vm.transpose(1, 2) @ Vm) - 2 * fro(Vm.transpose(1, 2) @ Ym) + fro(Ym.transpose(1, 2) @ Ym)
    (val * torch.from_numpy(g).double()).sum().backward()
    np.testing.assert_allclose(per_utt, val.detach().numpy(), rtol=1e-4)
    ref = V.grad.numpy()
    assert np.abs(d_emb - ref).max() <= 2e-5 * np.abs(ref).max()


@pytest.mark.parametrize("B,C,n,use_mask", [(2, 2, 1500, False), (1, 3, 900, True), (1, 4, 700, False)])
def test_batch_sdr(lib, B, C, n, use_mask):
    """onssen_batch_sdr_f32 against the NumPy restatement of batch_SDR_torch (values and permutation index)."""
    rng = np.random.default_rng(21)
    org = rand(rng, B, C, n)
    est = np.ascontiguousarray(org[:, ::-1] * 0.7 + 0.4 * rand(rng, B, C, n) + 0.1).astype(np.float32)
    mask = None
    if use_mask:
        mask = np.ones((B, n), np.float32); mask[:, n - 200:] = 0
    sdr = np.full(B, np.nan, np.float32); perm = np.full(B, -1, np.int32)
    ws = aligned_f32(lib.batch_sdr_workspace_bytes(B) // 4 + 64)
    lib.batch_sdr(P(est), P(org), P(mask), B, C, n, P(sdr), P(perm), P(ws), ws.nbytes, None)
    ref_sdr, ref_perm = O.batch_sdr(est, org, mask)
    np.testing.assert_allclose(sdr, ref_sdr, rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(perm, ref_perm)


@pytest.mark.parametrize("noise", [1e-2, 1e-4, 0.0])
def test_batch_sdr_near_perfect_estimates(lib, noise):
    """the residual power from a Gram matrix (ee - 2 s eo + s^2 oo) cancels; with fp32 sums the metric was off
    by 0.1 dB at 50 dB and NaN from ~70 dB (oracle masks, est == ref sanity checks).  The sums are fp64 now: 40 dB, 80 dB
    and the est == org ceiling (res_power = 1e-8, sdr.py:30) must all come out like the reference's direct sum."""
    rng = np.random.default_rng(33)
    B, C, n = 2, 2, 4000
    org = rand(rng, B, C, n)
    est = (org + noise * rand(rng, B, C, n)).astype(np.float32)
    sdr = np.full(B, np.nan, np.float32); perm = np.full(B, -1, np.int32)
    ws = aligned_f32(lib.batch_sdr_workspace_bytes(B) // 4 + 64)
    lib.batch_sdr(P(est), P(org), None, B, C, n, P(sdr), P(perm), P(ws), ws.nbytes, None)
    ref_sdr, ref_perm = O.batch_sdr(est, org, None)
    assert np.all(np.isfinite(sdr)) and ref_sdr.min() > 35.0
    # fp32 inputs: the centred signals differ from the fp64 restatement by ~1e-7 relative, i.e. a floor near -140 dB
    np.testing.assert_allclose(sdr, ref_sdr, atol=0.02 if noise else 0.5)
    np.testing.assert_array_equal(perm, ref_perm)


@pytest.mark.parametrize("psa", [False, True])
def test_loss_mask_term(lib, psa):
    """onssen_loss_mask_f32 against NumPy: both speaker assignments, MSA and PSA targets, strided mask views."""
    rng = np.random.default_rng(12)
    B, TF = 2, 700
    masks = rng.random((B, TF, 2)).astype(np.float32)            # interleaved like fc_mi's output
    mag = (np.abs(rand(rng, B, TF)) + 1e-3).astype(np.float32)
    s1, s2 = (mag * rng.random((B, TF))).astype(np.float32), (mag * rng.random((B, TF))).astype(np.float32)
    c1, c2 = rand(rng, B, TF).clip(-1, 1), rand(rng, B, TF).clip(-1, 1)
    out = np.full(B, np.nan, np.float32)
    ma, mb = masks[..., 0], masks[..., 1]
    ws = aligned_f32(lib.loss_mask_workspace_bytes(B) // 4 + 64)
    lib.loss_mask(P(masks), masks.ctypes.data + 4, 2 * TF, 2, P(mag), P(s1), P(s2), P(c1) if psa else None, P(c2) if psa else None,
                  B, TF, P(out), P(ws), ws.nbytes, None)
    t1, t2 = (np.minimum(mag, np.maximum(s1 * c1, 0)), np.minimum(mag, np.maximum(s2 * c2, 0))) if psa else (s1, s2)
    l1 = lambda a: np.abs(a.astype(np.float64)).sum(1)
    ref = np.minimum(l1(ma * mag - t1) + l1(mb * mag - t2), l1(mb * mag - t1) + l1(ma * mag - t2))
    np.testing.assert_allclose(out, ref, rtol=1e-5)


@pytest.mark.parametrize("psa", [False, True])
def test_loss_mask_gradient(lib, psa):
    """onssen_loss_mask_grad_f32 against torch autograd of the literal form (onssen/loss/loss_chimera.py:25-29 / :53-57):
    the permutation picked by the forward kernels, sign(0) = 0, the incoming gradient per utterance, interleaved output."""
    import torch
    rng = np.random.default_rng(13)
    B, TF = 3, 900
    masks = rng.random((B, TF, 2)).astype(np.float32)
    mag = (np.abs(rand(rng, B, TF)) + 1e-3).astype(np.float32)
    s1, s2 = (mag * rng.random((B, TF))).astype(np.float32), (mag * rng.random((B, TF))).astype(np.float32)
    c1, c2 = rand(rng, B, TF).clip(-1, 1), rand(rng, B, TF).clip(-1, 1)
    tg1, tg2 = (np.minimum(mag, np.maximum(s1 * c1, 0)), np.minimum(mag, np.maximum(s2 * c2, 0))) if psa else (s1, s2)
    # utterance 0 estimates (speaker 1, speaker 2), utterance 1 the swapped order: both assignments occur
    masks[0, :, 0], masks[0, :, 1] = 0.8 * tg1[0] / mag[0] + 0.05, 0.8 * tg2[0] / mag[0] + 0.05
    masks[1, :, 0], masks[1, :, 1] = 0.8 * tg2[1] / mag[1] + 0.05, 0.8 * tg1[1] / mag[1] + 0.05
    masks[2, 5, 0] = tg1[2, 5] / mag[2, 5]                                       # a residual that may be exactly 0
    out, perm = np.full(B, np.nan, np.float32), np.full(B, -1, np.int32)
    ws = aligned_f32(lib.loss_mask_workspace_bytes(B) // 4 + 64)
    cc = (P(c1), P(c2)) if psa else (None, None)
    lib.loss_mask(P(masks), masks.ctypes.data + 4, 2 * TF, 2, P(mag), P(s1), P(s2), *cc, B, TF, P(out), P(ws), ws.nbytes, None,
                  perm=P(perm))
    g = rng.standard_normal(B).astype(np.float32)
    d = np.full((B, TF, 2), np.nan, np.float32)
    lib.loss_mask_grad(P(masks), masks.ctypes.data + 4, 2 * TF, 2, P(mag), P(s1), P(s2), *cc, B, TF, P(g), P(perm), P(d),
                       d.ctypes.data + 4, 2 * TF, 2, None)
    tt = lambda a: torch.from_numpy(a.astype(np.float64))
    m = tt(masks).requires_grad_(True)
    x, t1, t2 = tt(mag), tt(s1), tt(s2)
    if psa:
        t1, t2 = torch.minimum(x, torch.relu(t1 * tt(c1))), torch.minimum(x, torch.relu(t2 * tt(c2)))
    ma, mb = m[..., 0], m[..., 1]
    l1 = lambda a: a.abs().sum(1)
    l_ab, l_ba = l1(ma * x - t1) + l1(mb * x - t2), l1(mb * x - t1) + l1(ma * x - t2)
    ref = torch.min(l_ab, l_ba)
    np.testing.assert_array_equal(perm, (l_ba < l_ab).numpy().astype(np.int32))
    assert perm[0] == 0 and perm[1] == 1
    (ref * tt(g)).sum().backward()
    np.testing.assert_allclose(out, ref.detach().numpy(), rtol=1e-5)
    ref_d = m.grad.numpy()
    # fp32 residuals within rounding of 0 may take either sign: compare where the fp64 residual is clearly non-zero
    ra = (ma * x - torch.where(torch.from_numpy(perm)[:, None] == 0, t1, t2)).detach().abs().numpy()
    rb = (mb * x - torch.where(torch.from_numpy(perm)[:, None] == 0, t2, t1)).detach().abs().numpy()
    clear = np.stack([ra, rb], -1) > 1e-6
    np.testing.assert_allclose(d[clear], ref_d[clear], rtol=1e-6, atol=1e-12)
    assert clear.mean() > 0.99 and np.isfinite(d).all()


_TRAIN_CASES = [(H, ug, B, T, _abi.LSTM_BWD_XCD, "0") for H, ug, B, T in
                [(8, 4, 3, 5), (24, 8, 18, 3), (40, 20, 5, 4), (16, 8, 35, 2), (30, 4, 3, 17), (8, 4, 1, 1), (8, 4, 70, 2)]] + \
               [(24, 8, 18, 3, _abi.LSTM_BWD_XCD, "1"), (8, 4, 3, 5, _abi.LSTM_BWD_STEPS, "0"), (24, 8, 18, 3, _abi.LSTM_BWD_STEPS, "0")]
_TRAIN_CASES = [c + ("xcd",) for c in _TRAIN_CASES] + \
               [(8, 4, 3, 5, _abi.LSTM_BWD_STEPS, "0", "steps_x3"), (24, 8, 18, 3, _abi.LSTM_BWD_STEPS, "0", "steps_f32"),
                (12, 8, 20, 4, _abi.LSTM_BWD_STEPS, "0", "steps_x3"),
                # 24-unit members (640 < H <= 768 on the device): persistent forward with saved state, launch-per-step backward
                (48, 24, 5, 4, _abi.LSTM_BWD_STEPS, "0", "xcd"), (40, 24, 18, 3, _abi.LSTM_BWD_STEPS, "0", "xcd")]


@pytest.mark.parametrize("H,ug,B,T,form,scramble,fwd", _TRAIN_CASES)
def test_lstm_train_forward_and_backward_recurrence(lib, monkeypatch, H, ug, B, T, form, scramble, fwd):
    """Row N1: training forward (saved gates / cell states) + backward recurrence of one bidirectional layer against
    nn.LSTM autograd on the CPU (the reference's `loss.backward()`, onssen/utils/train.py:80-84).  Split-bf16 products:
    gradients within 2e-4 of their largest entry.  ``fwd``: the persistent forward, or (round 4: what an aborted training step
    is re-run on, and H > 640) the launch-per-step forward with saved state in split-bf16 / exact fp32."""
    import torch
    from onssen_amd.nn._train import layer_gradients
    monkeypatch.setenv("ONSSEN_EMU_FORK", "1")
    monkeypatch.setenv("ONSSEN_EMU_SCRAMBLE_XCC", "0")
    lib.dll.onssen_xcd_spin_limit(40000000)
    F = 9
    xcd = form == _abi.LSTM_BWD_XCD
    sd = make_state_dict("chimera", F, H, 1, 4, 2, seed=H + ug, gain=2.0)
    rng = np.random.default_rng(5)
    x = _shm((B, T, F)); x[...] = rand(rng, B, T, F)
    Hp, NP, KQ, we = lib.lstm_geometry(H, ug)
    _, _, we3 = lib.lstm_geometry_x3(H, ug)
    Kp = (F + 3) // 4 * 4
    a, c, b3 = _shm((2, NP, Kp)), _shm((2, NP)), _shm((2, we3), dtype=np.uint16)
    nT = (lib.lstm_whhR_elems if xcd else lib.lstm_whhT_elems)(H, ug)
    wT = _shm((2, nT), dtype=np.uint16)
    bw = _shm((2, we))
    for d, sfx in enumerate(("", "_reverse")):
        srcs = []
        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            v = sd[f"rnn.{n}_l0{sfx}"]
            sv = _shm(v.shape); sv[...] = v
            srcs.append(sv)
        lib.lstm_pack(P(srcs[0]), P(srcs[1]), P(srcs[2]), P(srcs[3]), F, 0, H, ug, P(a[d]), P(bw[d]), P(c[d]), None)
        lib.lstm_pack_whh_bf16x3(P(srcs[1]), H, ug, P(b3[d]), None)
        (lib.lstm_pack_whhR_bf16x3 if xcd else lib.lstm_pack_whhT_bf16x3)(P(srcs[1]), H, ug, P(wT[d]), None)
    pl = _shm((2 * NP, (F + 31) // 32, 2, 32), dtype=np.uint16)
    lib.x3_image(P(a), Kp, 0, 1, 2 * NP, F, P(pl), None)
    ws = _shm((lib.blstm_workspace_bytes(B, T, F, H, 1, ug) // 4 + 64,))
    y, gates, cs = _shm((T, B, 2, Hp), fill=np.nan), _shm((T, B, 2, NP), fill=np.nan), _shm((T, B, 2, Hp), fill=np.nan)
    if fwd == "xcd":
        lib.lstm_train_forward(P(x), T * F, F, B, T, F, H, ug, P(pl), P(b3), P(c), P(y), P(gates), P(cs), P(ws), ws.nbytes, None)
        assert ws.view(np.uint32)[280] == 0
    elif fwd == "steps_x3":
        ld = (F + 31) // 32 * 32
        planes = _shm((2, 2 * NP, ld), dtype=np.uint16)
        lib.linear_pack_bf16x3(P(a), 2 * NP, F, Kp, ld, P(planes), None)
        lib.lstm_train_forward_form(P(x), T * F, F, B, T, F, H, ug, P(planes), P(b3), P(c), P(y), P(gates), P(cs), P(ws), ws.nbytes,
                                    _abi.BLSTM_BF16X3, None)
    else:
        lib.lstm_train_forward_form(P(x), T * F, F, B, T, F, H, ug, P(a), P(bw), P(c), P(y), P(gates), P(cs), P(ws), ws.nbytes, 0, None)

    # the reference: nn.LSTM on the CPU, a random linear functional of its output as the loss
    lstm = torch.nn.LSTM(F, H, 1, batch_first=True, bidirectional=True)
    lstm.load_state_dict({k[4:]: torch.from_numpy(np.asarray(v)) for k, v in sd.items() if k.startswith("rnn.")})
    xt = torch.from_numpy(np.array(x)).requires_grad_(True)
    yr, _ = lstm(xt)
    R = torch.from_numpy(rand(rng, B, T, 2 * H))
    (yr * R).sum().backward()
    got = np.concatenate([y[:, :, 0, :H], y[:, :, 1, :H]], -1).transpose(1, 0, 2)
    assert np.abs(got - yr.detach().numpy()).max() < 2e-5

    dy = _shm((T, B, 2, Hp))
    dy[:, :, :, :H] = R.numpy().transpose(1, 0, 2).reshape(T, B, 2, H)
    wsb = _shm((lib.lstm_train_backward_workspace_bytes(B, H, ug, form) // 4 + 64,))
    monkeypatch.setenv("ONSSEN_EMU_SCRAMBLE_XCC", scramble)     # scramble=1: the placement-independent protocol
    db_rows = _shm((B, 2, NP), fill=np.nan) if xcd else None
    gates_saved = _shm(gates.shape); gates_saved[...] = gates
    gates_before = np.array(gates)             # a private copy: the _img launch below may only READ the saved gates
    lib.lstm_train_backward(B, T, H, ug, P(wT), P(dy), P(gates), P(cs), P(wsb), wsb.nbytes, form, None, P(db_rows) if xcd else None)
    if xcd and (2 * NP) % 32 == 0:
        # round 5: the same launch leaving dP as the gradient GEMMs' x3 image instead of fp32 -- bit for bit onssen_x3_image_f32 of
        # the fp32 dP wherever a unit owns the column (holes of padded units keep what the buffer held)
        img = _shm((T * B, 2 * NP // 32, 2, 32), dtype=np.uint16); img[...] = 0
        db2 = _shm((B, 2, NP), fill=np.nan)
        wsb2 = _shm((lib.lstm_train_backward_workspace_bytes(B, H, ug, form) // 4 + 64,))
        lib.lstm_train_backward_img(B, T, H, ug, P(wT), P(dy), P(gates_saved), P(cs), P(wsb2), wsb2.nbytes, None, P(db2), P(img))
        ref = _shm((T * B, 2 * NP // 32, 2, 32), dtype=np.uint16)          # (forked workgroups: shared memory)
        dp_fp32 = _shm((T * B, 2 * NP)); dp_fp32[...] = np.array(gates).reshape(T * B, 2 * NP)
        lib.x3_image(P(dp_fp32), 2 * NP, 0, 1, T * B, 2 * NP, P(ref), None)
        owned = np.zeros(NP, bool)
        for ugi in range(Hp // ug):
            for ju in range(ug):
                if ugi * ug + ju < H:
                    owned[ugi * 4 * ug + ju * 4: ugi * 4 * ug + ju * 4 + 4] = True
        own2 = np.concatenate([owned, owned]).reshape(2 * NP // 32, 32)
        got, want = np.array(img), np.array(ref)
        assert np.array_equal(got[:, :, 0][:, own2], want[:, :, 0][:, own2]) and np.array_equal(got[:, :, 1][:, own2], want[:, :, 1][:, own2])
        assert np.array_equal(np.array(gates_saved), gates_before) and wsb2.view(np.uint32)[280] == 0
        np.testing.assert_allclose(np.array(db2), np.array(db_rows), rtol=1e-6, atol=1e-7)
    if xcd:      # the kernel's per-row sums of dP over time (the bias gradient's operand) against the dP it wrote
        np.testing.assert_allclose(np.array(db_rows), np.array(gates).sum(0), rtol=1e-5, atol=1e-6)
    if xcd:
        assert wsb.view(np.uint32)[280] == 0 and wsb.view(np.uint32)[281] == (1 if scramble == "1" else 0)
    w_ih = (lstm.weight_ih_l0.detach(), lstm.weight_ih_l0_reverse.detach())
    x_rows = torch.from_numpy(np.array(x)).transpose(0, 1).reshape(T * B, F)
    dx_rows, g = layer_gradients(torch.from_numpy(np.array(gates)), x_rows, torch.from_numpy(np.array(y)), w_ih, H, ug)

    def close(a_, b_, what):
        a_, b_ = a_.numpy(), b_.numpy()
        assert np.abs(a_ - b_).max() <= 2e-4 * max(np.abs(b_).max(), 1e-3), (what, np.abs(a_ - b_).max(), np.abs(b_).max())
    close(dx_rows.view(T, B, F).transpose(0, 1), xt.grad, "dx")
    for d, sfx in enumerate(("", "_reverse")):
        close(g[d][0], getattr(lstm, f"weight_ih_l0{sfx}").grad, f"dW_ih{sfx}")
        close(g[d][1], getattr(lstm, f"weight_hh_l0{sfx}").grad, f"dW_hh{sfx}")
        close(g[d][2], getattr(lstm, f"bias_ih_l0{sfx}").grad, f"db{sfx}")
    assert np.all(np.array(gates).reshape(T, B, 2, NP // 4, 4)[:, :, :, H:, :] == 0) if ug * (Hp // ug) == Hp and H % ug == 0 else True


@pytest.mark.parametrize("H,ug,Kx,first", [(10, 4, 9, True), (10, 4, 24, False), (8, 8, 16, False), (8, 4, 9, True)])   # the last two: no padded units -> two dense outputs in nn.LSTM row order straight from the GEMM
def test_layer_gradient_gemms_on_the_split_bf16_kernel(lib, H, ug, Kx, first):
    """nn/_train.py: the weight / input gradient contractions in the packed layouts on onssen_linear_x3p agree with
    the plain torch contractions in the reference's layouts (fp32), within the split-bf16 product error."""
    import torch
    from onssen_amd.nn._train import layer_gradients, layer_gradients_x3, packed_columns
    torch.manual_seed(H + Kx)
    T, B = 5, 3
    Hp = -(-H // ug) * ug
    NP = 4 * Hp
    in_features = Kx if first else 2 * H
    assert first or Kx == 2 * Hp
    w_ih = [torch.randn(4 * H, in_features) for _ in range(2)]
    cols = packed_columns(H, Hp, ug)
    dP = torch.zeros(T, B, 2, NP)
    dP[..., cols] = torch.randn(T, B, 2, 4 * H)                      # padded units carry zero gradient
    y = torch.zeros(T, B, 2, Hp); y[..., :H] = torch.randn(T, B, 2, H)
    Kp = (Kx + 3) // 4 * 4 if first else 2 * Hp
    wih_p = torch.zeros(2, NP, Kp)
    if first:
        xp = torch.randn(T * B, Kx)
        x_rows = xp
        for d in range(2):
            wih_p[d, cols, :Kx] = w_ih[d]
    else:
        xin = torch.zeros(T, B, 2, Hp); xin[..., :H] = torch.randn(T, B, 2, H)
        xp = xin.view(T * B, 2 * Hp)
        x_rows = xin[..., :H].reshape(T * B, 2 * H)
        for d in range(2):
            for dd in range(2):
                wih_p[d, cols, dd * Hp: dd * Hp + H] = w_ih[d][:, dd * H:(dd + 1) * H]
    dx_ref, g_ref = layer_gradients(dP, x_rows, y, w_ih, H, ug)
    dx, g = layer_gradients_x3(lib, None, dP, xp, y, wih_p, H, ug, in_features, True)

    def close(a_, b_, what):
        assert (a_ - b_).abs().max() <= 1e-4 * max(b_.abs().max().item(), 1e-3), what
    if first:
        close(dx[:, :Kx], dx_ref, "dx")
    else:
        close(dx.view(T, B, 2, Hp)[..., :H].reshape(T * B, 2 * H), dx_ref, "dx")
        assert Hp == H or dx.view(T, B, 2, Hp)[..., H:].abs().max() == 0
    for d in range(2):
        for k, nm in enumerate(("dW_ih", "dW_hh", "db")):
            close(g[d][k], g_ref[d][k], f"{nm}[{d}]")


@pytest.mark.parametrize("M,K,shift", [(70, 45, 0), (33, 64, -3), (64, 100, 7)])
def test_x3_image_of_the_transpose(lib, M, K, shift):
    """onssen_x3_image_t_f32 == onssen_x3_image_f32 of the explicitly transposed (and k-shifted) matrix, bit for bit."""
    rng = np.random.default_rng(M + K)
    src = rand(rng, K, M + 5)                                  # [K][ld], ld > M
    explicit = np.zeros((M, K), dtype=np.float32)
    for k in range(K):
        if 0 <= k + shift < K:
            explicit[:, k] = src[k + shift, :M]
    KB = (K + 31) // 32
    want = np.zeros((M, KB, 2, 32), dtype=np.uint16)
    got = np.full((M, KB, 2, 32), 0xFFFF, dtype=np.uint16)
    lib.x3_image(P(explicit), K, 0, 1, M, K, P(want), None)
    lib.x3_image_t(P(src), M + 5, M, K, shift, P(got), None)
    assert np.array_equal(got, want)


def test_head_linear_on_the_split_bf16_kernel(lib):
    """nn/_train.py: nn.Linear forward / backward (fc_dc, fc_mi of a training forward) on onssen_linear_x3p."""
    import torch
    from onssen_amd.nn._train import linear_x3_forward, linear_x3_backward
    torch.manual_seed(3)
    M, K, N = 37, 24, 45
    x, w, b, dy = torch.randn(M, K), torch.randn(N, K), torch.randn(N), torch.randn(M, N)
    out = linear_x3_forward(lib, None, x, w, b)
    dx, dW, db = linear_x3_backward(lib, None, dy, x, w)
    for got, ref in ((out, x @ w.t() + b), (dx, dy @ w), (dW, dy.t() @ x), (db, dy.sum(0))):
        assert (got - ref).abs().max() <= 1e-4 * ref.abs().max()


def test_linear_x3_images_plain_bf16_products(lib):
    """ONSSEN_EPI_BF16 (opt-in precision mode): onssen_linear_x3p multiplies only the hi halves of the images -- the result
    equals the fp64 product of the bf16-ROUNDED operands (accumulation error only), and is bf16-grade (1e-3..1e-2) away
    from the product of the unrounded ones."""
    rng = np.random.default_rng(12)
    M, K, N = 70, 75, 170
    x, W, bias = rand(rng, M, K), rand(rng, N, K), rand(rng, N)
    KB = (K + 31) // 32
    a_img, w_img = np.zeros((M, KB, 2, 32), np.uint16), np.zeros((N, KB, 2, 32), np.uint16)
    lib.x3_image(P(x), K, 0, 1, M, K, P(a_img), None)
    lib.x3_image(P(W), K, 0, 1, N, K, P(w_img), None)
    out = np.full((M, N), np.nan, np.float32)
    lib.linear_x3p(P(a_img), M, K, P(w_img), P(bias), N, _abi.EPI_BIAS | _abi.EPI_BF16, 0, 0.0, P(out), 1, N, 0, None)
    xr, Wr = O.bf16_round(x).astype(np.float64), O.bf16_round(W).astype(np.float64)
    np.testing.assert_allclose(out, xr @ Wr.T + bias, atol=2e-5, rtol=1e-5)
    full = x.astype(np.float64) @ W.T.astype(np.float64) + bias
    assert 1e-4 < np.abs(out - full).max() < 0.2


@pytest.mark.parametrize("n", [4096, 4099])
def test_dropout_one_pass(lib, n):
    """onssen_dropout_f32 (nn.LSTM's inter-layer dropout, onssen/nn/deep_clustering.py:15-22): kept elements are x / (1 - p),
    the rest 0; the keep rate is 1 - p; the same seed gives the same mask (what the backward pass relies on), another seed
    another one; in place works; p = 0 is the identity."""
    rng = np.random.default_rng(3)
    x = np.abs(rand(rng, n)) + 0.5
    p = 0.3
    a, b, c = (np.full(n, np.nan, np.float32) for _ in range(3))
    lib.dropout(P(x), n, p, 1234, P(a), None)
    lib.dropout(P(x), n, p, 1234, P(b), None)
    lib.dropout(P(x), n, p, 1235, P(c), None)
    np.testing.assert_array_equal(a, b)
    kept = a != 0
    np.testing.assert_allclose(a[kept], x[kept] / np.float32(1 - p), rtol=1e-6)
    assert abs(kept.mean() - (1 - p)) < 0.03 and 0.3 < ((c != 0) == kept).mean() < 0.75
    # neighbouring elements are not correlated
    assert abs(np.corrcoef(kept[:-1], kept[1:])[0, 1]) < 0.06
    y = x.copy()
    lib.dropout(P(y), n, p, 1234, P(y), None)
    np.testing.assert_array_equal(y, a)
    lib.dropout(P(x), n, 0.0, 7, P(b), None)
    np.testing.assert_array_equal(b, x)


@pytest.mark.parametrize("rows,D", [(300, 20), (70, 4)])
def test_l2norm_rows_forward_and_gradient(lib, rows, D):
    """onssen_l2norm_rows_f32 / _grad_f32 against F.normalize and its autograd in float64 (incl. an all-zero row: y = 0, dx = g / eps
    -- what max(||x||, eps) gives)."""
    import torch
    rng = np.random.default_rng(9)
    x = rand(rng, rows, D); x[3] = 0.0
    g = rand(rng, rows, D)
    y, dx = np.full((rows, D), np.nan, np.float32), np.full((rows, D), np.nan, np.float32)
    lib.l2norm_rows(P(x), rows, D, 1e-12, P(y), None)
    lib.l2norm_rows_grad(P(x), P(g), rows, D, 1e-12, P(dx), None)
    xt = torch.from_numpy(x).double().requires_grad_(True)
    yt = torch.nn.functional.normalize(xt, p=2, dim=-1, eps=1e-12)
    (yt * torch.from_numpy(g).double()).sum().backward()
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=2e-6, atol=1e-7)
    ok = np.arange(rows) != 3
    np.testing.assert_allclose(dx[ok], xt.grad.numpy()[ok], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(dx[3], g[3] / np.float32(1e-12), rtol=1e-6)


def test_batched_gemm_with_two_outputs_and_overlapping_windows(lib):
    """onssen_linear_x3p_batched_split and _alt (the weight-gradient GEMMs of both directions in one launch): rows mapped
    4u + gate -> gate*H + u into two dense outputs; _alt: the problems read overlapping windows of ONE operand image
    [h_f | x | h_r] and their outputs change places."""
    rng = np.random.default_rng(31)
    H, Kx, TB = 12, 40, 70
    NP, N1 = 4 * H, Kx + H
    KB = (TB + 31) // 32
    dP = rand(rng, 2, NP, TB)                          # per direction: rows of the transposed pre-activation gradient
    hf, hr, x = rand(rng, H, TB), rand(rng, H, TB), rand(rng, Kx, TB)
    img = lambda m: (lambda o: (lib.x3_image(P(m), m.shape[1], 0, 1, m.shape[0], m.shape[1], P(o), None), o)[1])(
        np.zeros((m.shape[0], KB, 2, 32), np.uint16))
    a = np.stack([img(np.ascontiguousarray(dP[0])), img(np.ascontiguousarray(dP[1]))])
    zero = np.zeros(N1, np.float32)
    to_lstm = lambda m: m.reshape(H, 4, -1).transpose(1, 0, 2).reshape(4 * H, -1)      # packed row 4u + gate -> gate*H + u
    ref_ih = [to_lstm(dP[d].astype(np.float64) @ x.T.astype(np.float64)) for d in range(2)]
    ref_hh = [to_lstm(dP[d].astype(np.float64) @ (hf, hr)[d].T.astype(np.float64)) for d in range(2)]
    # plain: per direction [x | h_d]
    w = np.stack([img(np.concatenate([x, hf])), img(np.concatenate([x, hr]))])
    ih, hh = np.full((2, 4 * H, Kx), np.nan, np.float32), np.full((2, 4 * H, H), np.nan, np.float32)
    lib.linear_x3p_batched_split(P(a), NP * KB * 64, NP, TB, P(w), N1 * KB * 64, P(zero), N1, 4, P(ih), 4 * H * Kx, Kx, H * Kx, Kx,
                                 P(hh), 4 * H * H, H, H * H, 2, None)
    for d in range(2):
        np.testing.assert_allclose(ih[d], ref_ih[d], atol=2e-4, rtol=1e-4)
        np.testing.assert_allclose(hh[d], ref_hh[d], atol=2e-4, rtol=1e-4)
    # alternating: one image [h_f | x | h_r], windows H rows apart
    w1 = img(np.concatenate([hf, x, hr]))
    ih2, hh2 = np.full((2, 4 * H, Kx), np.nan, np.float32), np.full((2, 4 * H, H), np.nan, np.float32)
    lib.linear_x3p_batched_split_alt(P(a), NP * KB * 64, NP, TB, P(w1), H * KB * 64, P(zero), N1, 4, P(hh2), 4 * H * H, H, H * H, H,
                                     P(ih2), 4 * H * Kx, Kx, H * Kx, Kx, 2, None)
    np.testing.assert_array_equal(ih2, ih)
    np.testing.assert_array_equal(hh2, hh)


@pytest.mark.parametrize("H,Kx,T,B", [(16, 40, 9, 5), (24, 48, 5, 18), (16, 13, 7, 3)])      # (Kx = 13: a first layer's odd feature count)
def test_weight_gradients_from_row_major_images(lib, H, Kx, T, B):
    """onssen_lstm_wgrad_images_f32 (linear_x3t_kernel: contraction over the ROWS of row-major x3 images, fragments through the
    transposing LDS read, h_prev as a row shift of y) against onssen_linear_x3p_batched_split_alt on the transposed images:
    the same MFMA sequence on the same operand values -- bit for bit."""
    rng = np.random.default_rng(H + T)
    NP, Hp, K = 4 * H, H, T * B                          # (NP % 32 == 0 for these H)
    dP, y, x = rand(rng, K, 2 * NP), rand(rng, K, 2 * Hp), rand(rng, K, Kx)
    KB = (K + 31) // 32
    def rows_img(m):
        o = np.zeros((m.shape[0], (m.shape[1] + 31) // 32, 2, 32), np.uint16)
        lib.x3_image(P(m), m.shape[1], 0, 1, m.shape[0], m.shape[1], P(o), None)
        return o
    def t_img(m, cols, shift):                           # transposed image of m[:, cols] with the row shift of "h of the step before"
        sub = np.ascontiguousarray(m[:, cols])
        o = np.zeros((sub.shape[1], KB, 2, 32), np.uint16)
        lib.x3_image_t(P(sub), sub.shape[1], sub.shape[1], K, shift, P(o), None)
        return o
    # reference: the transposed-image route of the training path
    a_t = t_img(dP, slice(0, 2 * NP), 0)
    w1 = np.concatenate([t_img(y, slice(0, Hp), -B), t_img(x, slice(0, Kx), 0), t_img(y, slice(Hp, 2 * Hp), B)])
    zero = np.zeros(max(Hp + Kx, 8), np.float32)
    ih_a, hh_a = np.full((2, 4 * H, Kx), np.nan, np.float32), np.full((2, 4 * H, H), np.nan, np.float32)
    lib.linear_x3p_batched_split_alt(P(a_t), NP * KB * 64, NP, K, P(w1), Hp * KB * 64, P(zero), Hp + Kx, 4, P(hh_a), 4 * H * H, H, H * H, Hp,
                                     P(ih_a), 4 * H * Kx, Kx, H * Kx, Kx, 2, None)
    # the row-major route
    dp_img, y_img, x_img = rows_img(dP), rows_img(y), rows_img(x)
    ih_b, hh_b = np.full_like(ih_a, np.nan), np.full_like(hh_a, np.nan)
    lib.lstm_wgrad_images(P(dp_img), P(y_img), P(x_img), K, B, NP, Hp, Kx, P(zero), 4, P(ih_b), 4 * H * Kx, Kx, H * Kx,
                          P(hh_b), 4 * H * H, H, H * H, None)
    assert not np.isnan(ih_b).any() and not np.isnan(hh_b).any()
    np.testing.assert_array_equal(ih_b, ih_a)
    np.testing.assert_array_equal(hh_b, hh_a)
    # ... and it is the right product: nn.LSTM's row order gate*H + u from packed rows 4u + gate
    to_lstm = lambda m: m.reshape(H, 4, -1).transpose(1, 0, 2).reshape(4 * H, -1)
    hprev = np.zeros((K, Hp)); hprev[B:] = y[:-B, :Hp]
    np.testing.assert_allclose(hh_b[0], to_lstm(dP[:, :NP].T.astype(np.float64) @ hprev), atol=3e-4, rtol=1e-4)
    hnext = np.zeros((K, Hp)); hnext[:-B] = y[B:, Hp:]
    np.testing.assert_allclose(hh_b[1], to_lstm(dP[:, NP:].T.astype(np.float64) @ hnext), atol=3e-4, rtol=1e-4)
    np.testing.assert_allclose(ih_b[1], to_lstm(dP[:, NP:].T.astype(np.float64) @ x.astype(np.float64)), atol=3e-4, rtol=1e-4)


@pytest.mark.parametrize("K,M,N", [(70, 44, 40), (45, 300, 170)])
def test_plain_weight_gradient_from_row_major_images(lib, K, M, N):
    """onssen_linear_x3t: C = A^T W over the rows of two row-major x3 images (M, N not multiples of 8: the images' zero padding
    is read, never stored) -- bit for bit what onssen_linear_x3p gives on the transposed images."""
    rng = np.random.default_rng(K + M)
    A, W = rand(rng, K, M), rand(rng, K, N)
    def rows_img(m):
        o = np.zeros((m.shape[0], (m.shape[1] + 31) // 32, 2, 32), np.uint16)
        lib.x3_image(P(m), m.shape[1], 0, 1, m.shape[0], m.shape[1], P(o), None)
        return o
    def t_img(m):
        o = np.zeros((m.shape[1], (K + 31) // 32, 2, 32), np.uint16)
        lib.x3_image_t(P(m), m.shape[1], m.shape[1], K, 0, P(o), None)
        return o
    zero = np.zeros(max(M, N, 8), np.float32)
    ref = np.full((M, N), np.nan, np.float32)
    a_t, w_t = t_img(A), t_img(W)
    lib.linear_x3p(P(a_t), M, K, P(w_t), P(zero), N, _abi.EPI_BIAS, 0, 0.0, P(ref), 1, N, 0, None)
    got = np.full((M, N + 3), np.nan, np.float32)
    a_r, w_r = rows_img(A), rows_img(W)
    lib.linear_x3t(P(a_r), P(w_r), K, M, N, P(zero), P(got), N + 3, None)
    np.testing.assert_array_equal(got[:, :N], ref)
    assert np.isnan(got[:, N:]).all()
    np.testing.assert_allclose(ref, A.T.astype(np.float64) @ W.astype(np.float64), atol=3e-4, rtol=1e-4)


@pytest.mark.parametrize("M,tile", [(273, "256"), (150, "128")])
def test_embedding_head_with_norms_and_its_backward(lib, M, tile, monkeypatch):
    """onssen_linear_x3p_norms (fc_dc + F.normalize in one GEMM that also leaves 1 / max(||.||, eps) per bin) and
    onssen_l2norm_rows_grad_y_f32 (the normalisation's backward from its output) against F.normalize(F.linear(.)) under float64
    autograd; a bin whose raw product is exactly zero takes the clamped branch (dx = g / eps)."""
    import torch
    monkeypatch.setenv("ONSSEN_X3Q_BM", tile)
    rng = np.random.default_rng(21)
    K, D, N = 75, 20, 440
    x, W, bias, g = rand(rng, M, K), rand(rng, N, K), rand(rng, N), rand(rng, M, N)
    W[40:60] = 0.0; bias[40:60] = 0.0                          # bin 2 of every row: raw product 0
    KB = (K + 31) // 32
    a_img, w_img = np.zeros((M, KB, 2, 32), np.uint16), np.zeros((N, KB, 2, 32), np.uint16)
    lib.x3_image(P(x), K, 0, 1, M, K, P(a_img), None)
    lib.x3_image(P(W), K, 0, 1, N, K, P(w_img), None)
    e, inv = np.full((M, N), np.nan, np.float32), np.full((M, N // D), np.nan, np.float32)
    lib.linear_x3p_norms(P(a_img), M, K, P(w_img), P(bias), N, D, 1e-12, P(e), P(inv), None)
    xt = torch.from_numpy(x).double()
    raw = (xt @ torch.from_numpy(W).double().t() + torch.from_numpy(bias).double()).reshape(M, N // D, D).requires_grad_(True)
    et = torch.nn.functional.normalize(raw, p=2, dim=-1, eps=1e-12)
    (et * torch.from_numpy(g).double().reshape(M, N // D, D)).sum().backward()
    nrm = raw.detach().norm(dim=-1).numpy()
    ok = np.ones(N // D, bool); ok[2] = False
    err = np.abs(e.reshape(M, N // D, D) - et.detach().numpy()) * np.maximum(nrm, 1e-12)[..., None]
    assert not np.isnan(e).any() and err.max() <= 3e-4
    np.testing.assert_allclose(inv[:, ok], 1.0 / nrm[:, ok], rtol=3e-4)
    assert np.all(inv[:, 2] == np.float32(1e12)) and not e[:, 40:60].any()
    d_raw = np.full((M, N), np.nan, np.float32)
    lib.l2norm_rows_grad_y(P(e), P(inv), P(g), M * (N // D), D, 1e-12, P(d_raw), None)
    d = d_raw.reshape(M, N // D, D)
    ref = raw.grad.numpy()
    # (a short vector's gradient is large and as uncertain as its direction: error relative to 1 / ||x||)
    np.testing.assert_allclose(d[:, ok] * nrm[:, ok, None], ref[:, ok] * nrm[:, ok, None], atol=2e-3, rtol=2e-3)
    np.testing.assert_allclose(d[:, 2], g.reshape(M, N // D, D)[:, 2] * np.float32(1e12), rtol=1e-6)


@pytest.mark.parametrize("B,T,F,C", [(2, 40, 13, 2), (3, 33, 9, 3)])
def test_dc_head_gradient_straight_into_gemm_operands(lib, B, T, F, C):
    """onssen_dc_head_grad_images_f32 (train-step-level fusion): from the normalised embedding + reciprocal norms + the forward's
    partial Grams to the row-major image, the transposed image and the block column sums of d(loss_dc)/d(fc_dc output) in one
    pass -- against the chain it replaces (onssen_loss_dc_grad_f32 -> onssen_l2norm_rows_grad_y_f32 ->
    onssen_x3_image_both_colsum_f32).  T is not a multiple of 32: row tiles span two utterances."""
    rng = np.random.default_rng(B + T)
    D, N, M = 20, F * 20, B * T
    raw = rand(rng, M, F, D)
    raw[5, 2] = 0.0                                              # a clamped norm
    nrm = np.maximum(np.linalg.norm(raw, axis=-1), 1e-12).astype(np.float32)
    emb = (raw / nrm[..., None]).astype(np.float32).reshape(M, N)
    inv = (1.0 / nrm).astype(np.float32)
    lab = rng.integers(0, C + 1, (B, T * F))                     # C = silent bin
    one_hot = np.zeros((B, T * F, C), np.float32)
    for c in range(C):
        one_hot[..., c] = lab == c
    mag = (rng.random((B, T * F)) + 0.1).astype(np.float32)
    g = (rand(rng, B) + 2.0).astype(np.float32)
    embv = np.ascontiguousarray(emb.reshape(B, T * F, D))
    per_utt, total = np.zeros(B, np.float32), np.zeros(B, np.float32)
    ws = aligned_f32(lib.loss_dc_workspace_bytes(B) // 4 + 64)
    lib.loss_dc(P(embv), P(one_hot), P(mag), B, T * F, D, C, P(per_utt), P(total), P(ws), ws.nbytes, None)
    # the chain
    dV = np.full((B, T * F, D), np.nan, np.float32)
    lib.loss_dc_grad(P(embv), P(one_hot), P(mag), B, T * F, D, C, P(g), P(dV), P(ws), ws.nbytes, None)
    d_raw = np.full((M, N), np.nan, np.float32)
    lib.l2norm_rows_grad_y(P(emb), P(inv), P(dV), M * F, D, 1e-12, P(d_raw), None)
    KB, NB = (M + 31) // 32, (N + 31) // 32
    rows_a, t_a, part_a = np.zeros((M, NB, 2, 32), np.uint16), np.zeros((N, KB, 2, 32), np.uint16), np.zeros((KB, N), np.float32)
    lib.x3_image_both_colsum(P(d_raw), N, N, M, P(rows_a), P(t_a), P(part_a), None)
    # the fused pass
    rows_b, t_b = np.full_like(rows_a, 0x7fc0), np.full_like(t_a, 0x7fc0)
    part_b = np.full((KB, N), np.nan, np.float32)
    lib.dc_head_grad_images(P(emb), P(inv), P(one_hot), P(mag), B, T, F, D, C, 1e-12, P(g), P(ws), ws.nbytes, P(rows_b), P(t_b),
                            P(part_b), None)
    f = lambda u: (u.astype(np.uint32) << 16).view(np.float32)
    val = lambda img: (f(img[:, :, 0]).astype(np.float64) + f(img[:, :, 1])).reshape(img.shape[0], -1)
    scale = np.abs(d_raw).max()
    ra, rb, ta, tb = val(rows_a), val(rows_b), val(t_a), val(t_b)
    assert np.abs(ra[:, :N] - d_raw).max() <= 2e-5 * scale       # (the images carry the values to 2^-16)
    np.testing.assert_allclose(rb, ra, atol=3e-5 * scale)        # same arithmetic, a different summation order of the row dot product
    np.testing.assert_allclose(tb, ta, atol=3e-5 * scale)
    assert not rb[:, N:].any() and not tb[:, M:].any()           # zero padding of both images
    np.testing.assert_allclose(part_b, part_a, atol=2e-4 * scale)
    with pytest.raises(_abi.OnssenError):                        # row tiles may span two utterances, not more
        lib.dc_head_grad_images(P(emb), P(inv), P(one_hot), P(mag), B * 2, T // 2, F, D, C, 1e-12, P(g), P(ws), ws.nbytes, P(rows_b),
                                P(t_b), P(part_b), None)


@pytest.mark.parametrize("M,C", [(300, 40), (129, 7)])
def test_bn_rows_train_forward_and_gradient(lib, M, C):
    """onssen_bn_rows_train_f32 / _grad_f32 against nn.BatchNorm1d in training mode under float64 autograd, applied the
    reference's way -- to the (B, C, T) permutation of the same rows (onssen/nn/deep_clustering.py:36-38)."""
    import torch
    rng = np.random.default_rng(12)
    x = rand(rng, M, C) * 0.7 + 0.3
    gamma, beta, g = rand(rng, C) + 1.5, rand(rng, C), rand(rng, M, C)
    y, dx = np.full((M, C), np.nan, np.float32), np.full((M, C), np.nan, np.float32)
    mean, invstd, dgamma, dbeta = (np.full(C, np.nan, np.float32) for _ in range(4))
    ws = aligned_f32(lib.bn_rows_workspace_bytes(M, C) // 4 + 64)
    lib.bn_rows_train(P(x), M, C, P(gamma), P(beta), 1e-5, P(y), P(mean), P(invstd), P(ws), ws.nbytes, None)
    lib.bn_rows_grad(P(x), P(g), M, C, P(gamma), P(mean), P(invstd), P(dx), P(dgamma), P(dbeta), P(ws), ws.nbytes, None)
    bn = torch.nn.BatchNorm1d(C).double().train()
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(gamma)); bn.bias.copy_(torch.from_numpy(beta))
    xt = torch.from_numpy(x).double().requires_grad_(True)
    B = 3 if M % 3 == 0 else 1
    yt = bn(xt.reshape(B, M // B, C).permute(0, 2, 1)).permute(0, 2, 1).reshape(M, C)
    (yt * torch.from_numpy(g).double()).sum().backward()
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(mean, x.astype(np.float64).mean(0), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(1 / invstd ** 2 - 1e-5, x.astype(np.float64).var(0), rtol=2e-5)
    np.testing.assert_allclose(bn.running_var.numpy(), 0.9 + 0.1 * (1 / invstd.astype(np.float64) ** 2 - 1e-5) * M / (M - 1), rtol=2e-5)
    np.testing.assert_allclose(dbeta, bn.bias.grad.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(dgamma, bn.weight.grad.numpy(), rtol=2e-5, atol=5e-5)
    assert np.abs(dx - xt.grad.numpy()).max() <= 3e-5 * max(np.abs(xt.grad.numpy()).max(), 1e-3)
    # channels whose |mean| is far larger than their spread -- E[x^2] - mean^2 in fp32 cancels (relative error
    # ~1e-7 * mean^2 / var = 1e-1 here); the shifted strip sums merged with Chan's formula do not
    x2 = (rand(rng, M, C) * 1e-2 + 10.0 + np.arange(C, dtype=np.float32)[None, :]).astype(np.float32)
    lib.bn_rows_train(P(x2), M, C, P(gamma), P(beta), 0.0, P(y), P(mean), P(invstd), P(ws), ws.nbytes, None)
    np.testing.assert_allclose(mean, x2.astype(np.float64).mean(0), rtol=1e-6)
    np.testing.assert_allclose(1 / invstd.astype(np.float64) ** 2, x2.astype(np.float64).var(0), rtol=2e-4)


@pytest.mark.parametrize("H,ug,in_dim,bidir", [(10, 4, 9, 0), (24, 8, 129, 0), (24, 8, 48, 1), (40, 20, 80, 1), (30, 4, 60, 1)])
def test_wih_pack_with_image_in_one_pass(lib, H, ug, in_dim, bidir):
    """onssen_lstm_pack_wih_image_f32 (round 5: the training forward's per-step pack) == onssen_lstm_pack_f32's packed W_ih and bias
    + onssen_x3_image_f32 of the packed matrix, bit for bit."""
    rng = np.random.default_rng(H + in_dim)
    Hp, NP, KQ, we = lib.lstm_geometry(H, ug)
    Kp = 2 * Hp if bidir else (in_dim + 3) // 4 * 4
    K = 2 * Hp if bidir else in_dim
    KB = (K + 31) // 32
    w_ih, w_hh = rand(rng, 4 * H, in_dim), rand(rng, 4 * H, H)
    b_ih, b_hh = rand(rng, 4 * H), rand(rng, 4 * H)
    a0, b0, c0 = np.full((NP, Kp), np.nan, np.float32), np.zeros(we, np.float32), np.full(NP, np.nan, np.float32)
    lib.lstm_pack(P(w_ih), P(w_hh), P(b_ih), P(b_hh), in_dim, bidir, H, ug, P(a0), P(b0), P(c0), None)
    img0 = np.zeros((NP, KB, 2, 32), np.uint16)
    lib.x3_image(P(a0), Kp, 0, 1, NP, K, P(img0), None)
    a1, c1 = np.full((NP, Kp), np.nan, np.float32), np.full(NP, np.nan, np.float32)
    img1 = np.full((NP, KB, 2, 32), 0xffff, np.uint16)
    lib.lstm_pack_wih_image(P(w_ih), P(b_ih), P(b_hh), in_dim, bidir, H, ug, P(a1), P(c1), P(img1), None)
    assert np.array_equal(a0, a1) and np.array_equal(c0, c1) and np.array_equal(img0, img1)


@pytest.mark.parametrize("max_norm", [5.0, 1e9, 0.0])
def test_clip_adam_against_the_formulas(lib, max_norm):
    """onssen_clip_adam_f32 (round 5): torch.nn.utils.clip_grad_norm_ + torch.optim.Adam's update restated in NumPy fp64, three steps
    over tensors of awkward sizes (one larger than a chunk, one unaligned view): clipping active (5.0), inactive (1e9) and off (0)."""
    rng = np.random.default_rng(3)
    sizes = [7, 4800, 32768 + 13, 129 * 20]
    raw = [aligned_f32(n + 1) for n in sizes]
    p = [r[:n] for r, n in zip(raw, sizes)]
    p[1] = raw[1][1:4801]                                   # 4-byte aligned only: the scalar path
    for a in p:
        a[...] = rand(rng, a.size)
    g = [rand(rng, n) * (3.0 if i == 2 else 0.5) for i, n in enumerate(sizes)]
    m = [np.zeros(n, np.float32) for n in sizes]
    v = [np.zeros(n, np.float32) for n in sizes]
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    P64, M64, V64 = [a.astype(np.float64) for a in p], [a.astype(np.float64) for a in m], [a.astype(np.float64) for a in v]
    nb = lib.clip_adam_workspace_bytes(sizes)
    ws = aligned_f32(nb // 4 + 4)
    for step in (1, 2, 3):
        for a in g:
            a[...] = rand(rng, a.size) * 2.0
        G64 = [a.astype(np.float64) for a in g]
        norm = np.sqrt(sum((a * a).sum() for a in G64))
        coef = min(1.0, max_norm / (norm + 1e-6)) if 0 < max_norm < np.inf else 1.0
        for i in range(len(sizes)):
            gg = G64[i] * coef
            M64[i] += (gg - M64[i]) * (1 - b1)
            V64[i] = V64[i] * b2 + (1 - b2) * gg * gg
            P64[i] -= lr / (1 - b1 ** step) * M64[i] / (np.sqrt(V64[i]) / np.sqrt(1 - b2 ** step) + eps)
        keep = [a.copy() for a in g]
        lib.clip_adam([P(a) for a in p], [P(a) for a in g], [P(a) for a in m], [P(a) for a in v], sizes, max_norm, lr, b1, b2, eps,
                      step, P(ws), nb, None)
        assert all(np.array_equal(a, k) for a, k in zip(g, keep))          # gradients untouched unless asked for
        if 0 < max_norm < np.inf:
            np.testing.assert_allclose(ws[0], norm, rtol=2e-6)
        for i in range(len(sizes)):
            np.testing.assert_allclose(p[i], P64[i], rtol=2e-6, atol=2e-7)
            np.testing.assert_allclose(m[i], M64[i], rtol=2e-6, atol=2e-7)      # (g - m cancels: fp32 round-off of O(|g|) terms)
            np.testing.assert_allclose(v[i], V64[i], rtol=2e-6, atol=1e-12)
    if 0 < max_norm < 1e6:                                  # write_grads: the clipped gradients land in g
        lib.clip_adam([P(a) for a in p], [P(a) for a in g], [P(a) for a in m], [P(a) for a in v], sizes, max_norm, lr, b1, b2, eps,
                      4, P(ws), nb, None, write_grads=True)
        norm = np.sqrt(sum((a.astype(np.float64) ** 2).sum() for a in keep))
        for a, k in zip(g, keep):
            np.testing.assert_allclose(a, k * min(1.0, max_norm / (norm + 1e-6)), rtol=2e-6)


@pytest.mark.parametrize("H,ug,F,L,with_r", [(10, 4, 9, 2, True), (24, 8, 33, 3, True), (40, 20, 20, 1, False), (12, 4, 7, 5, True)])
def test_training_repack_of_the_whole_stack_in_one_launch(lib, H, ug, F, L, with_r):
    """onssen_lstm_pack_train_f32 (round 5) == per (layer, direction) onssen_lstm_pack_wih_image_f32 + onssen_lstm_pack_whh_bf16x3 +
    onssen_lstm_pack_whhR_bf16x3, bit for bit (L = 5 with the row-slice images needs two launches)."""
    rng = np.random.default_rng(H + L)
    Hp, NP, KQ, we = lib.lstm_geometry(H, ug)
    _, _, we3 = lib.lstm_geometry_x3(H, ug)
    nR = lib.lstm_whhR_elems(H, ug)
    keep, ptr = [], {k: [] for k in ("w_ih", "w_hh", "b_ih", "b_hh", "a", "c", "ai", "b3", "r")}
    ref = []
    for l in range(L):
        in_l = F if l == 0 else 2 * H
        Kp = (F + 3) // 4 * 4 if l == 0 else 2 * Hp
        K = F if l == 0 else 2 * Hp
        KB = (K + 31) // 32
        for d in range(2):
            w_ih, w_hh, b_ih, b_hh = rand(rng, 4 * H, in_l), rand(rng, 4 * H, H), rand(rng, 4 * H), rand(rng, 4 * H)
            a1, c1 = np.full((NP, Kp), np.nan, np.float32), np.full(NP, np.nan, np.float32)
            i1, h1, r1 = np.full((NP, KB, 2, 32), 7, np.uint16), np.full(we3, 7, np.uint16), np.full(nR, 7, np.uint16)
            lib.lstm_pack_wih_image(P(w_ih), P(b_ih), P(b_hh), in_l, 1 if l else 0, H, ug, P(a1), P(c1), P(i1), None)
            lib.lstm_pack_whh_bf16x3(P(w_hh), H, ug, P(h1), None)
            lib.lstm_pack_whhR_bf16x3(P(w_hh), H, ug, P(r1), None)
            a2, c2 = np.full((NP, Kp), np.nan, np.float32), np.full(NP, np.nan, np.float32)
            i2, h2, r2 = np.full((NP, KB, 2, 32), 9, np.uint16), np.full(we3, 9, np.uint16), np.full(nR, 9, np.uint16)
            keep += [w_ih, w_hh, b_ih, b_hh]
            for k, v in (("w_ih", w_ih), ("w_hh", w_hh), ("b_ih", b_ih), ("b_hh", b_hh), ("a", a2), ("c", c2), ("ai", i2), ("b3", h2), ("r", r2)):
                ptr[k].append(P(v))
            ref.append(((a1, c1, i1, h1, r1), (a2, c2, i2, h2, r2)))
    lib.lstm_pack_train(L, F, H, ug, ptr["w_ih"], ptr["w_hh"], ptr["b_ih"], ptr["b_hh"], ptr["a"], ptr["c"], ptr["ai"], ptr["b3"],
                        ptr["r"] if with_r else None, None)
    for one, merged in ref:
        for k, (x, y) in enumerate(zip(one, merged)):
            if k == 4 and not with_r:
                assert (y == 9).all()
            else:
                assert np.array_equal(x, y), k


def test_feature_helper_kernels_match_the_reference_fixture(lib, golden_dir):
    """onssen_log_magnitude_f32 / onssen_cos_difference_f32 / onssen_one_hot_f32 (the kernels behind
    onssen_amd.data.feature_utils) against the REFERENCE's get_log_magnitude / get_cos_difference / get_one_hot outputs
    (tests/golden/g3_features.npz, tools/gen_golden_features.py; onssen/data/feature_utils.py:49-51,77-95)."""
    z = np.load(f"{golden_dir}/g3_features.npz")
    for tag in ("a", "b"):
        X, S1, S2 = (np.ascontiguousarray(z[f"{tag}_{k}"]) for k in ("X", "S1", "S2"))
        n = X.size
        for eps, key in ((1e-7, "log_magnitude"), (1e-3, "log_magnitude_eps3")):
            out = np.full(X.shape, np.nan, np.float32)
            lib.log_magnitude(P(X.view(np.float32)), n, eps, P(out), None)
            np.testing.assert_allclose(10.0 ** out.astype(np.float64), 10.0 ** z[f"{tag}_{key}"].astype(np.float64), rtol=2e-6, atol=1e-9)
        big = (np.abs(X) > 1e-3) & (np.abs(S1) > 1e-3) & (np.abs(S2) > 1e-3)
        for S, key in ((S1, "cos_s1"), (S2, "cos_s2")):
            out = np.full(X.shape, np.nan, np.float32)
            lib.cos_difference(P(X.view(np.float32)), P(S.view(np.float32)), n, P(out), None)
            np.testing.assert_allclose(out[big], z[f"{tag}_{key}"][big], atol=1e-5)
            assert np.all(np.abs(out) <= 1.0 + 1e-6)
        feat, m1, m2 = z[f"{tag}_log_magnitude"], np.abs(S1), np.abs(S2)
        for db in (40, 20):
            out, umax = np.full(X.shape + (2,), np.nan, np.float32), np.zeros(1, np.float32)
            lib.one_hot(P(feat), P(m1), P(m2), 1, n, float(db), P(umax), P(out), None)
            assert umax[0] == feat.max()
            np.testing.assert_array_equal(out.astype(np.float64), z[f"{tag}_one_hot_{db}"])   # magnitudes in: no tie tolerance needed
    # two utterances in one call: each thresholded against its own maximum
    fa, fb = z["a_log_magnitude"], z["a_log_magnitude"] - 1.0
    feat2 = np.ascontiguousarray(np.stack([fa, fb]))
    m1, m2 = np.abs(z["a_S1"]), np.abs(z["a_S2"])
    out, umax = np.full((2,) + fa.shape + (2,), np.nan, np.float32), np.zeros(2, np.float32)
    m1x2, m2x2 = np.ascontiguousarray(np.stack([m1, m1])), np.ascontiguousarray(np.stack([m2, m2]))
    lib.one_hot(P(feat2), P(m1x2), P(m2x2), 2, fa.size, 40.0, P(umax), P(out), None)
    np.testing.assert_array_equal(out[0], out[1])
    np.testing.assert_array_equal(out[0].astype(np.float64), z["a_one_hot_40"])
    assert lib.dll.onssen_log_magnitude_f32(None, 4, 1e-7, None, None) == -1 and lib.dll.onssen_one_hot_f32(P(fa), P(m1), P(m2), 0, 4, 40.0, P(umax), P(out), None) == -1


def test_param_guard_kernel(lib):
    """onssen_param_guard_u32 (csrc/optim.inc): mode 0 stores the sampled folds, mode 1 raises the sticky flag only when a tensor
    changed; first / last elements are always sampled; an optimizer-like update of every element is always seen."""
    import ctypes
    rng = np.random.default_rng(21)
    tensors = [rng.standard_normal(n).astype(np.float32) for n in (1, 7, 2048, 2049, 100_003)]
    ptrs = np.array([t.ctypes.data for t in tensors], np.int64)
    numel = np.array([t.size for t in tensors], np.int64)
    ref, flag = np.zeros(len(tensors), np.uint32), np.zeros(1, np.uint32)
    run = lambda mode: lib.param_guard(P(ptrs), P(numel), len(tensors), 2048, mode, P(ref), P(flag), None)
    run(0)
    assert len(set(ref.tolist())) == len(tensors)
    run(1)
    assert flag[0] == 0
    for t, idx in ((4, 0), (4, tensors[4].size - 1), (1, 3), (0, 0)):       # first / last element of a long tensor, any element of a short one
        keep = tensors[t][idx]
        tensors[t][idx] = np.nextafter(keep, np.float32(10.0))               # one ulp
        run(1)
        assert flag[0] == 1, (t, idx)
        tensors[t][idx] = keep
        flag[0] = 0
        run(1)
        assert flag[0] == 0
    tensors[4] *= np.float32(1.0 - 1e-3)                                     # what an optimizer step does: every element moves a little
    run(1)
    assert flag[0] == 1
    run(0)                                                                   # re-arm on the new values; the flag is sticky until the owner clears it
    assert flag[0] == 1
    flag[0] = 0
    run(1)
    assert flag[0] == 0
    a, b = tensors[2][0], tensors[2][2047]                                   # two sampled elements swapped: the fold is position dependent
    tensors[2][0], tensors[2][2047] = b, a
    run(1)
    assert flag[0] == 1
    assert lib.dll.onssen_param_guard_u32(None, None, 1, 16, 0, None, None, None) == -1
