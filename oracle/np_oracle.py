"""NumPy restatement of the reference hot path (oracle; test infrastructure only).

Every function cites the reference lines it restates (paths relative to the
upstream tree mounted read-only at /root/reference in the build container).

Pinning status
--------------
* Network rows (A4-A10: LSTM stack, BatchNorm, Linear heads, normalise,
  sigmoid): PINNED.  ``tools/gen_golden.py`` imports the reference's own
  ``onssen.nn`` modules on torch-CPU in the build container, loads this
  repo's deterministic weights into them and commits inputs + outputs under
  ``tests/golden/``; ``tests/test_oracle.py`` checks this file against them.
* Front/back end (A1-A3, A11: STFT / log-magnitude / phase / mask + iSTFT):
  **parity unpinned** by the reference.  The arithmetic lives in the
  third-party package ``librosa`` (un-vendored, version un-pinned; the call
  signatures used at onssen/data/feature_utils.py:19,39-44 imply < 0.10) and
  the reference holds no test or golden vector for it.  The restatement
  below follows librosa 0.7/0.8 ``core.stft`` / ``core.istft`` semantics and
  is cross-checked against ``torch.stft`` / ``torch.istft`` (an independent
  implementation) in ``tests/test_oracle.py``.
"""
import numpy as np

# --------------------------------------------------------------------------
# Front end  (onssen/data/feature_utils.py)
# --------------------------------------------------------------------------

def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True) (what librosa's
    default window='hann' resolves to), float64."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def stft(sig, n_fft, hop):
    """onssen/data/feature_utils.py:20  ``np.transpose(librosa.core.stft(sig,
    n_fft=window_size, hop_length=hop_size))``.

    librosa<0.10 defaults: win_length=n_fft, window='hann' (periodic),
    center=True, pad_mode='reflect'; frames = 1 + len(sig)//hop; the windowed
    frames are float64 (float32 signal x float64 window), rfft in float64,
    result stored as complex64.  Returns (T, F) complex64.
    """
    sig = np.asarray(sig, dtype=np.float32)
    pad = n_fft // 2
    y = np.pad(sig, pad, mode="reflect")
    n_frames = 1 + (len(y) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = y[idx].astype(np.float64) * hann_periodic(n_fft)[None, :]
    return np.fft.rfft(frames, axis=1).astype(np.complex64)


def log_magnitude(spec, epsilon=1e-7):
    """onssen/data/feature_utils.py:49-51  ``np.log10(np.abs(stft)+epsilon)``
    -> float32 for a complex64 input."""
    return np.log10(np.abs(spec) + np.float32(epsilon)).astype(np.float32)


def phase_re_im(spec):
    """onssen/data/feature_utils.py:54-64 get_phase: stack (Re, Im) on a new
    last axis -> (T, F, 2) float32.  (Raw parts, not a unit phasor.)"""
    return np.stack([np.real(spec), np.imag(spec)], axis=-1).astype(np.float32)


def cos_difference(stft_1, stft_2):
    """onssen/data/feature_utils.py:67-80: cos(angle(stft_1) - angle(stft_2))."""
    return np.cos(np.angle(stft_1) - np.angle(stft_2))


def one_hot_labels(feature_mix, mag_s1, mag_s2, db_threshold):
    """onssen/data/feature_utils.py:83-95 get_one_hot: e_argmax over the two source magnitudes (first
    maximum on ties), all-zero on bins with feature < max(feature) - db_threshold/20.  float64 (T,F,2)."""
    vals = np.argmax(np.asarray([mag_s1, mag_s2]), axis=0)
    Y = np.zeros(mag_s1.shape + (2,))
    Y[vals == 0, 0] = 1
    Y[vals == 1, 1] = 1
    Y[feature_mix < (np.max(feature_mix) - db_threshold / 20)] = 0
    return Y


# --------------------------------------------------------------------------
# Back end  (egs/wsj0-2mix/*/evaluate.py + librosa.core.istft)
# --------------------------------------------------------------------------

def istft(spec_tf, hop, length):
    """``librosa.core.istft(stft_est[i].T, hop_length=hop, length=nsample)``
    (egs/wsj0-2mix/deep_clustering/evaluate.py:45,
    egs/wsj0-2mix/chimera/evaluate.py:43).

    spec_tf: (T, F) complex.  n_fft = 2(F-1); periodic-Hann synthesis window;
    overlap-add; divide by the window sum-of-squares where it exceeds
    ``tiny``; drop the first n_fft//2 samples; fix length.  float64 result.

    One deliberate difference from librosa 0.7/0.8: its ``istft`` overlap-adds the frames and the window sum in ``dtype``
    = float32 by default; this restatement accumulates in float64 (the HIP kernel stores float32 frames and accumulates
    the <= 4 overlapping ones in float64).  The two differ by ~1e-7 of the signal scale -- far inside the 2e-6 tolerance
    of the parity tests, but it is why those tests do not ask for bit equality.  Cross-checked against torch.istft and
    scipy.signal.istft in tests/test_oracle.py (librosa itself is not installable here: parity unpinned).
    """
    spec_tf = np.asarray(spec_tf)
    T, F = spec_tf.shape
    n_fft = 2 * (F - 1)
    w = hann_periodic(n_fft)
    exp_len = n_fft + hop * (T - 1)
    y = np.zeros(exp_len, dtype=np.float64)
    wss = np.zeros(exp_len, dtype=np.float64)
    frames = np.fft.irfft(spec_tf.astype(np.complex128), n=n_fft, axis=1)
    for t in range(T):
        y[t * hop:t * hop + n_fft] += w * frames[t]
        wss[t * hop:t * hop + n_fft] += w * w
    nz = wss > np.finfo(np.float64).tiny
    y[nz] /= wss[nz]
    y = y[n_fft // 2:]
    if len(y) >= length:
        return y[:length]
    return np.pad(y, (0, length - len(y)))


def istft_f32_ola(spec_tf, hop, length):
    """The same inverse transform with librosa 0.7 / 0.8's dtype rule followed LITERALLY (``istft(..., dtype=np.float32)``, the
    default the reference's call sites take): ``numpy.fft.irfft`` yields float64 frames, ``ifft_window * irfft`` stays float64,
    but the output buffer ``y`` and the window sum-of-squares are float32 arrays, so every ``y[s:s+n_fft] += ytmp[:, frame]`` of
    the overlap-add rounds to float32, ``window_sumsquare(..., dtype=float32)`` accumulates ``w**2`` in float32, and the final
    ``y[nz] /= wss[nz]`` is a float32 division.  Returns float32.  ``istft`` above is the same arithmetic with float64
    accumulation; the two differ by a few float32 ulps of the signal scale, and the HIP kernel (float32 frames, the <= 4
    overlapping ones added in float64) must lie within that distance of BOTH (tests/test_oracle.py, tests/test_gpu_parity.py)."""
    spec_tf = np.asarray(spec_tf)
    T, F = spec_tf.shape
    n_fft = 2 * (F - 1)
    w = hann_periodic(n_fft)
    exp_len = n_fft + hop * (T - 1)
    y = np.zeros(exp_len, dtype=np.float32)
    wss = np.zeros(exp_len, dtype=np.float32)
    frames = w[None, :] * np.fft.irfft(spec_tf.astype(np.complex64), n=n_fft, axis=1)     # float64, like numpy.fft in librosa
    win_sq = (w * w)
    for t in range(T):
        y[t * hop:t * hop + n_fft] += frames[t]                  # float32 += float64 -> rounds to float32 each time
        wss[t * hop:t * hop + n_fft] += win_sq
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    y = y[n_fft // 2:]
    if len(y) >= length:
        return y[:length]
    return np.pad(y, (0, length - len(y)))


def mask_istft(stft_mix, masks, hop, length):
    """egs/wsj0-2mix/chimera/evaluate.py:34-43 (and deep_clustering/
    evaluate.py:31-45 after the masks are built): ``stft_est = stft_mix *
    mask``; one istft per speaker.  stft_mix (T,F) complex64, masks (C,T,F)
    real.  Returns (C, length) float64."""
    stft_mix = np.asarray(stft_mix)
    masks = np.asarray(masks, dtype=np.float64)
    return np.stack([istft(stft_mix * masks[c], hop, length)
                     for c in range(masks.shape[0])])


def dc_active_bins(feature_mix, db=40.0):
    """egs/wsj0-2mix/deep_clustering/evaluate.py:36-37: bins with
    ``feature >= max(feature) - 40/20`` take part in the clustering."""
    return feature_mix >= (np.max(feature_mix) - db / 20.0)


def dc_binary_masks(feature_mix, labels_active):
    """egs/wsj0-2mix/deep_clustering/evaluate.py:36-41: mask[0]=label,
    mask[1]=1-label on active bins, 0 on silent bins in both masks."""
    act = dc_active_bins(feature_mix)
    mask = np.zeros((2,) + feature_mix.shape, dtype=np.float64)
    mask[0, act] = labels_active
    mask[1, act] = 1 - labels_active
    return mask


# --------------------------------------------------------------------------
# Network  (onssen/nn/*.py; PyTorch nn.LSTM / BatchNorm1d / Linear semantics)
# --------------------------------------------------------------------------

def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def bf16_round(a):
    """float32 -> nearest bfloat16 (ties to even), returned as float32: what the opt-in ONSSEN_PRECISION=bf16 mode
    feeds the matrix cores (not a reference function; used to restate that mode's arithmetic for its tests)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse, rnd=None, exact_tail=False):
    """One direction of one nn.LSTM layer (batch_first, zero initial state).
    Gate row order in the 4H axis is i, f, g, o; the two bias vectors add.
    x: (B, T, In) -> (B, T, H).  (onssen/nn/deep_clustering.py:15-22,35.)
    ``rnd`` (default None = the reference's arithmetic) rounds the operands of the two matrix products
    (``exact_tail``: all input columns but the last)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    dt = x.dtype
    if rnd is not None and exact_tail:
        # the product's fused first layer (in_dim = 32k + 1, B > 16) forms the lone last input column as an fp32 rank-1
        # update outside the matrix cores: that column is NOT rounded
        gx = rnd(x[..., :-1]) @ rnd(w_ih[:, :-1]).T + x[..., -1:] * w_ih[:, -1] + (b_ih + b_hh)
        w_hh = rnd(w_hh)
    else:
        if rnd is not None:
            x, w_ih, w_hh = rnd(x), rnd(w_ih), rnd(w_hh)
        gx = x @ w_ih.T + (b_ih + b_hh)
    h = np.zeros((B, H), dtype=dt)
    c = np.zeros((B, H), dtype=dt)
    out = np.empty((B, T, H), dtype=dt)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = gx[:, t] + (h if rnd is None else rnd(h)) @ w_hh.T
        i = _sigmoid(g[:, 0:H])
        f = _sigmoid(g[:, H:2 * H])
        gg = np.tanh(g[:, 2 * H:3 * H])
        o = _sigmoid(g[:, 3 * H:4 * H])
        c = f * c + i * gg
        h = o * np.tanh(c)
        out[:, t] = h
    return out


def blstm_stack(x, sd, prefix, num_layers, collect=None, rnd=None, exact_tail0=False):
    """nn.LSTM(bidirectional=True, batch_first=True) in eval mode (inter-layer
    dropout inactive): layer output = [forward | reverse] on the last axis,
    which feeds the next layer.  ``sd`` is a reference-layout state_dict of
    numpy arrays: {prefix}weight_ih_l{k}[_reverse] ..."""
    for k in range(num_layers):
        outs = []
        for sfx, rev in (("", False), ("_reverse", True)):
            outs.append(lstm_direction(
                x,
                sd[f"{prefix}weight_ih_l{k}{sfx}"], sd[f"{prefix}weight_hh_l{k}{sfx}"],
                sd[f"{prefix}bias_ih_l{k}{sfx}"], sd[f"{prefix}bias_hh_l{k}{sfx}"], rev, rnd,
                exact_tail=exact_tail0 and k == 0))
        x = np.concatenate(outs, axis=-1)
        if collect is not None:
            collect.append(x)
    return x


def batchnorm_eval(x, sd, prefix, eps=1e-5):
    """nn.BatchNorm1d(2H) in eval mode applied on the channel (last) axis of
    (B,T,2H) (the reference permutes to (B,2H,T) and back,
    onssen/nn/deep_clustering.py:36-38)."""
    inv = 1.0 / np.sqrt(sd[prefix + "running_var"] + x.dtype.type(eps))
    return (x - sd[prefix + "running_mean"]) * inv * sd[prefix + "weight"] + sd[prefix + "bias"]


def l2_normalize(x, eps=1e-12):
    """F.normalize(p=2, dim=-1): x / max(||x||_2, eps)
    (onssen/nn/deep_clustering.py:41)."""
    n = np.sqrt(np.sum(x * x, axis=-1, keepdims=True))
    return x / np.maximum(n, x.dtype.type(eps))


def num_layers_of(sd, prefix="rnn."):
    k = 0
    while f"{prefix}weight_ih_l{k}" in sd:
        k += 1
    return k


def deep_clustering_forward(sd, x, dtype=np.float32, collect=None):
    """onssen/nn/deep_clustering.py:29-43 (eval mode).  x (B,T,F) ->
    embedding (B,T,F,D)."""
    sd = {k: np.asarray(v, dtype=dtype) for k, v in sd.items() if np.asarray(v).dtype.kind == "f"}
    x = np.asarray(x, dtype=dtype)
    B, T, F = x.shape
    r = blstm_stack(x, sd, "rnn.", num_layers_of(sd), collect)
    r = batchnorm_eval(r, sd, "bn.")
    if collect is not None:
        collect.append(r)
    e = r @ sd["fc_dc.weight"].T + sd["fc_dc.bias"]
    e = l2_normalize(e.reshape(B, T * F, -1))
    return e.reshape(B, T, F, -1)


def deep_clustering_forward_rounded(sd, x, rnd=bf16_round, exact_tail0=False, eps=1e-5):
    """The SAME network (onssen/nn/deep_clustering.py:29-43, eval mode) in the arithmetic of the product's opt-in
    ``ONSSEN_PRECISION=bf16`` mode (BASELINE cfg2's literal dtype): the operands of every matrix product -- layer inputs,
    h_{t-1}, W_ih, W_hh, the head's input and weights -- are rounded with ``rnd``; accumulation, biases, gates, cell
    state and the normalisation stay fp32.  Eval-mode BatchNorm is folded into fc_dc in fp32 BEFORE the rounding
    (W' = W diag(s), b' = b + W (beta - mu s), s = gamma / sqrt(var + eps)), as the product packs it.  Not a reference
    function: it restates that mode so that its tests can pin it (rnd = identity gives deep_clustering_forward)."""
    sd = {k: np.asarray(v, dtype=np.float32) for k, v in sd.items() if np.asarray(v).dtype.kind == "f"}
    x = np.asarray(x, dtype=np.float32)
    B, T, F = x.shape
    r = blstm_stack(x, sd, "rnn.", num_layers_of(sd), rnd=rnd, exact_tail0=exact_tail0)
    s = sd["bn.weight"] / np.sqrt(sd["bn.running_var"] + np.float32(eps))
    w = sd["fc_dc.weight"] * s
    b = sd["fc_dc.bias"] + sd["fc_dc.weight"] @ (sd["bn.bias"] - sd["bn.running_mean"] * s)
    e = rnd(r) @ rnd(w).T + b
    e = l2_normalize(e.reshape(B, T * F, -1))
    return e.reshape(B, T, F, -1)


def chimera_forward(sd, x, dtype=np.float32, prefix=""):
    """onssen/nn/chimera.py:30-46.  -> [embedding (B,T,F,D), mask_A, mask_B
    (B,T,F)]; the fc_mi output index is f*C + c."""
    sd = {k: np.asarray(v, dtype=dtype) for k, v in sd.items() if np.asarray(v).dtype.kind == "f"}
    x = np.asarray(x, dtype=dtype)
    B, T, F = x.shape
    r = blstm_stack(x, sd, prefix + "rnn.", num_layers_of(sd, prefix + "rnn."))
    e = r @ sd[prefix + "fc_dc.weight"].T + sd[prefix + "fc_dc.bias"]
    e = l2_normalize(e.reshape(B, T * F, -1)).reshape(B, T, F, -1)
    m = _sigmoid(r @ sd[prefix + "fc_mi.weight"].T + sd[prefix + "fc_mi.bias"])
    m = m.reshape(B, T, F, -1)
    return [e, m[..., 0], m[..., 1]]


def phase_net_forward(sd, x_mag, x_phase, dtype=np.float32):
    """onssen/nn/phase_network.py:34-67 with the only shape-consistent value
    of its undefined free variable (output_dim = input_dim)."""
    sdf = {k: np.asarray(v, dtype=dtype) for k, v in sd.items() if np.asarray(v).dtype.kind == "f"}
    x_mag = np.asarray(x_mag, dtype=dtype)
    x_phase = np.asarray(x_phase, dtype=dtype)
    B, T, F = x_mag.shape
    e, mA, mB = chimera_forward(sdf, x_mag, dtype, prefix="chimera.")
    outs = []
    L = num_layers_of(sdf, "rnn.")
    for m in (mA, mB):
        inp = np.concatenate([x_mag * m, x_phase.reshape(B, T, -1)], axis=2)
        r = blstm_stack(inp, sdf, "rnn.", L)
        r = batchnorm_eval(r, sdf, "bn.")
        p = r @ sdf["fc_phase.weight"].T + sdf["fc_phase.bias"]
        p = p.reshape(B, T, F, -1) + x_phase
        outs.append(l2_normalize(p))
    return [e, mA, mB, outs[0], outs[1]]


# ----------------------------------------------------------------------------- N1: deep-clustering loss value
def loss_dc_per_utt(embedding, one_hot, mag_mix):
    """Per-utterance deep-clustering loss term of onssen/loss/loss_dc.py:24-42 (fp64 here):
    silent bins drop out of V (:29-30), both factors are weighted by sqrt(|x_i| / sum_j |x_j|) (:34-36), and the
    three affinity terms are Frobenius NORMS, not squared norms (loss_util.py:7-11, loss_dc.py:39-42).
    embedding (B, TF, D), one_hot (B, TF, C), mag_mix (B, TF) -> (B,)"""
    V = np.asarray(embedding, np.float64)
    Y = np.asarray(one_hot, np.float64)
    mag = np.asarray(mag_mix, np.float64)
    V = Y.sum(2, keepdims=True) * V
    w = np.sqrt(mag / mag.sum(1, keepdims=True))[..., None]
    V, Y = V * w, Y * w
    fro = lambda a: np.sqrt((a * a).reshape(a.shape[0], -1).sum(1))
    Vt = V.transpose(0, 2, 1)
    return fro(Vt @ V) - 2.0 * fro(Vt @ Y) + fro(Y.transpose(0, 2, 1) @ Y)


def loss_dc(embedding, one_hot, mag_mix):
    """The (B, B) tensor upstream returns (loss_dc.py:44: (B,) * (B,1) broadcasts) -- its mean is the training loss
    (onssen/utils/train.py:78-79).  embedding (B,T,F,D), one_hot (B,T,F,C), mag_mix (B,T,F)."""
    B = embedding.shape[0]
    per = loss_dc_per_utt(embedding.reshape(B, -1, embedding.shape[-1]), one_hot.reshape(B, -1, one_hot.shape[-1]),
                          mag_mix.reshape(B, -1))
    return per[None, :] * np.asarray(mag_mix, np.float64).reshape(B, -1).sum(1)[:, None]


# ----------------------------------------------------------------------------- N4: enhancement network
def enhance_forward(sd, x, mag_noisy, dtype=np.float32):
    """onssen/nn/enhancement.py:38-52 (eval mode): mask = sigmoid(fc_mi(bn(blstm(x)))) (:44-48), restoration layer
    relu(fc_pre(mag_noisy)) (:49), product (:50), relu(fc_post(.)) (:51).  x, mag_noisy (B,T,F) -> clean (B,T,F)."""
    sd = {k: np.asarray(v, dtype=dtype) for k, v in sd.items() if np.asarray(v).dtype.kind == "f"}
    x, mag = np.asarray(x, dtype=dtype), np.asarray(mag_noisy, dtype=dtype)
    r = blstm_stack(x, sd, "rnn.", num_layers_of(sd))
    r = batchnorm_eval(r, sd, "bn.")
    mask = 1.0 / (1.0 + np.exp(-(r @ sd["fc_mi.weight"].T + sd["fc_mi.bias"])))
    pre = np.maximum(mag @ sd["fc_pre.weight"].T + sd["fc_pre.bias"], 0)
    return np.maximum((pre * mask) @ sd["fc_post.weight"].T + sd["fc_post.bias"], 0)


# ----------------------------------------------------------------------------- N4: SI-SDR with best permutation
def batch_sdr(estimation, origin, mask=None):
    """onssen/evaluate/sdr.py:11-87 in fp64: zero-mean both (:62-63), optional mask AFTER centring (:19-21), scale-
    invariant SDR of every (estimate, source) pair (:23-33), best permutation in sorted(permutations) order (:72-82).
    (B, C, n) -> sdr (B,), perm index (B,)"""
    from itertools import permutations
    e = np.asarray(estimation, np.float64)
    o = np.asarray(origin, np.float64)
    B, C, n = e.shape
    e = e - e.mean(2, keepdims=True)
    o = o - o.mean(2, keepdims=True)
    if mask is not None:
        m = np.asarray(mask, np.float64)[:, None, :]
        e, o = e * m, o * m
    tab = np.zeros((B, C, C))
    for i in range(C):
        for j in range(C):
            op = (o[:, j] ** 2).sum(1, keepdims=True) + 1e-8
            scale = (o[:, j] * e[:, i]).sum(1, keepdims=True) / op
            true = scale * o[:, j]
            res = e[:, i] - true
            tab[:, i, j] = 10 * np.log10((true ** 2).sum(1) + 1e-8) - 10 * np.log10((res ** 2).sum(1) + 1e-8)
    perms = sorted(set(permutations(range(C))))
    tot = np.stack([sum(tab[:, k, p[k]] for k in range(C)) for p in perms], 1)
    return tot.max(1) / C, tot.argmax(1)
