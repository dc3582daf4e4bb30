"""End-to-end separation: waveform -> K1..K10 -> waveforms.

Counterpart of tester.eval + get_est_sig (onssen/utils/test.py:29-41,
egs/wsj0-2mix/chimera/evaluate.py:23-45, deep_clustering/evaluate.py:22-47)
with the device->host->device hop removed for the mask-inference models."""
import numpy as np
import torch

from . import options
from .features import mask_istft, stft_logmag
from .nn._core import _XcdStatus, recovering


def _ragged(wav, lengths, hop_size):
    """(lengths, frames) as int32 device tensors for a ragged batch of waveforms, or (None, None)."""
    if lengths is None:
        return None, None
    from .features import _lengths_i32
    lengths = _lengths_i32(lengths, wav.shape[0], wav.shape[-1], wav.device, "lengths", lo=hop_size)
    return lengths, (1 + lengths // hop_size).to(torch.int32)


@recovering
@torch.no_grad()
def separate_chimera(model, wav, window_size=256, hop_size=64, lengths=None):
    """wav (B, n) cuda float32 -> (B, 2, n): masks straight from the network.  ``lengths`` (B,): a ragged batch of whole
    utterances padded to n samples (see separate_dc)."""
    lengths, frames = _ragged(wav, lengths, hop_size)
    logmag, ri = stft_logmag(wav, window_size, hop_size, lengths=lengths)
    _, masks = model.embedding_and_masks(logmag, frames)
    out = mask_istft(ri, masks, hop_size, wav.shape[-1], frames=frames, lengths=lengths)
    _XcdStatus.flush()            # an aborted recurrence is caught HERE (and the call re-run, see `recovering`), not by the next call
    return out


_CLUSTER_WS = {}           # (device, B, T, F, D[, "compact"], stream) -> buffer of a uniform shape
_CLUSTER_SCRATCH = {}      # (device, stream) -> grow-only buffer of the ragged / shape-changing calls (see dc_masks)
_CLUSTER_PINNED = set()    # keys of _CLUSTER_WS handed out for / during a hipGraph capture: never evicted


def _cluster_ws(device, key, nb, head, ragged):
    """Workspace of the clustering back end, private to the calling STREAM (round 5: every call rewrites the header --
    centroids, counters, the status word -- so two streams of one process separating at once must not share a buffer; work
    on one stream is ordered and reuses its own).  Uniform shapes: one buffer per (shape, stream) (hipGraph-capturable),
    most recently used last, at most 8 that no graph points into.  Ragged batches bring a new longest utterance every time:
    ONE grow-only buffer per (device, stream).  Either way it is allocated uninitialised and only its first ``head`` bytes
    (the library's ``comp_offset``: everything in front of the compacted array, which is as large as the embedding and is
    written before it is read) are zeroed."""
    capturing = torch.cuda.is_current_stream_capturing()
    stream = torch.cuda.current_stream(device).cuda_stream
    if ragged:
        if capturing:
            raise RuntimeError("dc_masks: ragged batches (frames=...) cannot be captured in a hipGraph (shared grow-only workspace)")
        ws = _CLUSTER_SCRATCH.get((device, stream))
        if ws is None or ws.numel() < nb:
            ws = _CLUSTER_SCRATCH[(device, stream)] = torch.empty(max(nb, int(nb * 1.25)), dtype=torch.uint8, device=device)
        ws[:head].zero_()                  # (the status word of an earlier call was examined by _XcdStatus before this one is issued)
        return ws
    if capturing:
        # a capture replays on whatever stream the graph is launched on: the buffer of the eager warm-up call of this shape
        # (any stream) is the one to capture, and from then on it belongs to the graph
        hit = next((k for k in _CLUSTER_WS if k[:-1] == key), None)
        if hit is None:
            raise RuntimeError("dc_masks: call it once eagerly for this shape before capturing it in a hipGraph (workspace allocation)")
        _CLUSTER_PINNED.add(hit)
        return _CLUSTER_WS[hit]
    key = key + (stream,)
    ws = _CLUSTER_WS.pop(key, None)
    if ws is None:
        loose = [k for k in _CLUSTER_WS if k not in _CLUSTER_PINNED]
        while len(loose) >= 8:
            _CLUSTER_WS.pop(loose.pop(0))
        ws = torch.empty(nb, dtype=torch.uint8, device=device)
        ws[:head].zero_()                  # status word starts out zero
    _CLUSTER_WS[key] = ws
    return ws


def dc_masks_from_features(model, logmag, db_threshold=40.0, iters=20, frames=None, tol=1e-4):
    """Deep-clustering masks (B,T,F,2) straight from the mixture's log-magnitude WITHOUT materialising the embedding
    (round 4): the active bins are known before the network runs (evaluate.py:36-37), so the threshold is turned into a
    target map first (onssen_dc_index_f32), ``model``'s fc_dc GEMM stores only the active bins' normalised rows -- straight
    into the compacted array the clustering reads (onssen_linear_x3p_compact) -- and threshold / initialisation / Lloyd /
    masks run on that (onssen_dc_cluster_compact_f32).  Bit-identical masks to ``dc_masks(model([logmag])[0], logmag)``,
    minus the 10 320 B/frame embedding write, its re-read and the compaction pass.

    Returns None when this forward cannot take that route (not an eval-mode ``deep_clustering`` on the persistent split-bf16
    path, an embedding width the GEMM's register epilogue does not hold, a forced launch-per-step re-run): the caller then
    computes the embedding and calls ``dc_masks``."""
    import os
    from .hip import get_lib
    from .nn._core import (_XcdPolicy, _XcdSerial, _XcdStatus, _stream, as_frames, heads_take_image, precision, run_blstm, use_hip_path)
    from .nn.deep_clustering import deep_clustering
    B, T, F = logmag.shape
    D = getattr(model, "embedding_dim", 0)
    if (not isinstance(model, deep_clustering) or not use_hip_path(model) or F != model.input_dim or precision() == "f32"
            or (frames is not None and precision() == "bf16")
            or options.get("dc_cluster") != "1" or options.get("dc_compact") != "1"
            or _XcdPolicy.force_steps != 0 or not heads_take_image(B, model.hidden_dim, (D,)) or D > 32):
        return None
    lib = get_lib()
    logmag = logmag.float().contiguous()
    if frames is not None:
        frames = as_frames(frames, B, T, logmag.device)
    nb, comp_off, dest_off = lib.dc_compact_layout(B, T, F, D)
    ws = _cluster_ws(logmag.device, (logmag.device, B, T, F, D, "compact"), nb, comp_off, frames is not None)
    st = torch.cuda.current_stream().cuda_stream
    fr = frames.data_ptr() if frames is not None else None
    # (round 5 measured "no": the map depends on the features only, so its three small launches were put on a side stream under
    #  the first layer's recurrence -- 1.888 / 1.879 ms against 1.875 / 1.882 ms per step in the captured graph, and the map's
    #  kernels, squeezed onto the 16 CUs the recurrence leaves, took 135 us instead of 10: profiles/NOTES.md)
    lib.dc_index(logmag.data_ptr(), B, T, F, D, float(db_threshold), ws.data_ptr(), nb, st, frames=fr)
    y = run_blstm(model._packed, model._ws, logmag, need_y=False, frames=frames)
    img = getattr(y, "x3_image", None)
    if img is None:                            # (the recurrence fell back between the check above and its own plan)
        return None
    wsb, off = img
    Hp = y.shape[3]
    hd = model._head.get(Hp)
    lib.linear_x3p_compact(wsb.data_ptr() + off, T * B, 2 * Hp, hd.img.data_ptr(), hd.b.data_ptr(), hd.N, D, 1e-12,
                           ws.data_ptr() + dest_off, T * F, F, ws.data_ptr() + comp_off, B, T * F * D, precision() == "bf16", st)
    masks = torch.empty(B, T, F, 2, device=logmag.device, dtype=torch.float32)
    _XcdSerial.before(logmag.device)            # the persistent Lloyd launch wants its workgroups resident together too
    lib.dc_cluster_compact(B, T, F, D, iters, masks.data_ptr(), ws.data_ptr(), nb, st, tol=float(tol))
    _XcdSerial.after(logmag.device)
    _XcdStatus.post_cluster(ws, int(lib.dll.onssen_dc_cluster_status_offset(B, D)))
    return masks


def dc_masks(emb, logmag, db_threshold=40.0, iters=20, frames=None, tol=1e-4):
    """Binary deep-clustering masks (B,T,F,2) on the device: threshold at max - db/20, 2-means on the active
    bins' embeddings (counterpart of evaluate.py:36-41, where it is sklearn on the host).

    Default: the active bins are compacted once and all Lloyd iterations run in ONE persistent launch (8 workgroups per
    utterance meeting at a counter); its waits are bounded, and a wait that gave up is reported like an aborted recurrence
    (``_XcdStatus``: the owning call is re-run with the launch-per-iteration form, which is also what runs inside
    ``_XcdPolicy.forced_steps()`` and with ONSSEN_DC_PERSISTENT=0).  ``iters`` / ``tol``: at most that many Lloyd iterations,
    stopped earlier at the exact fixed point or by sklearn's rule (``KMeans(tol=1e-4)``, upstream's default: summed squared
    centroid shift <= tol x mean per-feature variance); ``tol=0`` iterates to the fixed point.

    ``frames`` (B,): a ragged batch -- utterance b owns its first frames[b] frames; its padding is never active, takes no
    part in the threshold or the sums, and gets zero masks.  Such calls (a new longest utterance per batch) share ONE
    grow-only workspace per device, allocated uninitialised with only its header zeroed."""
    import os
    from . import _abi
    from .hip import get_lib
    from .nn._core import _XcdPolicy, _XcdSerial, _XcdStatus
    lib = get_lib()
    B, T, F, D = emb.shape
    emb, logmag = emb.contiguous(), logmag.contiguous()
    nb = int(lib.dll.onssen_dc_cluster_workspace_bytes(B, T, F, D))
    head = lib.dc_compact_layout(B, T, F, D)[1]        # the library's own offset of the compacted array = the header's size
    ws = _cluster_ws(emb.device, (emb.device, B, T, F, D), nb, head, frames is not None)
    persistent = options.get("dc_cluster") == "1" and _XcdPolicy.force_steps == 0
    masks = torch.empty(B, T, F, 2, device=emb.device, dtype=torch.float32)
    if frames is not None:
        from .features import _lengths_i32
        frames = _lengths_i32(frames, B, T, emb.device, "frames")
    if persistent:
        _XcdSerial.before(emb.device)
    lib.dc_cluster(emb.data_ptr(), logmag.data_ptr(), B, T, F, D, float(db_threshold), iters, masks.data_ptr(),
                   ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream,
                   flags=0 if persistent else _abi.DC_CLUSTER_LAUNCH_PER_ITERATION,
                   frames=frames.data_ptr() if frames is not None else None, tol=float(tol))
    if persistent:
        _XcdSerial.after(emb.device)
        _XcdStatus.post_cluster(ws, int(lib.dll.onssen_dc_cluster_status_offset(B, D)))
    return masks


@recovering
@torch.no_grad()
def separate_dc(model, wav, window_size=256, hop_size=64, db_threshold=40.0, host_kmeans=False, lengths=None):
    """Deep-clustering separation, waveform in -> (B, 2, n) waveforms out, entirely on the GPU
    (STFT -> network -> threshold + 2-means -> binary masks -> mask-apply + iSTFT).  ``host_kmeans=True``
    clusters with sklearn KMeans(n_clusters=2, random_state=0) on the host exactly as upstream does
    (egs/wsj0-2mix/deep_clustering/evaluate.py:36-38); the two differ only in the arbitrary cluster
    numbering and in bins that sit between the clusters.

    ``lengths`` (B,): a RAGGED batch of whole utterances -- row b holds lengths[b] valid samples of the n it is padded to
    (the reference separates them one at a time, onssen/utils/test.py:29-41; together they fill the chip).  Every row's
    result inside its own length is bit for bit what the batch-1 call on wav[b:b+1, :lengths[b]] returns; zeros after it."""
    lengths, frames = _ragged(wav, lengths, hop_size)
    logmag, ri = stft_logmag(wav, window_size, hop_size, lengths=lengths)
    if not host_kmeans:
        masks = dc_masks_from_features(model, logmag, db_threshold, frames=frames)      # the embedding never leaves the GEMM ...
        if masks is None:                                                                 # ... unless this forward cannot do that
            emb, = model([logmag]) if frames is None else model([logmag], frames=frames)
            masks = dc_masks(emb, logmag, db_threshold, frames=frames)
        out = mask_istft(ri, masks, hop_size, wav.shape[-1], frames=frames, lengths=lengths)
        _XcdStatus.flush()        # an aborted recurrence is reported by THIS call, not by the next one
        return out
    if frames is not None:
        raise ValueError("separate_dc: host_kmeans=True takes uniform batches only")
    emb, = model([logmag])
    _XcdStatus.flush()
    from sklearn.cluster import KMeans
    B, T, F, D = emb.shape
    masks = torch.zeros(B, T, F, 2, device=wav.device, dtype=torch.float32)
    for b in range(B):   # upstream evaluates with batch 1 (evaluate.py:34-35)
        feat = logmag[b]
        act = feat >= (feat.max() - db_threshold / 20.0)
        label = KMeans(n_clusters=2, random_state=0, n_init=10).fit_predict(emb[b][act].cpu().numpy())
        lab = torch.from_numpy(label.astype(np.int64)).to(wav.device).float()
        masks[b][act] = torch.stack([lab, 1.0 - lab], -1)
    return mask_istft(ri, masks, hop_size, wav.shape[-1])


class DCPipeline:
    """Deep-clustering separation of a STREAM of equally shaped batches, software-pipelined over consecutive batches (round 6;
    the evaluation loop of onssen/utils/test.py:29-41 / egs/wsj0-2mix/deep_clustering/evaluate.py:31-45 hands over one batch after
    the other).  A two-layer BLSTM of <= 32 rows fills the chip's 8 XCDs only with 8-row recurrence groups, whose time step costs
    nearly what a 16-row group's does -- so ``push(batch n)`` runs, in ONE persistent launch, layer 1 of batch n-1 on half of the
    XCDs and layer 0 of batch n on the other half (``onssen_blstm_pipe2_forward_f32``), with the rest of both batches' work around it:

        STFT, target map, input projection of batch n -> [ layer 1 (n-1) || layer 0 (n) ] -> layer 1's projection of batch n;
        fc_dc (active bins only) + 2-means + masks + iSTFT of batch n-1

    and returns the separated batch n-1 -- (B, 2, n_samples), valid until the next-but-one ``push`` -- or None for the first
    batch; ``flush()`` drains the last one.  Every batch gets exactly the arithmetic ``separate_dc`` gives it on 16-row recurrence
    groups without the fused first layer (bit for bit what the same rows get inside a 64-row ``separate_dc`` call; last-bit
    differences against the default 32-row call, inside the same tolerance); the price is one batch of latency.  Both parities of
    the step are captured as hipGraphs (``graph=True``).

    Needs an eval-mode ``deep_clustering`` with num_layers = 2, hidden <= 640, B <= 32 in the default split-bf16 arithmetic on the
    persistent recurrence; anything else raises (use ``separate_dc``).  A launch that gave up a bounded wait is reported by the
    next ``push`` / ``flush`` (XcdAborted; ``separate_dc_stream`` re-runs the affected batches with ``separate_dc``)."""

    def __init__(self, model, B, n_samples, window_size=256, hop_size=64, db_threshold=40.0, iters=20, tol=1e-4, graph=True):
        from . import _abi
        from .hip import get_lib
        from .nn._core import _XcdPolicy, _version_key, heads_take_image, precision
        from .nn.deep_clustering import deep_clustering
        dev = next(model.parameters()).device
        D = getattr(model, "embedding_dim", 0)
        F = window_size // 2 + 1
        why = None
        if not isinstance(model, deep_clustering) or model.num_layers != 2:
            why = "a deep_clustering model with num_layers = 2"
        elif model.training or dev.type != "cuda":
            why = "an eval-mode model on a ROCm device"
        elif F != model.input_dim:
            why = f"window_size // 2 + 1 == input_dim ({model.input_dim})"
        elif not 1 <= B <= 32 or model.hidden_dim > 640:
            why = "1 <= B <= 32 and hidden_dim <= 640"
        elif precision() != "bf16x3" or options.get("recurrence") != "1" or not _XcdPolicy.persistent_allowed():
            why = "the default split-bf16 arithmetic on the persistent recurrence"
        elif options.get("dc_cluster") != "1" or options.get("dc_compact") != "1" or D > 32 or not heads_take_image(B, model.hidden_dim, (D,)):
            why = "the compacted device-side clustering (embedding_dim in 4, 8, 16, 20; dc_cluster / dc_compact on)"
        if why:
            raise RuntimeError(f"DCPipeline needs {why}; use separate_dc")
        self.model, self.lib, self.dev = model, get_lib(), dev
        self.B, self.n, self.nfft, self.hop = int(B), int(n_samples), int(window_size), int(hop_size)
        self.T, self.F, self.D = 1 + self.n // self.hop, F, D
        self.db, self.iters, self.tol = float(db_threshold), int(iters), float(tol)
        self.H, self.ug = model.hidden_dim, 4 * -(-model.hidden_dim // 128)
        self.flags = _abi.BLSTM_BF16X3 | _abi.BLSTM_XCD
        lib, T = self.lib, self.T
        mk = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        self.wav = [torch.zeros(B, self.n, device=dev) for _ in range(2)]
        self.logmag = [mk(B, T, F) for _ in range(2)]
        self.ri = [mk(B, T, F, 2) for _ in range(2)]
        self.out = [torch.zeros(B, 2, self.n, device=dev) for _ in range(2)]
        self.masks = mk(B, T, F, 2)
        self.cnb, self.comp_off, self.dest_off = lib.dc_compact_layout(B, T, F, D)
        self.cws = []
        for _ in range(2):
            w = torch.empty(self.cnb, dtype=torch.uint8, device=dev)
            w[:self.comp_off].zero_()
            self.cws.append(w)
        self.cstat = int(lib.dll.onssen_dc_cluster_status_offset(B, D))
        self.wnb = lib.blstm_pipe2_workspace_bytes(B, T, F, self.H, self.ug)
        self.ws = torch.zeros(self.wnb, dtype=torch.uint8, device=dev)          # zeroed ONCE (ABI)
        self.img_off, _ = lib.blstm_pipe2_y_image(B, T, F, self.H, self.ug)
        self.count = 0               # batches pushed since the last reset
        self.graphs = [None, None]
        self.use_graph = bool(graph)
        self._wkey = None
        self._version_key = _version_key
        self._prime()

    # -- one pipeline step on the current stream: buffers of parity p take batch n, those of 1 - p hold batch n - 1
    def _enqueue(self, p, back_end=True):
        lib, B, T, F, D = self.lib, self.B, self.T, self.F, self.D
        st = torch.cuda.current_stream().cuda_stream
        pk = self.model._packed.get(self.ug)
        hd = self.model._head.get(pk.Hp)
        q = 1 - p
        lib.stft_logmag(self.wav[p].data_ptr(), B, self.n, self.n, self.nfft, self.hop, 1e-7, self.logmag[p].data_ptr(),
                        self.ri[p].data_ptr(), st)
        lib.dc_index(self.logmag[p].data_ptr(), B, T, F, D, self.db, self.cws[p].data_ptr(), self.cnb, st)
        lib.blstm_pipe2_forward(self.logmag[p].data_ptr(), T * F, F, B, T, F, self.H, self.ug,
                                [t.data_ptr() for t in pk.wih_img], [t.data_ptr() for t in pk.whh_x3], [t.data_ptr() for t in pk.bias],
                                self.ws.data_ptr(), self.wnb, self.flags, st)
        if not back_end:
            return
        cw = self.cws[q]
        lib.linear_x3p_compact(self.ws.data_ptr() + self.img_off, T * B, 2 * pk.Hp, hd.img.data_ptr(), hd.b.data_ptr(), hd.N, D, 1e-12,
                               cw.data_ptr() + self.dest_off, T * F, F, cw.data_ptr() + self.comp_off, B, T * F * D, False, st)
        lib.dc_cluster_compact(B, T, F, D, self.iters, self.masks.data_ptr(), cw.data_ptr(), self.cnb, st, tol=self.tol)
        m = self.masks
        lib.mask_istft(self.ri[q].data_ptr(), m.data_ptr(), m.stride(0), m.stride(3), m.stride(1), m.stride(2), B, 2, T, self.nfft,
                       self.hop, self.n, self.out[p].data_ptr(), st)

    def _prime(self):
        """Two eager steps on silence: every kernel has run once, both target maps and the layer-1 projection hold finite data."""
        from .nn._core import _XcdStatus
        _XcdStatus.poll()
        for p in (0, 1):
            self._enqueue(p, back_end=p == 1)
        self._post()
        self.count = 0

    def _post(self):
        from .nn._core import _XcdStatus
        _XcdStatus.post(self.ws)
        for w in self.cws:
            _XcdStatus.post_cluster(w, self.cstat)

    def _capture(self):
        key = self._version_key(self.model.rnn.flat_weights() + [self.model.fc_dc.weight, self.model.bn.running_mean])
        if self._wkey == key and self.graphs[0] is not None:
            return
        pk = self.model._packed.get(self.ug)         # (re)pack eagerly: inside the capture these only mark the images as captured
        self.model._head.get(pk.Hp)
        torch.cuda.synchronize(self.dev)
        self.graphs = [None, None]
        s = torch.cuda.Stream(self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        for p in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                self._enqueue(p)
            self.graphs[p] = g
        torch.cuda.current_stream(self.dev).wait_stream(s)
        self._wkey = key

    def step(self, p):
        if self.use_graph:
            self.graphs[p].replay()
        else:
            self._enqueue(p)

    def replay(self):
        """One steady-state step on the input buffers as they are (``self.wav[parity]``: bench.py fills them once and times this):
        no copy, no status traffic.  At least one batch must be in flight (the parities alternate from there)."""
        if self.count == 0:
            raise RuntimeError("DCPipeline.replay: push a batch first")
        if self.use_graph:
            self._capture()
        self.step(self.count & 1)
        self.count += 1

    @torch.no_grad()
    def push(self, wav, check=True):
        """Hand over batch n (B, n_samples) float32 on the device; returns the separated batch n-1 (B, 2, n_samples) or None."""
        from .nn._core import _XcdStatus
        if tuple(wav.shape) != (self.B, self.n) or not wav.is_cuda:
            raise ValueError(f"DCPipeline.push: expected a ({self.B}, {self.n}) tensor on {self.dev}, got {tuple(wav.shape)} on {wav.device}")
        if check:
            _XcdStatus.poll()                      # reports of earlier steps that have landed (raises XcdAborted)
        if self.use_graph:
            self._capture()
        p = self.count & 1
        self.wav[p].copy_(wav, non_blocking=True)
        if self.count == 0:
            # the first batch of a stream has nothing behind it: no back end (the other parity's clustering workspace holds a map that
            # was consumed by the drain, or none) -- an eager step, the captured graphs are the steady state
            self._enqueue(p, back_end=False)
        else:
            self.step(p)
        if check:
            self._post()
        self.count += 1
        return self.out[p] if self.count > 1 else None

    @torch.no_grad()
    def flush(self):
        """Drain: the separated LAST batch (or None if nothing is in flight); the pipeline is empty afterwards."""
        from .nn._core import _XcdStatus
        if self.count == 0:
            return None
        p = self.count & 1
        self.wav[p].zero_()
        if self.use_graph:
            self._capture()
        self.step(p)
        self._post()
        self.count = 0
        _XcdStatus.flush()
        return self.out[p]

    def reset(self):
        """Forget the batch in flight (after an aborted step: the exchange header was zeroed by the status poll)."""
        self.count = 0


class DCRaggedPipeline:
    """``DCPipeline`` for a stream of RAGGED batches of whole utterances (round 6c) -- the shape the reference evaluates one by one
    (onssen/utils/test.py:29-41) and ``separate_dc(..., lengths=)`` runs K at a time: every batch is B <= 16 rows padded to ITS OWN
    longest utterance.  A ragged batch of <= 16 rows runs on stacked 4-row recurrence groups, whose time step costs what an 8-row
    group's does, one layer after the other; here ``push(batch n)`` runs layer 1 of batch n-1 and layer 0 of batch n in ONE persistent
    launch on stacked 8-row groups (``onssen_blstm_pipe2_forward_ragged_f32``: each half of the launch has its own number of time steps
    and its own row lengths), with the rest of both batches' work around it exactly as in ``DCPipeline``, and returns the separated batch
    n-1 -- (B, 2, n_{n-1}), valid until the next-but-one ``push`` -- or None for the first batch; ``flush()`` drains the last one.

    Every utterance's result inside its own length is bit for bit what ``separate_dc(model, wav, lengths=lengths)`` gives it (stacked
    tiles in both: a tile column never sees its neighbours), i.e. what the batch-1 call on that utterance returns; zeros after it.
    Eager launches (a new longest utterance per batch: nothing to capture); buffers are sized once for ``n_cap`` samples per row.
    Needs what ``DCPipeline`` needs, with B <= 16."""

    @staticmethod
    def why_not(model, B, window_size=256):
        """None if this model / batch size / mode can run here, else what is missing (no allocation, no launch)."""
        from .nn._core import _XcdPolicy, heads_take_image, precision
        from .nn.deep_clustering import deep_clustering
        if not isinstance(model, deep_clustering) or model.num_layers != 2:
            return "a deep_clustering model with num_layers = 2"
        dev = next(model.parameters()).device
        D = getattr(model, "embedding_dim", 0)
        if model.training or dev.type != "cuda":
            return "an eval-mode model on a ROCm device"
        if window_size // 2 + 1 != model.input_dim:
            return f"window_size // 2 + 1 == input_dim ({model.input_dim})"
        if not 1 <= B <= 16 or model.hidden_dim > 640:
            return "1 <= B <= 16 (ragged rows run on stacked tiles) and hidden_dim <= 640"
        if precision() != "bf16x3" or options.get("recurrence") != "1" or not _XcdPolicy.persistent_allowed():
            return "the default split-bf16 arithmetic on the persistent recurrence"
        if options.get("dc_cluster") != "1" or options.get("dc_compact") != "1" or D > 32 or not heads_take_image(B, model.hidden_dim, (D,)):
            return "the compacted device-side clustering (embedding_dim in 4, 8, 16, 20; dc_cluster / dc_compact on)"
        return None

    def __init__(self, model, B, n_cap, window_size=256, hop_size=64, db_threshold=40.0, iters=20, tol=1e-4):
        from . import _abi
        from .hip import get_lib
        dev = next(model.parameters()).device
        D = getattr(model, "embedding_dim", 0)
        F = window_size // 2 + 1
        why = self.why_not(model, B, window_size)
        if why is None and n_cap < hop_size:
            why = "n_cap >= hop_size"
        if why:
            raise RuntimeError(f"DCRaggedPipeline needs {why}; use separate_dc(..., lengths=)")
        self.model, self.lib, self.dev = model, get_lib(), dev
        self.B, self.n_cap, self.nfft, self.hop = int(B), int(n_cap), int(window_size), int(hop_size)
        self.T_cap, self.F, self.D = 1 + self.n_cap // self.hop, F, D
        self.db, self.iters, self.tol = float(db_threshold), int(iters), float(tol)
        self.H, self.ug = model.hidden_dim, 4 * -(-model.hidden_dim // 128)
        self.flags = _abi.BLSTM_BF16X3 | _abi.BLSTM_XCD
        lib, T = self.lib, self.T_cap
        mk = lambda k: torch.empty(k, device=dev, dtype=torch.float32)
        self.logmag = [torch.zeros(B * T * F, device=dev) for _ in range(2)]
        self.ri = [mk(B * T * F * 2) for _ in range(2)]
        self.out = [mk(B * 2 * self.n_cap) for _ in range(2)]
        self.masks = mk(B * T * F * 2)
        self.cnb, self.comp_off, _ = lib.dc_compact_layout(B, T, F, D)
        self.cws = []
        for _ in range(2):
            w = torch.empty(self.cnb, dtype=torch.uint8, device=dev)
            w[:self.comp_off].zero_()
            self.cws.append(w)
        self.cstat = int(lib.dll.onssen_dc_cluster_status_offset(B, D))
        self.wnb = lib.blstm_pipe2_workspace_bytes(B, T, F, self.H, self.ug)
        self.ws = torch.zeros(self.wnb, dtype=torch.uint8, device=dev)          # zeroed ONCE (ABI)
        self.img_off, _ = lib.blstm_pipe2_y_image(B, T, F, self.H, self.ug)
        self.meta = [None, None]     # per parity: (T, n, frames, lengths) of the batch its buffers hold
        self.count = 0

    def _step(self, p, cur, back_end):
        """The pair launch for the batch ``cur`` = (T, n, frames, lengths, logmag, ri) beside the batch of parity 1 - p, then
        (``back_end``) that batch's head GEMM, clustering, masks and iSTFT into ``out[p]``."""
        lib, B, F, D = self.lib, self.B, self.F, self.D
        st = torch.cuda.current_stream().cuda_stream
        pk = self.model._packed.get(self.ug)
        hd = self.model._head.get(pk.Hp)
        q = 1 - p
        T, n, frames, lengths, x, _ = cur
        Tq, nq, frames_q, lengths_q, _, ri_q = self.meta[q] if back_end else cur
        lib.blstm_pipe2_forward_ragged(x.data_ptr(), T * F, F, B, self.T_cap, T, frames.data_ptr(), Tq, frames_q.data_ptr(), F, self.H,
                                       self.ug, [t.data_ptr() for t in pk.wih_img], [t.data_ptr() for t in pk.whh_x3],
                                       [t.data_ptr() for t in pk.bias], self.ws.data_ptr(), self.wnb, self.flags, st)
        if not back_end:
            return None
        cw = self.cws[q]
        _, comp_off, dest_off = lib.dc_compact_layout(B, Tq, F, D)
        lib.linear_x3p_compact(self.ws.data_ptr() + self.img_off, Tq * B, 2 * pk.Hp, hd.img.data_ptr(), hd.b.data_ptr(), hd.N, D, 1e-12,
                               cw.data_ptr() + dest_off, Tq * F, F, cw.data_ptr() + comp_off, B, Tq * F * D, False, st)
        lib.dc_cluster_compact(B, Tq, F, D, self.iters, self.masks.data_ptr(), cw.data_ptr(), self.cnb, st, tol=self.tol)
        lib.mask_istft(ri_q.data_ptr(), self.masks.data_ptr(), Tq * F * 2, 1, F * 2, 2, B, 2, Tq, self.nfft, self.hop, nq,
                       self.out[p].data_ptr(), st, frames=frames_q.data_ptr(), lengths=lengths_q.data_ptr())
        return self.out[p][:B * 2 * nq].view(B, 2, nq)

    def _post(self, q, back_end):
        from .nn._core import _XcdStatus
        _XcdStatus.post(self.ws)
        if back_end:
            _XcdStatus.post_cluster(self.cws[q], self.cstat)

    def _advance(self, cur, check):
        """Target map of ``cur`` (the features are in place), the pipeline step, the status posts."""
        T, n, frames, lengths, x, _ = cur
        p = self.count & 1
        self.lib.dc_index(x.data_ptr(), self.B, T, self.F, self.D, self.db, self.cws[p].data_ptr(), self.cnb,
                          torch.cuda.current_stream().cuda_stream, frames=frames.data_ptr())
        back_end = self.count > 0
        out = self._step(p, cur, back_end)
        self.meta[p] = cur
        if check:
            self._post(1 - p, back_end)
        self.count += 1
        return out

    @torch.no_grad()
    def push(self, wav, lengths, check=True):
        """Hand over batch n: ``wav`` (B, n) float32 on the device, row b holding ``lengths[b]`` <= n valid samples (n <= n_cap);
        returns the separated batch n-1 (B, 2, n_{n-1}) or None."""
        from .features import _lengths_i32
        from .nn._core import _XcdStatus
        if wav.dim() != 2 or wav.shape[0] != self.B or not wav.is_cuda or not self.hop <= wav.shape[1] <= self.n_cap:
            raise ValueError(f"DCRaggedPipeline.push: expected a ({self.B}, n <= {self.n_cap}) tensor on {self.dev}, got "
                             f"{tuple(wav.shape)} on {wav.device}")
        wav = wav.float()
        if wav.stride(1) != 1:
            wav = wav.contiguous()
        B, n = wav.shape
        lengths = _lengths_i32(lengths, B, n, self.dev, "lengths", lo=max(self.hop, self.nfft // 2 + 1))
        frames = (1 + lengths // self.hop).to(torch.int32)
        if check:
            _XcdStatus.poll()                      # reports of earlier steps that have landed (raises XcdAborted)
        p = self.count & 1
        T = 1 + n // self.hop
        self.lib.stft_logmag(wav.data_ptr(), B, n, wav.stride(0), self.nfft, self.hop, 1e-7, self.logmag[p].data_ptr(),
                             self.ri[p].data_ptr(), torch.cuda.current_stream().cuda_stream, n_per_utt=lengths.data_ptr())
        return self._advance((T, n, frames, lengths, self.logmag[p], self.ri[p]), check)

    @torch.no_grad()
    def push_features(self, logmag, stft_ri, frames, lengths, n, check=True):
        """``push`` for a batch whose features come from elsewhere (the evaluation loader's items, onssen/data/wsj0_2mix.py:231-245,
        collated by ``evaluate.tester.collate``): ``logmag`` (B, T, F) and ``stft_ri`` (B, T, F, 2) float32 on the device, row b owning
        its first ``frames[b]`` <= T frames and ``lengths[b]`` <= n output samples.  The tensors are used where they are (no copy) and
        kept until their batch has left the pipeline."""
        from .features import _lengths_i32
        from .nn._core import _XcdStatus
        B, T, F = logmag.shape
        if (B != self.B or F != self.F or T > self.T_cap or n > self.n_cap or tuple(stft_ri.shape) != (B, T, F, 2) or not logmag.is_cuda
                or not stft_ri.is_cuda):
            raise ValueError(f"DCRaggedPipeline.push_features: expected ({self.B}, T <= {self.T_cap}, {self.F}) features and their "
                             f"(..., 2) spectrum on {self.dev}, n <= {self.n_cap}; got {tuple(logmag.shape)}, {tuple(stft_ri.shape)}, n = {n}")
        logmag, stft_ri = logmag.float().contiguous(), stft_ri.float().contiguous()
        frames = _lengths_i32(frames, B, T, self.dev, "frames")
        lengths = _lengths_i32(lengths, B, n, self.dev, "lengths")
        if check:
            _XcdStatus.poll()
        return self._advance((T, int(n), frames, lengths, logmag, stft_ri), check)

    @torch.no_grad()
    def flush(self):
        """Drain: the separated LAST batch (or None if nothing is in flight); the pipeline is empty afterwards."""
        from .nn._core import _XcdStatus
        if self.count == 0:
            return None
        p = self.count & 1
        q = 1 - p
        # the launch's other half needs SOME batch: the last one's own features again (its layer-0 output is not used)
        out = self._step(p, self.meta[q], True)
        self._post(q, True)
        self.count = 0
        _XcdStatus.flush()
        return out

    def reset(self):
        """Forget the batch in flight (after an aborted step: the exchange header was zeroed by the status poll)."""
        self.count = 0


@torch.no_grad()
def separate_dc_ragged_stream(model, batches, window_size=256, hop_size=64, db_threshold=40.0):
    """Generator: ``separate_dc(model, wav, lengths=lengths)`` over an iterable of ragged batches ``(wav (B, n), lengths (B,))`` --
    B <= 16 whole utterances each, every batch padded to its own longest one -- through ``DCRaggedPipeline``: yields one (B, 2, n)
    result per batch, in order (a fresh tensor each), bit for bit what ``separate_dc`` returns for it.  The pipeline's buffers are
    sized for the longest batch seen so far (a longer one drains it and starts a larger one); a batch it cannot take (another B,
    more than 16 rows, a model or mode ``DCRaggedPipeline`` refuses) goes through ``separate_dc``; a step whose persistent launch
    gave up a bounded wait is recovered by separating the batches it touched again with ``separate_dc``."""
    import warnings
    from .nn._core import XcdAborted, _XcdPolicy, _XcdStatus
    pipe, held = None, []                      # held: (wav, lengths) whose result has not been yielded yet (at most 2)
    sep = lambda w, l: separate_dc(model, w, window_size, hop_size, db_threshold, lengths=l)

    def drain():
        nonlocal held
        if pipe and held:
            try:
                res = pipe.flush().clone()
                held = []
                return [res]
            except XcdAborted as e:
                _XcdPolicy.recovered += 1
                warnings.warn(f"onssen_amd: {e}  Re-running the last batch with separate_dc.", RuntimeWarning)
                pipe.reset()
        res = [sep(w, l) for w, l in held[-1:]] if pipe else []
        held = []
        return res

    for wav, lengths in batches:
        B, n = wav.shape
        if pipe is not None and pipe is not False and (B != pipe.B or n > pipe.n_cap):
            yield from drain()
            pipe = None
        if pipe is None:
            try:
                pipe = DCRaggedPipeline(model, B, int(n * 1.25) if B <= 16 else n, window_size, hop_size, db_threshold)
            except RuntimeError:
                pipe = False
        if pipe is False:
            yield sep(wav, lengths)
            pipe = None
            continue
        held.append((wav, lengths))
        try:
            out = pipe.push(wav, lengths)
            if out is not None:
                res = out.clone()
                _XcdStatus.flush()             # the step that produced it has completed cleanly (the next one is not enqueued yet)
                held.pop(0)
                yield res
        except XcdAborted as e:
            _XcdPolicy.recovered += 1
            warnings.warn(f"onssen_amd: {e}  Re-running the batches of that pipeline step with separate_dc.", RuntimeWarning)
            pipe.reset()
            for w, l in held:
                yield sep(w, l)
            held = []
    yield from drain()


@torch.no_grad()
def separate_dc_stream(model, batches, window_size=256, hop_size=64, db_threshold=40.0, graph=True):
    """Generator: ``separate_dc`` over an iterable of equally shaped (B, n) device batches, through ``DCPipeline`` -- yields one
    (B, 2, n) result per batch, in order (a fresh tensor each).  A step whose persistent launch gave up a bounded wait is
    recovered here: the two batches it touched are separated again with ``separate_dc`` and the pipeline restarts.  Batches
    the pipeline cannot take (see DCPipeline) go through ``separate_dc`` one by one."""
    import warnings
    from .nn._core import XcdAborted, _XcdPolicy, _XcdStatus
    pipe, held = None, []                      # held: inputs whose result has not been yielded yet (at most 2)
    for wav in batches:
        if pipe is None:
            try:
                pipe = DCPipeline(model, wav.shape[0], wav.shape[1], window_size, hop_size, db_threshold, graph=graph)
            except RuntimeError:
                pipe = False
        if pipe is False or tuple(wav.shape) != (pipe.B, pipe.n):
            for h in held:                     # (a shape change: drain what is in flight first)
                yield separate_dc(model, h, window_size, hop_size, db_threshold)
            held = []
            if pipe:
                pipe.reset()
            yield separate_dc(model, wav, window_size, hop_size, db_threshold)
            continue
        held.append(wav)
        try:
            out = pipe.push(wav)
            if out is not None:
                res = out.clone()
                _XcdStatus.flush()             # the step that produced it has completed cleanly (the next one is not enqueued yet)
                held.pop(0)
                yield res
        except XcdAborted as e:
            _XcdPolicy.recovered += 1
            warnings.warn(f"onssen_amd: {e}  Re-running the batches of that pipeline step with separate_dc.", RuntimeWarning)
            pipe.reset()
            for h in held:
                yield separate_dc(model, h, window_size, hop_size, db_threshold)
            held = []
    if pipe and held:
        try:
            res = pipe.flush().clone()
            yield res
        except XcdAborted as e:
            _XcdPolicy.recovered += 1
            warnings.warn(f"onssen_amd: {e}  Re-running the last batch with separate_dc.", RuntimeWarning)
            pipe.reset()
            yield separate_dc(model, held[-1], window_size, hop_size, db_threshold)
