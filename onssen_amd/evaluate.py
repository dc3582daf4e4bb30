"""Evaluation metric of the separation recipes on the device: batch SI-SDR with the best source
permutation, same signature and results as onssen.evaluate.batch_SDR_torch (onssen/evaluate/sdr.py:40-87)."""
import torch

from .hip import get_lib

_WS = {}


def batch_SDR_torch(estimation, origin, mask=None, return_perm=False, lengths=None):
    """estimation, origin (batch, nsource, nsample); mask optional (batch, nsample).  Returns (batch,) mean SDR of the
    best permutation (and, with return_perm, the permutation's index in lexicographic order, like upstream).
    ``lengths`` (batch,) (extension): a ragged batch -- row b holds lengths[b] <= nsample samples, the rest is padding that
    takes no part (same value, bit for bit, as the batch-1 call on the row's own samples)."""
    assert estimation.size() == origin.size(), "Estimation and original sources should have same shape."
    B, C, n = estimation.size()
    assert C < n, "Axis 1 should be the number of sources, and axis 2 should be the signal."
    if not estimation.is_cuda:
        raise RuntimeError("onssen_amd.evaluate.batch_SDR_torch runs on a ROCm device; there is no CPU fallback")
    if C > 4:
        raise ValueError("onssen_batch_sdr_f32 scans the permutations of at most 4 sources")
    lib, dev = get_lib(), estimation.device
    est, org = estimation.float().contiguous(), origin.float().contiguous()
    mk = mask.float().contiguous() if mask is not None else None
    ws = _WS.get((dev, B))
    if ws is None:
        ws = _WS[(dev, B)] = torch.empty(lib.batch_sdr_workspace_bytes(B), dtype=torch.uint8, device=dev)
    sdr = torch.empty(B, device=dev, dtype=torch.float32)
    perm = torch.empty(B, device=dev, dtype=torch.int32)
    if lengths is not None:
        from .features import _lengths_i32
        lengths = _lengths_i32(lengths, B, n, dev, "lengths")
    lib.batch_sdr(est.data_ptr(), org.data_ptr(), mk.data_ptr() if mk is not None else None, B, C, n, sdr.data_ptr(),
                  perm.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream,
                  lengths=lengths.data_ptr() if lengths is not None else None)
    return (sdr, perm.long()) if return_perm else sdr


# ----------------------------------------------------------------------------- tester.get_est_sig-shaped adapters
class tester:
    """Body of onssen.utils.tester (onssen/utils/test.py:7-41) around this package's back end: ``args`` carries ``model``,
    ``test_loader`` and ``device`` like upstream; checkpoint loading stays with the caller (``model`` is used as given
    unless ``args`` has a ``checkpoint_path`` whose ``final.mdl`` exists).  ``eval()`` returns the mean SI-SDR (upstream
    only prints it).  The evaluation loader yields, per utterance (onssen/data/wsj0_2mix.py:231-245, batch 1 upstream, any
    batch here): ``[feature_mix (B,T,F)]``, ``[stft_r_mix (B,T,F), stft_i_mix (B,T,F), sig_ref (B,C,n)]``."""

    def __init__(self, args):
        import os
        g = (lambda k, d=None: args.get(k, d)) if isinstance(args, dict) else (lambda k, d=None: getattr(args, k, d))
        self.model_name, self.test_loader = g("model_name"), g("test_loader")
        self.device = torch.device(g("device", "cuda:0"))
        self.model = g("model")
        ck = g("checkpoint_path")
        if ck and os.path.exists(os.path.join(ck, "final.mdl")):
            self.model.load_state_dict(torch.load(os.path.join(ck, "final.mdl"))["model"])
        self.model = self.model.to(self.device)

    def get_est_sig(self, input, label, output):
        raise NotImplementedError

    @staticmethod
    def collate(items):
        """K evaluation items (``[feature (1,T_k,F), ...], [stft_r (1,T_k,F), stft_i (1,T_k,F), sig_ref (1,C,n_k)]``, the
        loader contract above) -> ONE ragged batch: every tensor zero-padded to the longest utterance, plus ``frames`` (K,)
        and ``lengths`` (K,) int32 device tensors.  Inputs and all labels but the last are padded along the frame axis, the
        last label (``sig_ref``) along the sample axis."""
        from torch.nn.utils.rnn import pad_sequence
        rows = [([t[r] for t in inp], [t[r] for t in lab]) for inp, lab in items for r in range(inp[0].shape[0])]
        dev = rows[0][0][0].device
        frames = [r[0][0].shape[0] for r in rows]
        lengths = [r[1][-1].shape[-1] for r in rows]
        inp = [pad_sequence([r[0][k] for r in rows], batch_first=True) for k in range(len(rows[0][0]))]
        lab = [pad_sequence([r[1][k] for r in rows], batch_first=True) for k in range(len(rows[0][1]) - 1)]
        lab.append(pad_sequence([r[1][-1].transpose(0, -1) for r in rows], batch_first=True).transpose(1, -1).contiguous())
        return inp, lab, (torch.tensor(frames, dtype=torch.int32).to(dev, non_blocking=True),
                          torch.tensor(lengths, dtype=torch.int32).to(dev, non_blocking=True))

    def _one(self, input, label, ragged):
        """SI-SDR of one forward (a loader item, or K of them collated: ``ragged`` = (frames, lengths))."""
        if ragged is None:
            output = self.model(input)
            sig_est, sig_ref = self.get_est_sig(input, label, output)
            return batch_SDR_torch(sig_est, sig_ref)
        frames, lengths = ragged
        output = self.model(input, frames=frames)
        sig_est, sig_ref = self.get_est_sig(input, label, output, frames=frames, lengths=lengths)
        return batch_SDR_torch(sig_est, sig_ref, lengths=lengths)

    def _forwards(self, batch, bucket):
        """(input, label, ragged) per forward: the loader's items as they come, or K of them collated."""
        if int(batch) <= 1:
            for input, label in self.test_loader:
                yield input, label, None
            return
        K, G = int(batch), max(1, int(bucket))

        def batches(items):      # similar lengths together: longest first
            items = sorted(items, key=lambda it: -it[0][0].shape[1]) if G > 1 else items
            chunk = []
            for it in items:
                chunk.append(it)
                if sum(i[0][0].shape[0] for i in chunk) >= K:
                    yield self.collate(chunk)
                    chunk = []
            if chunk:
                yield self.collate(chunk)
        items = []
        for item in self.test_loader:
            items.append(item)
            if sum(i[0][0].shape[0] for i in items) >= K * G:
                yield from batches(items)
                items = []
        if items:
            yield from batches(items)

    def eval(self, window=8, batch=1, bucket=1):
        """Mean SI-SDR over the loader.

        ``batch`` = 1 is the reference's loop (onssen/utils/test.py:29-41: one utterance per forward).  That shape keeps 2
        of the chip's 8 XCDs busy, so ``batch`` = K > 1 (16 fills the chip) evaluates K utterances of DIFFERENT lengths
        per forward: they are zero-padded to the longest (``collate``) and every stage -- network, clustering / masks,
        iSTFT, SI-SDR -- is told each row's own extent, so that every utterance's SDR is bit for bit the one the batch-1
        loop computes.  ``get_est_sig`` must then accept ``frames=`` and ``lengths=`` (tester_dc / tester_chimera do).
        ``bucket`` = G > 1: G * K utterances are read ahead, sorted by length and cut into G batches of similar lengths -- a
        batch costs what its longest utterance costs, and real corpora spread over a factor of 5 in length; the mean does not
        depend on the order.

        Upstream synchronises on every utterance (``.item()``); here ``window`` forwards are queued back to back -- the host
        launches forward i+1 while the device works on forward i -- and the status words of their persistent launches are
        examined once per window.  An aborted launch re-runs that window on the launch-per-step recurrence (inference
        is functional: same inputs, same outputs)."""
        import warnings
        from .nn._core import StalePackedWeights, XcdAborted, XcdNonFinite, _XcdPolicy, _XcdStatus
        total, count = 0.0, 0
        self.model = self.model.eval()

        one = self._one

        def rerun(pend, e):
            import contextlib
            _XcdPolicy.recovered += 1
            stale = isinstance(e, StalePackedWeights)        # the weight guard: the images are already dropped, the same kernels again
            if not isinstance(e, XcdNonFinite):
                warnings.warn(f"onssen_amd: {e}  Re-running {len(pend)} forward(s) "
                              + ("on freshly packed weights." if stale else "on the launch-per-step recurrence."), RuntimeWarning)
            try:
                _XcdStatus.flush(policy=False)   # the window's other launches ran into the same abort word: drain their reports
            except XcdAborted:
                pass
            with (contextlib.nullcontext() if stale else _XcdPolicy.forced_steps()):
                sdrs = [one(input, label, ragged) for input, label, ragged, _ in pend]
                _XcdStatus.flush()
            return sdrs

        def close(pend):
            sdrs = [sdr for _, _, _, sdr in pend]
            try:
                _XcdStatus.flush()
            except XcdAborted as e:
                sdrs = rerun(pend, e)
            return float(torch.cat([x.reshape(-1) for x in sdrs]).double().sum()), sum(x.numel() for x in sdrs)

        forwards = lambda: self._forwards(batch, bucket)

        with torch.no_grad():
            pend = []
            for input, label, ragged in forwards():
                try:
                    pend.append((input, label, ragged, one(input, label, ragged)))
                except XcdAborted as e:          # reported by this forward's look at the statuses of the launches before it
                    pend.append((input, label, ragged, None))
                    sdrs = rerun(pend, e)
                    total += float(torch.cat([x.reshape(-1) for x in sdrs]).double().sum())
                    count += sum(x.numel() for x in sdrs)
                    pend = []
                    continue
                if len(pend) >= max(1, int(window)):
                    t, c = close(pend)
                    total, count, pend = total + t, count + c, []
            if pend:
                t, c = close(pend)
                total, count = total + t, count + c
        return total / max(count, 1)


def _mix_ri(label):
    stft_r_mix, stft_i_mix, sig_ref = label
    return torch.stack([stft_r_mix.float(), stft_i_mix.float()], -1).contiguous(), sig_ref


class tester_dc(tester):
    """egs/wsj0-2mix/deep_clustering/evaluate.py:11-47: threshold at max - 40/20, 2-means on the active bins'
    embeddings, binary masks, mask-apply + iSTFT -- on the device (``host_kmeans=True``: upstream's sklearn path)."""

    def __init__(self, args, hop_size=64, host_kmeans=False):
        super().__init__(args)
        self.hop_size, self.host_kmeans = hop_size, host_kmeans

    def eval(self, window=8, batch=1, bucket=1, pipeline=True):
        """``tester.eval``; with ``batch`` = K in 2 .. 16 and a two-layer network the forwards are software-pipelined over consecutive
        batches (round 6c, ``separation.DCRaggedPipeline``: layer 1 of forward i-1 and layer 0 of forward i share one persistent
        launch; the active bins' embedding goes straight from the head GEMM into the clustering), same SI-SDR per utterance bit for
        bit -- 1.35 x the utterances per second of the plain ``batch`` = 16 loop.  ``pipeline=False``, ``host_kmeans``, other models,
        modes or batch sizes: the plain loop.  An aborted persistent launch re-runs the forwards it touched on the plain loop's
        recovery path."""
        import warnings
        from .nn._core import XcdAborted, _XcdPolicy, _XcdStatus
        from .separation import DCRaggedPipeline
        K = int(batch)
        self.model = self.model.eval()
        F = getattr(self.model, "input_dim", 0)
        if not pipeline or self.host_kmeans or not 2 <= K <= 16 or DCRaggedPipeline.why_not(self.model, K, 2 * (F - 1)) is not None:
            return super().eval(window, batch, bucket)
        hop = self.hop_size
        total, count = 0.0, 0
        pipe, held = None, []          # held: the forwards whose estimate has not come back yet (at most 2)

        def plain(items, forced):
            import contextlib
            t, c = 0.0, 0
            with (_XcdPolicy.forced_steps() if forced else contextlib.nullcontext()):
                for input, label, ragged in items:
                    sdr = self._one(input, label, ragged)
                    _XcdStatus.flush()
                    t, c = t + float(sdr.double().sum()), c + sdr.numel()
            return t, c

        def settle(est, item):
            (_, label, (frames, lengths)) = item
            sdr = batch_SDR_torch(est, label[-1].float(), lengths=lengths)
            return float(sdr.double().sum()), sdr.numel()

        def recover(e):
            nonlocal held, pipe
            _XcdPolicy.recovered += 1
            warnings.warn(f"onssen_amd: {e}  Re-running {len(held)} forward(s) on the launch-per-step recurrence.", RuntimeWarning)
            try:
                _XcdStatus.flush(policy=False)
            except XcdAborted:
                pass
            if pipe:
                pipe.reset()
            t, c = plain(held, True)
            held = []
            return t, c

        def drain():
            nonlocal held
            if not (pipe and held):
                return 0.0, 0
            try:
                est = pipe.flush()
                t, c = settle(est, held[0])
                held = []
                return t, c
            except XcdAborted as e:
                return recover(e)

        with torch.no_grad():
            for item in self._forwards(batch, bucket):
                input, label, ragged = item
                feature_mix, = input
                ri, sig_ref = _mix_ri(label)
                B, T, _ = feature_mix.shape
                n = sig_ref.shape[-1]
                if pipe is not None and (B != pipe.B or T > pipe.T_cap or n > pipe.n_cap):
                    t, c = drain()
                    total, count, pipe = total + t, count + c, None
                if pipe is None and DCRaggedPipeline.why_not(self.model, B, 2 * (F - 1)) is None:
                    pipe = DCRaggedPipeline(self.model, B, int(1.25 * max(n, hop * T)), 2 * (F - 1), hop, 40.0)
                if pipe is None:                  # (a last, smaller chunk the pipeline cannot take)
                    t, c = plain([item], False)
                    total, count = total + t, count + c
                    continue
                held.append(item)
                try:
                    est = pipe.push_features(feature_mix, ri, ragged[0], ragged[1], n)
                    if est is not None:
                        t, c = settle(est, held.pop(0))
                        total, count = total + t, count + c
                except XcdAborted as e:
                    t, c = recover(e)
                    total, count = total + t, count + c
            t, c = drain()
            total, count = total + t, count + c
        return total / max(count, 1)

    def get_est_sig(self, input, label, output, frames=None, lengths=None):
        from .features import mask_istft
        from .separation import dc_masks
        feature_mix, = input
        embedding, = output
        ri, sig_ref = _mix_ri(label)
        if self.host_kmeans and frames is not None:
            raise ValueError("tester_dc(host_kmeans=True) evaluates one utterance per forward (eval(batch=1))")
        if self.host_kmeans:
            import numpy as np
            from sklearn.cluster import KMeans
            masks = torch.zeros(tuple(feature_mix.shape) + (2,), device=feature_mix.device)
            for b in range(feature_mix.shape[0]):
                act = feature_mix[b] >= (feature_mix[b].max() - 40 / 20)
                lab = KMeans(n_clusters=2, random_state=0, n_init=10).fit_predict(embedding[b][act].cpu().numpy())
                lab = torch.from_numpy(lab.astype(np.int64)).to(feature_mix.device).float()
                masks[b][act] = torch.stack([lab, 1.0 - lab], -1)
        else:
            masks = dc_masks(embedding, feature_mix.float(), 40.0, frames=frames)
        return mask_istft(ri, masks, self.hop_size, sig_ref.shape[-1], frames=frames, lengths=lengths), sig_ref.float()


class tester_chimera(tester):
    """egs/wsj0-2mix/chimera/evaluate.py:23-45: the network's masks, mask-apply + iSTFT."""

    def __init__(self, args, hop_size=64):
        super().__init__(args)
        self.hop_size = hop_size

    def get_est_sig(self, input, label, output, frames=None, lengths=None):
        from .features import mask_istft
        _, mask_A, mask_B = output
        ri, sig_ref = _mix_ri(label)
        masks = torch.stack([mask_A, mask_B], -1)
        return mask_istft(ri, masks, self.hop_size, sig_ref.shape[-1], frames=frames, lengths=lengths), sig_ref.float()
