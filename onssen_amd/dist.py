"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm;
"gloo" in the CPU tests).

Inference shards *utterances*: a BLSTM cannot be split along time (the reverse direction needs the
whole chunk) and splitting the hidden units would put a collective inside every one of the T serial
steps, while utterances are independent -- so the data path has no collective at all.
Training is data parallel with one exchange per optimizer step: a bucketed all-reduce of the
gradients placed between ``backward()`` and ``clip_grad_norm_`` (onssen/utils/train.py:82-83), so
that clipping sees the global gradient.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced slice of ``n_items`` utterances for ``rank`` (sizes differ by <= 1)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_utterances(local, n_items, world, group=None):
    """Concatenate per-rank result tensors (dim 0 = utterances of ``shard_range``) on every rank."""
    if world == 1:
        return local
    sizes = [shard_range(n_items, r, world) for r in range(world)]
    pad = max(hi - lo for lo, hi in sizes)
    buf = local.new_zeros((pad,) + tuple(local.shape[1:]))
    buf[:local.shape[0]] = local
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    return torch.cat([o[:hi - lo] for o, (lo, hi) in zip(outs, sizes)], dim=0)


def gradient_buckets(model):
    """Parameters grouped so that a bucket becomes ready as one unit during backward: one bucket per
    LSTM (layer, direction) -- reversed registration order ~ the order gradients are produced -- and one
    for everything else (heads, BatchNorm).  ~11.5-23 MB each at H=600: large enough for xGMI's
    per-link bandwidth, small enough to overlap with the remaining backward."""
    lstm, rest = {}, []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        key = None
        if "weight_ih_l" in name or "weight_hh_l" in name or "bias_ih_l" in name or "bias_hh_l" in name:
            key = name.split(".")[:-1] + [name.split("_l")[-1]]          # module path + "<k>[_reverse]"
            key = ".".join(key)
        if key is None:
            rest.append(p)
        else:
            lstm.setdefault(key, []).append(p)
    buckets = [ps for _, ps in sorted(lstm.items(), reverse=True)]
    if rest:
        buckets.insert(0, rest)
    return buckets


def allreduce_gradients(model, world, group=None, buckets=None):
    """Average gradients across ranks AFTER backward: one flat fp32 all-reduce(sum) per bucket, issued
    asynchronously, then scaled by 1/world and scattered back.  (The non-overlapped form; ``GradientReducer`` issues
    the same buckets while backward is still running.)"""
    if world == 1:
        return
    buckets = buckets if buckets is not None else gradient_buckets(model)
    pending = []
    for ps in buckets:
        ps = [p for p in ps if p.grad is not None]
        if not ps:
            continue
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, ps))
    for work, flat, ps in pending:
        work.wait()
        flat.div_(world)
        off = 0
        for p in ps:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n


class GradientReducer:
    """The exchange of ``train_step`` overlapped with backward ("bucket by LSTM layer so layer-L buckets
    reduce while layer L-1 back-propagates"; slot: onssen/utils/train.py:82-83).

    A bucket's all-reduce is issued the moment its last gradient exists:
      * heads / BatchNorm (they sit on top of the stack: their gradients come first) and, on the ATen path, the LSTM
        parameters: ``register_post_accumulate_grad_hook`` per parameter, a countdown per bucket;
      * the HIP BLSTM backward (one autograd Function for the whole stack, nn/_train.py) hands every layer's fresh
        gradient tensors to ``layer_hook`` as soon as that layer's contractions are queued -- the reduce of layer l then
        runs on RCCL's stream under the backward recurrence of layer l-1 -- and collects them, averaged in place, before
        it returns them to autograd.
    ``finish()`` (before clipping) waits for what is still in flight and writes the averages into ``p.grad``; a bucket that
    never became complete -- some parameter of it took no part in this loss (chimera trained with ``loss_dc`` only leaves
    ``fc_mi`` without a gradient) -- is reduced there, over the parameters that DO have a gradient, as
    ``allreduce_gradients`` does: the autograd graph is the same on every rank, so every rank issues the same
    collectives in the same order and replicas cannot drift apart silently.  The hooks act only between ``begin()`` and
    ``finish()``: a backward outside ``train_step`` issues nothing.
    ``issued_in_backward`` counts the buckets issued before ``finish()`` was called -- the tests assert on it."""

    def __init__(self, model, world, group=None):
        self.world, self.group = world, group
        self.buckets = gradient_buckets(model)
        self.pending, self.layer_pending = [], []
        self.issued_in_backward = 0
        self._count = [0] * len(self.buckets)
        self._done = [False] * len(self.buckets)
        self._active = False
        self._handles = []
        if world > 1:
            for bi, ps in enumerate(self.buckets):
                for p in ps:
                    self._handles.append(p.register_post_accumulate_grad_hook(lambda p, bi=bi: self._ready(bi)))

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    def begin(self):
        self._count = [0] * len(self.buckets)
        self._done = [False] * len(self.buckets)
        self.pending, self.layer_pending = [], []
        self.issued_in_backward = 0
        self._active = True

    def _issue(self, tensors):
        flat = torch.cat([t.reshape(-1) for t in tensors])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.issued_in_backward += 1
        return work, flat, tensors

    def _ready(self, bi):
        if not self._active:                  # a backward outside begin() .. finish(): nobody would wait for the reduce
            return
        self._count[bi] += 1
        ps = [p for p in self.buckets[bi] if p.requires_grad]
        if self._count[bi] == len(ps):
            self._done[bi] = True
            if any(getattr(p, "_onssen_reduced", False) for p in ps):     # reduced inside the HIP backward already
                for p in ps:
                    p._onssen_reduced = False
                return
            self.pending.append(self._issue([p.grad for p in ps]))

    # -- called by BLSTMTrainFunction.backward (nn/_train.py) --------------------------------------------------
    def layer_hook(self, grad_tensors, params):
        if self.world > 1:
            self.layer_pending.append(self._issue(grad_tensors))
            for p in params:
                p._onssen_reduced = True

    def layer_collect(self):
        for work, flat, tensors in self.layer_pending:
            work.wait()
            flat.div_(self.world)
            off = 0
            for t in tensors:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
        self.layer_pending = []

    def finish(self):
        issued = self.issued_in_backward
        self._active = False
        self.layer_collect()
        # buckets that never completed (a parameter without a gradient in this loss): reduce what exists, in bucket order
        for bi, ps in enumerate(self.buckets):
            if self._done[bi]:
                continue
            if any(getattr(p, "_onssen_reduced", False) for p in ps):     # this layer went through layer_hook
                for p in ps:
                    p._onssen_reduced = False
                continue
            grads = [p.grad for p in ps if p.requires_grad and p.grad is not None]
            if grads:
                self.pending.append(self._issue(grads))
        for work, flat, tensors in self.pending:
            work.wait()
            flat.div_(self.world)
            off = 0
            for t in tensors:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
        self.pending = []
        self.issued_in_backward = issued


def _reducer_for(model, world, group):
    r = getattr(model, "_onssen_reducer", None)
    if r is None or r.world != world or r.group is not group:
        if r is not None:
            r.close()
        r = GradientReducer(model, world, group)
        object.__setattr__(model, "_onssen_reducer", r)
    return r


# ---- RCCL next to the persistent recurrences ---------------------------------------------------------------------------
# A recurrence exchange group needs 30 of its XCD's 32 CUs resident at the same time; GradientReducer
# runs RCCL's all-reduces UNDER the backward recurrence of the layer below.  RCCL launches one workgroup per channel and
# the dispatcher deals workgroups round-robin over the 8 XCDs, so <= 16 channels = <= 2 workgroups per XCD = the 2 spare
# CUs, whichever of the two kernels arrives first (measured with a spinning co-tenant: profiles/r03_cotenant_probe.txt).
RCCL_MAX_CHANNELS = 16
# A co-tenant that holds more (or a rank that waits inside an all-reduce for a slower peer while its CUs are held) only
# DELAYS the recurrence's start-up barrier; the bounded wait must outlast ordinary rank skew: ~4 s instead of ~0.2 s.
TRAIN_SPIN_LIMIT = 8_000_000


def configure_rccl(max_channels=RCCL_MAX_CHANNELS):
    """Cap RCCL's channel count BEFORE the first communicator exists (``NCCL_MAX_NCHANNELS`` / ``NCCL_MIN_NCHANNELS``; an
    explicit setting in the environment wins).  Called by ``init_process_group``."""
    import os
    os.environ.setdefault("NCCL_MAX_NCHANNELS", str(max_channels))
    if int(os.environ.get("NCCL_MIN_NCHANNELS", "1")) > int(os.environ["NCCL_MAX_NCHANNELS"]):
        os.environ["NCCL_MIN_NCHANNELS"] = os.environ["NCCL_MAX_NCHANNELS"]
    return int(os.environ["NCCL_MAX_NCHANNELS"])


def init_process_group(backend="nccl", **kwargs):
    """``torch.distributed.init_process_group`` for this package's training loop: one process per GPU over RCCL
    (backend "nccl" IS RCCL on ROCm), with RCCL's CU budget capped so that its kernels fit beside the persistent
    recurrences (``configure_rccl``) and the recurrences' bounded waits raised to outlast rank skew."""
    if backend == "nccl":
        configure_rccl()
    dist.init_process_group(backend, **kwargs)
    if backend == "nccl" and torch.cuda.is_available():
        from .hip import get_lib
        lib = get_lib()
        if lib.dll.onssen_xcd_spin_limit(-1) < TRAIN_SPIN_LIMIT:
            lib.dll.onssen_xcd_spin_limit(TRAIN_SPIN_LIMIT)


def _abort_group(group=None):
    """Tear down this rank's communicator after a failure in the middle of a step's collectives (see ``train_step``)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    try:
        from torch.distributed.distributed_c10d import _abort_process_group
        _abort_process_group(group)
    except Exception:             # (gloo, or a torch without the private hook: the peers' timeouts are what is left)
        pass


def _forward_backward(model, optimizer, loss_fn, input, label, reducer):
    from .nn import _train
    loss = None
    from . import loss as _loss_mod, options
    if (getattr(model, "fused_loss_dc", None) is not None and loss_fn is _loss_mod.loss_dc and options.get("loss") == "1"
            and not model._forward_hooks and not model._forward_pre_hooks):
        # this step holds the labels while the forward runs: head + loss as one autograd node (nn/deep_clustering.fused_loss_dc).
        # It does not go through nn.Module.__call__: a model with forward (pre-)hooks, another loss function, or
        # loss = "torch" (ONSSEN_LOSS_HIP=0) takes the plain model(input) -> loss_fn(output, label) route below
        loss = model.fused_loss_dc(input, label)
    if loss is None:
        loss = loss_fn(model(input), label)
    loss_avg = torch.mean(loss)
    optimizer.zero_grad()
    if reducer is not None:
        reducer.begin()
    _train.LAYER_GRAD_REDUCER[0] = reducer
    try:
        loss_avg.backward()
    finally:
        _train.LAYER_GRAD_REDUCER[0] = None
    if reducer is not None:
        reducer.finish()
    return loss_avg


def train_step(model, optimizer, loss_fn, input, label, world=1, group=None, clip_norm=5.0):
    """One optimizer step in the order of onssen/utils/train.py:75-86 with the data-parallel
    exchange inserted before gradient clipping (issued bucket by bucket DURING backward, see GradientReducer).
    Returns the local mean loss (float).

    An aborted persistent recurrence launch (forward or backward; ``nn/_core._XcdPolicy``) never reaches the weights: the
    status words are examined BEFORE the optimizer step; with ``world > 1`` the ranks agree on the outcome through a
    one-element MAX all-reduce of a status code (0 = fine, 1 = somebody's persistent launch aborted -- or saw non-finite
    activations (the default ``nonfinite="propagate"``: the re-run propagates them like nn.LSTM, as the reference would) --,
    2 = somebody hit a fatal error: non-finite activations in the strict ``nonfinite="raise"`` mode, a second abort) -- an aborted rank's garbage is
    already inside everybody's averaged gradients --, and then EVERY rank either runs forward / backward / exchange again
    (the rank that aborted on the launch-per-step HIP recurrences, which also propagate NaNs like nn.LSTM: the training
    path never leaves the library, round 4) or raises.  That agreement covers what the status words report.  An
    EXCEPTION thrown inside forward / backward on one rank (a library error code, a HIP runtime error) is different: this
    rank may have issued only some of the step's bucket all-reduces, so nothing it could issue next would pair with what
    its peers are waiting in.  It aborts its communicator (round 5) and re-raises; the peers' pending collectives then fail
    at the process group's timeout instead of pairing with an unrelated collective or waiting forever.  BatchNorm's running
    statistics are put back before a re-run; the persistent form stays enabled for the next step."""
    from . import _abi
    from .nn._core import XcdAborted, _XcdPolicy, _XcdStatus
    reducer = _reducer_for(model, world, group) if world > 1 else None
    if reducer is None and getattr(model, "_onssen_reducer", None) is not None:     # left over from a world > 1 step
        model._onssen_reducer.close()
        object.__setattr__(model, "_onssen_reducer", None)
    on_gpu = input[0].is_cuda
    bufs = [b.detach().clone() for b in model.buffers()] if on_gpu and model.training else None

    def agree(err, fatal_if_aborted=False):
        """(status over all ranks, this rank's exception): 0 fine, 1 re-run, 2 fatal."""
        code = 0 if err is None else 1 if isinstance(err, XcdAborted) and not fatal_if_aborted else 2
        if world > 1 and on_gpu:
            flag = torch.tensor([float(code)], device=input[0].device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            code = int(flag.item())
        return code

    def examine():
        if not on_gpu:
            return None
        try:
            _XcdStatus.flush()        # aborted exchange / non-finite activations: examined here, not after the update
        except _abi.OnssenError as e:
            return e
        return None

    def guarded():
        try:
            return _forward_backward(model, optimizer, loss_fn, input, label, reducer)
        except BaseException:
            if world > 1:
                if reducer is not None:
                    reducer._active = False
                _abort_group(group)
            raise

    loss_avg = guarded()
    err = examine()
    code = agree(err)
    if code == 2:
        raise err if err is not None else _abi.OnssenError("onssen_amd.train_step: another rank reported a fatal error in this step")
    if code == 1:
        import contextlib
        import warnings
        _XcdPolicy.recovered += 1
        warnings.warn(f"onssen_amd.train_step: {err or 'a persistent recurrence aborted on another rank'}  "
                      "Re-running forward / backward of this step" + (" on the launch-per-step recurrences." if err else "."), RuntimeWarning)
        if bufs is not None:
            with torch.no_grad():
                for b, old in zip(model.buffers(), bufs):
                    b.copy_(old)
        with (_XcdPolicy.forced_steps() if err is not None else contextlib.nullcontext()):
            loss_avg = guarded()
        err = examine()               # a second abort (only possible on a rank that did not abort the first time) is fatal,
        if agree(err, fatal_if_aborted=True) == 2:     # ... and every rank learns of it before anybody raises
            raise err if err is not None else _abi.OnssenError("onssen_amd.train_step: another rank failed in the re-run of this step")
    if hasattr(optimizer, "step_clipped"):        # utils.ClipAdam: the clipping rides in the optimizer's own two passes
        optimizer.step_clipped(clip_norm)
    else:
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip_norm)
        optimizer.step()
    from .nn._core import invalidate_packed_weights
    invalidate_packed_weights()          # (a fused optimizer moves the parameters without bumping their versions)
    return float(loss_avg.item())
