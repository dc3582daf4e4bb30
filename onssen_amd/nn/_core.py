"""Shared host-side plumbing of the onssen_amd.nn modules: parameter
containers with the reference's state_dict layout, weight packing for the HIP
kernels, workspace caching and the launch sequences.

PyTorch is used for device memory, streams and (training only) autograd; the
inference arithmetic is in libonssen_hip.so.
"""
import math
import os

import torch
import torch.nn as nn

from .. import _abi, options
from ..hip import get_lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def precision():
    """Arithmetic of the MFMA contractions (ONSSEN_PRECISION):
    "bf16x3" (default) = split-bf16: every fp32 product is three bf16 MFMAs with fp32 accumulation, ~1e-5
    relative per dot product; embeddings match the reference to <=4e-6 abs at the BASELINE configs;
    "f32" = exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), ~1e-6;
    "bf16" = OPT-IN reduced precision (BASELINE cfg2's literal dtype): plain bf16 products with fp32 accumulation in
    the XCD-form recurrence and the image GEMMs, bf16-grade results (~1e-2 relative), outside the 1e-4 parity
    contract; gates, cell state, normalisation, FFTs stay fp32."""
    p = options.get("precision")
    if p not in ("f32", "bf16x3", "bf16"):
        raise ValueError(f"ONSSEN_PRECISION={p!r}: expected 'f32', 'bf16x3' or 'bf16'")
    return p


def _split_bf16():
    return precision() in ("bf16x3", "bf16")


class XcdAborted(_abi.OnssenError):
    """A bounded wait of a persistent recurrence launch gave up: the outputs (and, in training, the gradients) of the call
    that contained it are invalid.  ``separation.separate_*``, ``evaluate.tester.eval`` and ``dist.train_step`` catch it and
    re-run that call on the launch-per-step / ATen recurrence (``_XcdPolicy``); a bare ``model(x)`` raises it."""


class XcdNonFinite(XcdAborted):
    """ONSSEN_NONFINITE=propagate: a NaN / Inf reached the persistent recurrence (it cannot pass the tagged exchange and was
    replaced by 0).  Raised INSTEAD of the plain error so that the entry points that re-run aborted calls re-run this one on
    the launch-per-step recurrence, which propagates non-finite values exactly like nn.LSTM does (the reference's
    semantics: NaN in, NaN out).  Not an abort: no back-off."""


class StalePackedWeights(XcdAborted):
    """An inference forward ran on packed weight images that no longer belong to the live parameters: something moved the weights
    without bumping ``tensor._version`` and without ``invalidate_packed_weights()`` -- a FUSED torch optimizer built by hand,
    ``p.data`` arithmetic, a foreign kernel -- while the model stayed in eval mode.  Found by the weight guard (``_WeightGuard``: a
    device-side sampled checksum, compared asynchronously); by the time this is raised the stale images are already dropped, so the
    call only has to be repeated: ``separation.separate_*`` and ``evaluate.tester.eval`` do that by themselves (``recovering``), a
    bare ``model(x)`` raises it at the next forward / ``flush()`` like an aborted launch.  Not an abort: no back-off, and the re-run
    stays on the persistent kernels."""


class _XcdPolicy:
    """What happens after an aborted persistent launch (a co-tenant kernel held more than the 2 spare CUs of an XCD for
    longer than the bounded wait, two persistent launches overlapped, ...): the abort is *per launch*, not a property of
    the process, so the persistent form stays enabled --

      * the call that aborted is re-run by its owner inside ``forced_steps()`` (the launch-per-step recurrences, inference
        and training alike), see ``recovering``;
      * the next call tries the persistent form again; only consecutive aborts back off: after the k-th abort in a row
        the next 2^(k-1) - 1 BLSTM launches (at most 63) skip it, a persistent launch that completes resets the streak.
    Counters are for tests / bench output."""
    skip = 0               # BLSTM stack launches that still avoid the persistent form
    streak = 0             # aborts since the last persistent launch that completed
    aborts = 0
    recovered = 0          # calls re-run after an abort
    force_steps = 0        # depth of forced_steps() scopes
    persistent_launches = 0
    fallback_launches = 0

    @classmethod
    def persistent_allowed(cls):
        return cls.force_steps == 0 and cls.skip == 0

    @classmethod
    def note_launch(cls, persistent):
        if persistent:
            cls.persistent_launches += 1
        else:
            cls.fallback_launches += 1
            if cls.force_steps == 0 and cls.skip > 0:
                cls.skip -= 1

    @classmethod
    def on_abort(cls):
        cls.aborts += 1
        cls.streak += 1
        cls.skip = min(2 ** (cls.streak - 1) - 1, 63)

    @classmethod
    def on_ok(cls):
        cls.streak = 0

    @classmethod
    def forced_steps(cls):
        import contextlib

        @contextlib.contextmanager
        def scope():
            cls.force_steps += 1
            try:
                yield
            finally:
                cls.force_steps -= 1
        return scope()


def recovering(fn):
    """Decorator of the entry points that end with ``_XcdStatus.flush()``: a call whose persistent recurrence aborted is
    run again on the launch-per-step recurrence (its inputs are untouched: inference is functional), with a warning."""
    import functools
    import warnings

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        try:
            return fn(*args, **kwargs)
        except StalePackedWeights as e:            # the images are already dropped: the same call again, on the same kernels
            _XcdPolicy.recovered += 1
            warnings.warn(f"onssen_amd: {e}  Re-running this call on freshly packed weights.", RuntimeWarning)
            return fn(*args, **kwargs)
        except XcdAborted as e:
            _XcdPolicy.recovered += 1
            if not isinstance(e, XcdNonFinite):
                warnings.warn(f"onssen_amd: {e}  Re-running this call on the launch-per-step recurrence.", RuntimeWarning)
            with _XcdPolicy.forced_steps():
                return fn(*args, **kwargs)
    return wrapper


def persistent_capable(H):
    """Does the XCD-local persistent INFERENCE recurrence hold a layer of H units?  Up to 640 in every precision mode; up to 768
    in the default split-bf16 mode (round 4: 32 members of 24 units -- every CU of an XCD --, W_hh in 144 of the 256 registers)."""
    return H <= 640 or (H <= 768 and precision() == "bf16x3")


def recurrence_plan(B, H):
    """(ug, flags) for onssen_blstm_forward_f32.

    H <= 640 (768 in split-bf16): the XCD-local persistent recurrence (one launch per layer, every (direction, 4 / 8 / 16-row group)
    inside one XCD, at most 32 unit groups -> ug = 4*ceil(H/128)) -- split-bf16 / bf16 products on x3 images, or
    (precision f32, round 3) exact-fp32 MFMAs on fp32 images.  Otherwise one
    launch per time step with 8 hidden units per workgroup (split-bf16 up to H = 640, exact fp32 above).  ONSSEN_XCD=0 forces the per-step form (so does a
    ``_XcdPolicy.forced_steps()`` scope or a back-off after consecutive aborts), ONSSEN_UG overrides its unit-group
    size, ONSSEN_ABLATE sets the profiling-only ablation bits."""
    flags = int(options.get("ablate")) << 8
    xcd = persistent_capable(H) and options.get("recurrence") == "1" and _XcdPolicy.persistent_allowed()
    x3 = _split_bf16() and (H <= 640 or xcd)
    if x3:
        flags |= _abi.BLSTM_BF16X3
    if xcd:
        # (precision f32: the same persistent launch in exact fp32 -- flags carry BLSTM_XCD without BLSTM_BF16X3)
        if precision() == "bf16":
            flags |= _abi.BLSTM_BF16
        ug = int(options.get("recurrence_unit_group"))      # 0 = the default; 24 at H <= 640: the 25-member A/B form (round 5)
        return (ug if ug > 0 and precision() == "bf16x3" else 4 * -(-H // 128)), flags | _abi.BLSTM_XCD
    if precision() == "bf16":
        raise RuntimeError("ONSSEN_PRECISION=bf16 exists only in the XCD-form recurrence (H <= 640, ONSSEN_XCD=1)")
    if options.get("split_rows") == "1":
        flags |= _abi.BLSTM_SPLIT_ROWS
    return int(options.get("step_unit_group")), flags


class _XcdSerial:
    """ONSSEN_XCD_SERIALIZE=1 (opt-in): one persistent launch in flight per device.  The persistent kernels' members spin on
    each other and need their whole exchange group resident; two of them issued on DIFFERENT streams of one process can be
    dispatched interleaved, each holding CUs the other's groups need, until the bounded waits give up (both abort, both are
    re-run: correct, but a 0.2 s latency spike).  With the switch on, every persistent launch records an event behind itself and
    a persistent launch on another stream first makes its stream wait for it (stream-ordered on the device, no host
    synchronisation).  Off by default because the hazard is rare and the overlap is worth having: exchange groups are
    independent chains, a group that finds its XCD taken simply starts when the other kernel's group has finished
    (`tools/two_stream_probe.py`: two models on two streams, 20 rounds: 39.5 ms unserialised with no abort, 51.5 ms serialised).
    Not applied while a hipGraph is being captured.  Other PROCESSES on the same GPU are outside its reach: see _XcdPolicy /
    INTEGRATION.md."""
    last = {}                # device index -> (raw stream handle, event behind the last persistent launch)

    @staticmethod
    def enabled():
        return options.get("serialize_persistent") == "1"

    @classmethod
    def before(cls, device):
        if not cls.enabled() or torch.cuda.is_current_stream_capturing():
            return
        rec = cls.last.get(device.index)
        if rec is not None:
            cur = torch.cuda.current_stream(device)
            if rec[0] != cur.cuda_stream:
                cur.wait_event(rec[1])

    @classmethod
    def after(cls, device):
        if not cls.enabled() or torch.cuda.is_current_stream_capturing():
            return
        cur = torch.cuda.current_stream(device)
        rec = cls.last.get(device.index)
        ev = rec[1] if rec is not None else torch.cuda.Event()     # one event per device, re-recorded
        ev.record(cur)
        cls.last[device.index] = (cur.cuda_stream, ev)


class _WeightGuard:
    """Device-side sampled checksum of the source tensors of a set of packed weight images (csrc/optim.inc: param_guard_kernel).
    ``arm(tensors)`` when the images are built; ``check()`` whenever they are REUSED by an inference forward: one launch of
    len(tensors) small workgroups that raises a sticky device flag on a mismatch, fetched asynchronously with the persistent
    kernels' status words (``_XcdStatus.post_guard`` / ``poll``) -- no synchronisation.  Eager forwards only: nothing is added to a
    hipGraph capture (a graph is bound to the images it captured; re-capture after the weights change).
    ``weight_guard`` / ONSSEN_WEIGHT_GUARD=0 disables it."""
    SAMPLES = 2048
    stale_seen = 0
    tick = 0               # bumped once per BLSTM stack forward (run_blstm): a guard checks at most once per tick
                           # (the heads' images are fetched more than once per forward on some paths)

    def __init__(self, what):
        self.what = what
        self.n = 0
        self.last_tick = -1
        self.posted = False  # one status copy in flight per guard: the device flag is sticky, a later copy still finds it

    def arm(self, tensors):
        if options.get("weight_guard") != "1" or not tensors or not tensors[0].is_cuda:
            self.n = 0
            return
        dev = tensors[0].device
        live = [t.detach() for t in tensors if t.numel() > 0 and t.element_size() == 4 and t.is_contiguous()]
        self.keep = live                                                     # the table holds raw pointers
        self.n = len(live)
        if self.n == 0:
            return
        self.ptrs = torch.tensor([t.data_ptr() for t in live], dtype=torch.int64).to(dev)
        self.numel = torch.tensor([t.numel() for t in live], dtype=torch.int64).to(dev)
        self.ref = torch.zeros(self.n, dtype=torch.int32, device=dev)
        self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        get_lib().param_guard(self.ptrs.data_ptr(), self.numel.data_ptr(), self.n, self.SAMPLES, 0, self.ref.data_ptr(), self.flag.data_ptr(), _stream())

    def check(self):
        if self.n == 0 or self.last_tick == _WeightGuard.tick or options.get("weight_guard") != "1":
            return
        if torch.cuda.is_current_stream_capturing():
            return       # a captured graph points at THESE images whatever the parameters do later (re-capture after an update): the check
                         # would cost every replay two dependent launches (measured +0.018 ms on the 1.86 ms headline step) and could not be acted on
        self.last_tick = _WeightGuard.tick
        get_lib().param_guard(self.ptrs.data_ptr(), self.numel.data_ptr(), self.n, self.SAMPLES, 1, self.ref.data_ptr(), self.flag.data_ptr(), _stream())
        if not self.posted:
            _XcdStatus.post_guard(self)


class _XcdStatus:
    """The persistent kernels bound every wait and report through workspace words instead of hanging:
    [280] != 0 -> a wait gave up, the launch aborted, outputs are invalid; [281] == 1 -> some exchange group was
    spread over several XCDs and used the slower placement-independent accesses; [282] == 1 -> a non-finite h was seen
    (a NaN cannot carry the exchange's data tag: it was replaced by 0, so the outputs are NOT the reference's NaNs).
    The words are fetched with an asynchronous copy after each eager forward and examined
      * by ``flush()`` -- called where a result is about to be consumed anyway: the end of ``separation.separate_*``,
        ``dist.train_step`` BEFORE the optimizer step (an aborted forward or backward must never reach the weights), at
        interpreter exit, and after every forward when ONSSEN_CHECK=1 (tests);
      * otherwise at the next forward (never inside a graph capture), so that a plain ``model(x)`` does not pay a
        synchronisation per call.
    On an abort the words are reset (a stale abort word would abort every later launch on that workspace) and
    ``XcdAborted`` is raised; what happens next is ``_XcdPolicy``'s business (the call is re-run on the launch-per-step
    recurrence by its owner, the persistent form stays enabled)."""
    pending = []
    safe_protocol_seen = False

    @classmethod
    def post(cls, wsb):
        if torch.cuda.is_current_stream_capturing():
            return
        host = torch.empty(3, dtype=torch.int32).pin_memory()
        host.copy_(wsb[1120:1132].view(torch.int32), non_blocking=True)     # u32 words 280, 281, 282
        ev = torch.cuda.Event()
        ev.record()
        cls.pending.append((ev, host, wsb))

    @classmethod
    def post_guard(cls, guard):
        """Flag word of a weight guard (``_WeightGuard.check``): non-zero = the parameters no longer match the packed images."""
        if torch.cuda.is_current_stream_capturing():
            return
        host = torch.empty(1, dtype=torch.int32).pin_memory()
        host.copy_(guard.flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        guard.posted = True
        cls.pending.append((ev, host, guard))

    @classmethod
    def post_cluster(cls, ws, offset):
        """Status word of the persistent Lloyd kernel (separation.dc_masks): non-zero = a bounded wait gave up."""
        if torch.cuda.is_current_stream_capturing():
            return
        host = torch.empty(1, dtype=torch.int32).pin_memory()
        host.copy_(ws[offset:offset + 4].view(torch.int32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        cls.pending.append((ev, host, (ws, offset)))

    @classmethod
    def poll(cls, wait=False, policy=True):
        """Examine the recorded status words whose copies have landed.  The back-off policy is updated ONCE per call -- an
        abort if any examined launch aborted, otherwise "ok" if a persistent recurrence completed (a completed launch of
        another workspace in the same poll must not reset the streak an abort has just extended) -- and not at all with
        ``policy=False`` (draining the reports of launches that ran into an abort word that was already counted)."""
        keep, err = [], None
        saw_abort, saw_ok = False, False
        for ev, host, wsb in cls.pending:
            if wait:
                ev.synchronize()
            if not ev.query():
                keep.append((ev, host, wsb))
                continue
            if isinstance(wsb, _WeightGuard):           # weight guard: the live parameters differ from the packed images' source
                wsb.posted = False
                if int(host[0]) != 0:
                    wsb.flag.zero_()
                    invalidate_packed_weights()         # every image of the process is rebuilt at its next use
                    _WeightGuard.stale_seen += 1
                    err = err or StalePackedWeights(
                        f"packed weight images of {wsb.what} were stale: the parameters changed without a version bump (a fused "
                        "optimizer step, p.data arithmetic, ...) while the model stayed in eval mode, and at least one forward ran on "
                        "the old images.  They have been dropped; repeat the call (model.repack() after such updates avoids this).")
                continue
            if isinstance(wsb, tuple):                  # clustering status word
                if int(host[0]) != 0:
                    ws, off = wsb
                    ws[off:off + 4].zero_()
                    saw_abort = True
                    err = err if isinstance(err, XcdAborted) else XcdAborted(
                        f"persistent 2-means launch gave up a bounded wait (code {int(host[0]) & 0xffffffff:#x}: its workgroups were not "
                        "co-resident): the masks of that call are not the converged ones.")
                continue
            if int(host[1]) == 1:
                cls.safe_protocol_seen = True
            if int(host[0]) == 0:
                saw_ok = True
            if int(host[0]) != 0 or int(host[2]) != 0:
                wsb[1120:1132].zero_()                  # abort / non-finite words: the next launch starts clean
                if int(host[0]) != 0:
                    # an aborted launch leaves the monotonic start-up state inconsistent (arrivals counted, generation not
                    # advanced): the next persistent launch on this workspace would walk through its start-up barrier.
                    # Nothing is in flight here (the event has fired): the owner zeroes the whole exchange header again.
                    wsb[:_abi.BLSTM_WS_HEADER].zero_()
                    saw_abort = True
                    err = err if isinstance(err, XcdAborted) else XcdAborted(
                        f"XCD-local persistent recurrence aborted (code {int(host[0])}): a bounded wait gave up, the outputs "
                        "(and, in training, the gradients) of that call are invalid.")
                elif options.get("nonfinite") == "propagate":
                    err = err or XcdNonFinite(
                        "non-finite activations inside the persistent recurrence: re-run on the launch-per-step recurrence, "
                        "which propagates them like nn.LSTM (ONSSEN_NONFINITE=propagate).")
                else:
                    err = err or _abi.OnssenError(
                        "non-finite activations inside the persistent recurrence (NaN / Inf in the input or the weights): "
                        "they cannot pass its tagged exchange and were replaced by 0, so the outputs of that call are not the "
                        "reference's NaNs.  ONSSEN_XCD=0 (launch per step) propagates them like nn.LSTM; ONSSEN_NONFINITE=propagate "
                        "makes separate_* / tester.eval / train_step re-run such a call that way by themselves.")
        cls.pending = keep
        if policy:
            if saw_abort:
                _XcdPolicy.on_abort()                   # once per poll: the launches of a call (and of its window) abort together
            elif saw_ok:
                _XcdPolicy.on_ok()
        if err is not None:
            raise err

    @classmethod
    def flush(cls, policy=True):
        """Wait for every recorded status and raise if a launch aborted (see the class docstring)."""
        if cls.pending and not torch.cuda.is_current_stream_capturing():
            cls.poll(wait=True, policy=policy)


def _flush_at_exit():
    try:
        _XcdStatus.flush()
    except _abi.OnssenError as e:       # nobody is left to catch it: at least say so
        import sys
        print(f"onssen_amd: {e}", file=sys.stderr)


import atexit
atexit.register(_flush_at_exit)


class BLSTMParams(nn.Module):
    """Parameter container with nn.LSTM(bidirectional=True)'s names, shapes
    and default init, so reference checkpoints load unchanged:
    weight_ih_l{k}[_reverse] (4H,in_k), weight_hh_l{k}[_reverse] (4H,H),
    bias_ih_l{k}[_reverse], bias_hh_l{k}[_reverse] (4H); rows i,f,g,o."""

    def __init__(self, input_size, hidden_size, num_layers, dropout=0.0):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.num_layers, self.dropout = num_layers, float(dropout)
        bound = 1.0 / math.sqrt(hidden_size)
        for k in range(num_layers):
            in_k = input_size if k == 0 else 2 * hidden_size
            for sfx in ("", "_reverse"):
                for name, shape in ((f"weight_ih_l{k}{sfx}", (4 * hidden_size, in_k)),
                                    (f"weight_hh_l{k}{sfx}", (4 * hidden_size, hidden_size)),
                                    (f"bias_ih_l{k}{sfx}", (4 * hidden_size,)),
                                    (f"bias_hh_l{k}{sfx}", (4 * hidden_size,))):
                    self.register_parameter(name, nn.Parameter(torch.empty(shape).uniform_(-bound, bound)))

    def flatten_parameters(self):
        """cuDNN weight-packing hint in the reference (deep_clustering.py:34); no-op here."""

    def flat_weights(self):
        out = []
        for k in range(self.num_layers):
            for sfx in ("", "_reverse"):
                out += [getattr(self, f"weight_ih_l{k}{sfx}"), getattr(self, f"weight_hh_l{k}{sfx}"),
                        getattr(self, f"bias_ih_l{k}{sfx}"), getattr(self, f"bias_hh_l{k}{sfx}")]
        return out

    def autograd_forward(self, x, training):
        """Training path (needs autograd).  On a ROCm device: ALWAYS the HIP kernels (nn/_train.py) -- the
        persistent forward with saved state (H <= 768) + the persistent backward recurrence (H <= 640) where they apply (no abort
        back-off in force), otherwise the launch-per-step forward with saved state (split-bf16 up to H = 640, exact fp32
        above) + the launch-per-step backward: the re-run of a training step whose persistent launch aborted and wide layers
        stay inside the library (round 4; rounds 1-3 fell back to the stock ATen / MIOpen LSTM there).  A CPU tensor RAISES (round 6)
        unless the test scaffolding switch ``cpu_autograd`` / ONSSEN_CPU_AUTOGRAD=1 is set (tests/conftest.py sets it for the gloo
        tests of the multi-process logic, which have no GPU to run on)."""
        if x.is_cuda:
            from ._train import BLSTMTrainFunction
            persistent = (self.hidden_size <= 768 and options.get("recurrence") == "1" and _XcdPolicy.persistent_allowed())
            if self.hidden_size <= 768 and options.get("recurrence") == "1":
                _XcdPolicy.note_launch(persistent)
            if getattr(self, "_train_packed", None) is None:
                object.__setattr__(self, "_train_packed", PackedBLSTM(self))
            p_drop = float(self.dropout) if training and self.num_layers > 1 else 0.0
            return BLSTMTrainFunction.apply(x, self._train_packed, p_drop, persistent, *self.flat_weights())
        if options.get("cpu_autograd") != "1":
            raise RuntimeError("onssen_amd: the BLSTM stack needs tensors on a ROCm device; there is no CPU fallback (ONSSEN_CPU_AUTOGRAD=1 "
                               "is test scaffolding for the multi-process logic over gloo, never a product path)")
        B = x.shape[0]
        z = x.new_zeros(2 * self.num_layers, B, self.hidden_size)
        out, _, _ = torch._VF.lstm(x, (z, z), self.flat_weights(), True, self.num_layers,
                                   self.dropout if training else 0.0, training, True, True)
        return out


_WEIGHT_EPOCH = [0]          # bumped by invalidate_packed_weights(): every packed image is rebuilt at its next use


def invalidate_packed_weights():
    """Drop every packed weight image of the process (they are rebuilt from the live parameters at the next forward).

    The packed images are keyed on (data_ptr, tensor._version, device) of their source parameters.  In-place updates
    through autograd-visible ops (``p.copy_()``, ``load_state_dict``, the for-loop / foreach optimizers) bump ``_version``
    and are seen; updates through ``p.data`` (``p.data.add_()``, EMA / weight-averaging code, hand-written checkpoint
    loaders) change neither the version nor the pointer -- call this (or ``module.repack()``) after such an update.
    **FUSED optimizers do not bump it either** (round 5, measured on torch 2.10 + ROCm: ``torch.optim.Adam(fused=True)``
    leaves ``_version`` at 1 after 300 steps; ``tools/micro/train_then_eval_diag.py``) -- and ``build_optimizer`` returns one.
    Rounds 3-4 therefore trained with the BLSTM images of step 0 in every forward (only the heads, whose training GEMMs
    split their weights per call, learned), and an eval-mode forward after training used whichever images an earlier
    eval-mode forward had left.  Since round 5 nothing relies on the version alone where the weights are expected to
    move: the training forward rebuilds its images every step (``PackedBLSTM.get(force=True)``), every optimizer from
    ``build_optimizer`` and ``dist.train_step`` call this after ``step()``, and switching a model between ``train()`` and
    ``eval()`` calls it.  ``load_state_dict`` and ``module.to()/.float()/...`` call it themselves;
    ``ONSSEN_CHECK_WEIGHTS=1`` adds a device-side checksum of the parameters to the key (one synchronisation per forward: a
    debugging aid that finds forgotten calls)."""
    _WEIGHT_EPOCH[0] += 1


_GRAPH_KEEPALIVE = []        # packed images a hipGraph capture has seen and that were rebuilt since: a captured graph holds raw pointers
                             # into them, so they are parked here instead of being returned to the allocator (the graph then replays on the
                             # OLD weights -- re-capture after the weights change -- but never on memory somebody else owns)


def _park_if_captured(owner, names):
    if getattr(owner, "_in_graph", False):
        _GRAPH_KEEPALIVE.append([getattr(owner, n, None) for n in names])
        owner._in_graph = False


def _version_key(tensors):
    key = tuple((t.data_ptr(), t._version, str(t.device)) for t in tensors) + (_WEIGHT_EPOCH[0],)
    if options.get("check_weights") == "1" and not torch.cuda.is_current_stream_capturing():
        key += (float(torch.stack([t.detach().double().abs().sum() for t in tensors]).sum()),)
    return key


class PackedWeightsMixin:
    """nn.Module mixin of the onssen_amd.nn models: keeps the packed weight images honest across the ways parameters
    change behind autograd's back (see invalidate_packed_weights)."""

    def _init_packed_hooks(self):
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: invalidate_packed_weights())

    def _apply(self, fn, *args, **kwargs):          # .to() / .cuda() / .float() / .half() ...
        out = super()._apply(fn, *args, **kwargs)
        invalidate_packed_weights()
        return out

    def train(self, mode=True):                     # .train() / .eval(): the weights may have moved under any optimizer since
        if bool(mode) != self.training:
            invalidate_packed_weights()
        return super().train(mode)

    def repack(self):
        """Rebuild the packed weight images from the live parameters at the next forward."""
        invalidate_packed_weights()
        return self

    invalidate = repack


class PackedBLSTM:
    """Device images of a BLSTMParams for the HIP kernels, rebuilt when any
    parameter changes (in-place optimizer steps bump ``_version``)."""

    def __init__(self, params: BLSTMParams):
        self.p = params
        self.cache = {}     # ug -> packed images (a batch-size class can prefer another unit-group size)

    def get(self, ug, force=False, lean=False):
        """``force``: rebuild whatever the version key says (the training forward: its parameters change every step, and a
        fused optimizer changes them WITHOUT bumping ``_version`` -- see invalidate_packed_weights).  ``lean``: only the
        images the persistent training path reads (no split planes of W_ih, no first-layer fragment image)."""
        img = self.cache.get(ug)
        if img is None:
            img = self.cache[ug] = _PackedImages(self.p, ug)
        return img.get(force, lean)


class _PackedImages:
    def __init__(self, params, ug):
        self.p, self.ug = params, ug
        self.key = None
        self._whhT = None
        self.guard = _WeightGuard(f"the BLSTM stack ({params.num_layers} x {params.hidden_size})")

    def get(self, force=False, lean=False):
        p, lib = self.p, get_lib()
        flat = p.flat_weights()
        key = _version_key(flat) + (bool(lean),)
        capturing = flat[0].is_cuda and torch.cuda.is_current_stream_capturing()
        if key == self.key and not force:
            self._in_graph = getattr(self, "_in_graph", False) or capturing
            self.guard.check()                   # (asynchronous: see _WeightGuard)
            return self
        _park_if_captured(self, ("wih", "whh", "bias", "whh_x3", "wih_x3", "wih_img", "wih_frag0", "bias0_tail", "_whhT"))
        self._in_graph = capturing
        dev = flat[0].device
        H, L = p.hidden_size, p.num_layers
        self.Hp, self.NP, self.KQ, we = lib.lstm_geometry(H, self.ug)
        self.wih, self.whh, self.bias, self.whh_x3, self.wih_x3, self.wih_img = [], [], [], [], [], []
        self.wih_frag0 = None
        self.bias0_tail = None
        _, _, we3 = lib.lstm_geometry_x3(H, self.ug)
        st = _stream()
        if lean:
            self._get_lean(flat, dev, H, L, we3, st)
            self.key = key
            self.guard.n = 0                     # (the training forward rebuilds its images every step: nothing to guard)
            return self
        for l in range(L):
            in_l = p.input_size if l == 0 else 2 * H
            Kp = (in_l + 3) // 4 * 4 if l == 0 else 2 * self.Hp
            a = torch.empty(2, self.NP, Kp, device=dev, dtype=torch.float32)
            b = None if lean else torch.empty(2, we, device=dev, dtype=torch.float32)
            c = torch.empty(2, self.NP, device=dev, dtype=torch.float32)
            K_l = in_l if l == 0 else 2 * self.Hp
            ai = torch.empty(2 * self.NP, (K_l + 31) // 32, 2, 32, device=dev, dtype=torch.int16)
            for d in range(2):
                w_ih, w_hh, b_ih, b_hh = [t.detach().contiguous() for t in flat[(2 * l + d) * 4:(2 * l + d) * 4 + 4]]
                if lean:     # the per-step pack of the training forward: packed W_ih + bias + its x3 image in one pass, no fp32 W_hh image
                    lib.lstm_pack_wih_image(w_ih.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(), in_l, 0 if l == 0 else 1, H, self.ug,
                                            a[d].data_ptr(), c[d].data_ptr(), ai[d * self.NP:].data_ptr(), st)
                else:
                    lib.lstm_pack(w_ih.data_ptr(), w_hh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(), in_l,
                                  0 if l == 0 else 1, H, self.ug, a[d].data_ptr(), b[d].data_ptr(), c[d].data_ptr(), st)
            b3 = torch.empty(2, we3, device=dev, dtype=torch.int16)
            for d in range(2):
                w_hh = flat[(2 * l + d) * 4 + 1].detach().contiguous()
                lib.lstm_pack_whh_bf16x3(w_hh.data_ptr(), H, self.ug, b3[d].data_ptr(), st)
            ld = (in_l if l == 0 else 2 * self.Hp)
            ld = (ld + 31) // 32 * 32
            a3 = None
            if not lean:     # (the launch-per-step split-bf16 GEMM's operand: the persistent path reads the x3 image below)
                a3 = torch.empty(2, 2 * self.NP, ld, device=dev, dtype=torch.int16)     # hi plane, lo plane
                lib.linear_pack_bf16x3(a.data_ptr(), 2 * self.NP, in_l if l == 0 else 2 * self.Hp, Kp, ld, a3.data_ptr(), st)
            # x3 image of the same matrix (ONSSEN_BLSTM_XCD form: pre-split operands, onssen_linear_x3p)
            if not lean:
                lib.x3_image(a.data_ptr(), Kp, 0, 1, 2 * self.NP, K_l, ai.data_ptr(), st)
            self.wih.append(a), self.whh.append(b), self.bias.append(c), self.whh_x3.append(b3), self.wih_x3.append(a3)
            self.wih_img.append(ai)
            if l == 0 and in_l <= 129 and self.ug <= 20 and not lean:   # fragment image for the fused first-layer input projection (FUSE_IN0: <= 4 MFMA k-chunks)
                kc = (in_l + 31) // 32
                f0 = torch.empty(2, (self.Hp // self.ug) * kc * (self.ug // 4) * 1024, device=dev, dtype=torch.int16)
                for d in range(2):
                    w_ih = flat[(2 * l + d) * 4].detach().contiguous()
                    lib.lstm_pack_wih_bf16x3(w_ih.data_ptr(), in_l, H, self.ug, f0[d].data_ptr(), st)
                self.wih_frag0 = f0
                # FUSE_TAIL (in_dim = 32k + 1): the bias followed by the last column of the packed W_ih
                self.bias0_tail = torch.cat([c.reshape(-1), a[:, :, in_l - 1].reshape(-1)]).contiguous() if in_l % 32 == 1 and in_l > 1 else None
        self.key = key
        self._whhT = None
        self.guard.arm(flat)
        return self

    def _get_lean(self, flat, dev, H, L, we3, st):
        """The per-step pack of the training forward (round 5): every image the persistent training path reads -- packed W_ih + bias
        + its x3 image, the W_hh fragment images of the forward recurrence and (H <= 640) the row-slice images of the backward
        recurrence -- for ALL layers and directions in ONE launch (onssen_lstm_pack_train_f32), no fp32 W_hh image, no split planes,
        no first-layer fragment image."""
        lib, p = get_lib(), self.p
        cont = [t.detach().contiguous() for t in flat]
        nR = lib.lstm_whhR_elems(H, self.ug) if H <= 640 else 0
        ptr = {k: [] for k in ("w_ih", "w_hh", "b_ih", "b_hh", "a", "c", "ai", "b3", "r")}
        whhR = []
        for l in range(L):
            in_l = p.input_size if l == 0 else 2 * H
            Kp = (in_l + 3) // 4 * 4 if l == 0 else 2 * self.Hp
            K_l = in_l if l == 0 else 2 * self.Hp
            a = torch.empty(2, self.NP, Kp, device=dev, dtype=torch.float32)
            c = torch.empty(2, self.NP, device=dev, dtype=torch.float32)
            ai = torch.empty(2 * self.NP, (K_l + 31) // 32, 2, 32, device=dev, dtype=torch.int16)
            b3 = torch.empty(2, we3, device=dev, dtype=torch.int16)
            r = torch.empty(2, nR, device=dev, dtype=torch.int16) if nR else None
            for d in range(2):
                w_ih, w_hh, b_ih, b_hh = cont[(2 * l + d) * 4:(2 * l + d) * 4 + 4]
                ptr["w_ih"].append(w_ih.data_ptr()), ptr["w_hh"].append(w_hh.data_ptr())
                ptr["b_ih"].append(b_ih.data_ptr()), ptr["b_hh"].append(b_hh.data_ptr())
                ptr["a"].append(a[d].data_ptr()), ptr["c"].append(c[d].data_ptr()), ptr["ai"].append(ai[d * self.NP:].data_ptr())
                ptr["b3"].append(b3[d].data_ptr())
                if nR:
                    ptr["r"].append(r[d].data_ptr())
            self.wih.append(a), self.whh.append(None), self.bias.append(c), self.whh_x3.append(b3), self.wih_x3.append(None)
            self.wih_img.append(ai)
            whhR.append(r)
        lib.lstm_pack_train(L, p.input_size, H, self.ug, ptr["w_ih"], ptr["w_hh"], ptr["b_ih"], ptr["b_hh"], ptr["a"], ptr["c"], ptr["ai"],
                            ptr["b3"], ptr["r"] if nR else None, st)
        self._keep = cont                  # (contiguous copies, if any were made, must outlive the launch)
        self._whhT = {_abi.LSTM_BWD_XCD: whhR} if nR else None

    def whh_bwd(self, form):
        """Per layer, both directions' images of W_hh for the backward recurrence (training only; built lazily):
        form LSTM_BWD_XCD -> row slices (onssen_lstm_pack_whhR_bf16x3), LSTM_BWD_STEPS -> column slices (whhT)."""
        if self._whhT is None:
            self._whhT = {}
        if form not in self._whhT:
            p, lib = self.p, get_lib()
            flat = p.flat_weights()
            xcd = form == _abi.LSTM_BWD_XCD
            n = (lib.lstm_whhR_elems if xcd else lib.lstm_whhT_elems)(p.hidden_size, self.ug)
            pack = lib.lstm_pack_whhR_bf16x3 if xcd else lib.lstm_pack_whhT_bf16x3
            out = []
            for l in range(p.num_layers):
                img = torch.empty(2, n, device=flat[0].device, dtype=torch.int16)
                for d in range(2):
                    w_hh = flat[(2 * l + d) * 4 + 1].detach().contiguous()
                    pack(w_hh.data_ptr(), p.hidden_size, self.ug, img[d].data_ptr(), _stream())
                out.append(img)
            self._whhT[form] = out
        return self._whhT[form]


class PackedHead:
    """nn.Linear(2H -> N) re-laid for the [fwd(Hp)|rev(Hp)] activations, with
    an optional eval-mode BatchNorm1d(2H) folded in."""

    def __init__(self, linear: nn.Linear, bn, H):
        self.lin, self.bn, self.H = linear, bn, H
        self.key = None
        self.guard = _WeightGuard(f"the {tuple(linear.weight.shape)} head" + (" + BatchNorm" if bn is not None else ""))

    def get(self, Hp):
        lin, bn = self.lin, self.bn
        ts = [lin.weight, lin.bias] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])
        key = (_version_key(ts), Hp)
        capturing = lin.weight.is_cuda and torch.cuda.is_current_stream_capturing()
        if key == self.key:
            self._in_graph = getattr(self, "_in_graph", False) or capturing
            self.guard.check()
            return self
        _park_if_captured(self, ("w", "b", "planes", "img"))
        self._in_graph = capturing
        lib, dev, N = get_lib(), lin.weight.device, lin.weight.shape[0]
        self.w = torch.empty(N, 2 * Hp, device=dev, dtype=torch.float32)
        self.b = torch.empty(N, device=dev, dtype=torch.float32)
        w, b = lin.weight.detach().contiguous(), lin.bias.detach().contiguous()
        if bn is not None:
            g, be, mu, var = [t.detach().contiguous() for t in ts[2:]]
            lib.head_pack(w.data_ptr(), b.data_ptr(), N, self.H, Hp, g.data_ptr(), be.data_ptr(), mu.data_ptr(),
                          var.data_ptr(), float(bn.eps), self.w.data_ptr(), self.b.data_ptr(), _stream())
        else:
            lib.head_pack(w.data_ptr(), b.data_ptr(), N, self.H, Hp, None, None, None, None, 0.0,
                          self.w.data_ptr(), self.b.data_ptr(), _stream())
        self.ld3 = (2 * Hp + 31) // 32 * 32
        self.planes = torch.empty(2, N, self.ld3, device=dev, dtype=torch.int16)
        lib.linear_pack_bf16x3(self.w.data_ptr(), N, 2 * Hp, 2 * Hp, self.ld3, self.planes.data_ptr(), _stream())
        self.img = torch.empty(N, self.ld3 // 32, 2, 32, device=dev, dtype=torch.int16)     # x3 image (onssen_linear_x3p)
        lib.x3_image(self.w.data_ptr(), 2 * Hp, 0, 1, N, 2 * Hp, self.img.data_ptr(), _stream())
        self.N, self.key = N, key
        self.guard.arm(ts)
        return self


class _Workspaces:
    """Caller-owned scratch buffers, cached per shape so that steady-state
    forwards allocate nothing (and stay hipGraph-capturable)."""

    def __init__(self):
        self.cache = {}

    def get(self, key, nbytes, device):
        buf = self.cache.get(key)
        if buf is None or buf.numel() < nbytes or buf.device != device:
            buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)   # the header must start out zero (ABI)
            self.cache[key] = buf
        return buf

    def scratch(self, tag, nbytes, header, device):
        """Grow-only buffer for call shapes that keep changing (ragged batches of whole utterances: a new longest
        utterance per batch): allocated uninitialised, only the first ``header`` bytes zeroed, reused as it is by every later
        call that fits -- a shape-keyed cache would allocate and memset a 100-300 MB workspace per batch.  The callee
        is told that the rest is dirty (ONSSEN_BLSTM_WS_DIRTY).  Not for hipGraph capture (a regrowth would free memory a
        captured graph still points into): captures take the shape-keyed cache."""
        key = ("scratch", tag)
        buf = self.cache.get(key)
        if buf is None or buf.numel() < nbytes or buf.device != device:
            buf = torch.empty(max(nbytes, int(nbytes * 1.25)), dtype=torch.uint8, device=device)
            buf[:header].zero_()
            self.cache[key] = buf
        return buf


def require_device(x, who):
    if not x.is_cuda:
        raise RuntimeError(f"{who}: the HIP inference path needs tensors on a ROCm device (got {x.device}); "
                           "onssen_amd has no CPU fallback")


def heads_take_image(B, H, groups=()):
    """True when this forward's heads can all read the recurrence's x3 output image (XCD form, no residual, L2-norm
    groups of 20 / 40 / 80): the caller may then pass ``need_y=False`` and the fp32 rows are never written."""
    _, flags = recurrence_plan(B, H)
    return (bool(flags & _abi.BLSTM_XCD) and bool(flags & _abi.BLSTM_BF16X3)
            and all(g % 4 == 0 and 80 % g == 0 and 80 // g <= 4 for g in groups))


def as_frames(frames, B, T, device):
    """Per-row frame counts of a ragged batch as an int32 device tensor (validated on the host when they come from it)."""
    if torch.is_tensor(frames) and frames.is_cuda:
        fr = frames.to(device=device, dtype=torch.int32).contiguous()
    else:
        host = torch.as_tensor(frames, dtype=torch.int64).reshape(-1)
        if host.numel() != B or int(host.min()) < 1 or int(host.max()) > T:
            raise ValueError(f"frames: expected {B} values in [1, {T}], got {host.tolist()}")
        fr = host.to(device=device, dtype=torch.int32)
    if fr.numel() != B:
        raise ValueError(f"frames: expected {B} values, got {fr.numel()}")
    return fr


def run_blstm(packed: PackedBLSTM, ws: _Workspaces, x, tag="rnn", need_y=True, frames=None):
    """x (B,T,In) float32 cuda -> y (T,B,2,Hp) time-major (padded units are 0).  With ``need_y=False`` (see
    heads_take_image) only the x3 image attached to the result is valid.

    ``frames`` (int32 device tensor, see as_frames): a RAGGED batch of whole utterances padded to T -- row b is live at
    t < frames[b] and holds h = c = 0 (output rows 0) elsewhere, so that its outputs are bit for bit those of a batch-1
    call on that utterance alone (the reference evaluates one utterance at a time, onssen/utils/test.py:29-41).  Runs the
    ragged instantiation of the persistent recurrence (split-bf16, first layer unfused); exact fp32 takes the
    launch-per-step recurrence; the opt-in bf16 mode has no ragged form."""
    lib = get_lib()
    p = packed.p
    B, T, In = x.shape
    _WeightGuard.tick += 1
    if not torch.cuda.is_current_stream_capturing():
        _XcdStatus.poll()
    if frames is not None and precision() == "bf16":
        raise RuntimeError("ragged batches (frames=...) are not available with ONSSEN_PRECISION=bf16")
    if frames is not None and precision() == "f32":
        with _XcdPolicy.forced_steps():                 # the exact-fp32 persistent kernel has no ragged instantiation
            ug, flags = recurrence_plan(B, p.hidden_size)
    else:
        ug, flags = recurrence_plan(B, p.hidden_size)
    pk = packed.get(ug)
    if In != p.input_size:
        raise RuntimeError(f"input feature size {In} != {p.input_size}")
    if x.stride(2) != 1:
        x = x.contiguous()
    nbytes = lib.blstm_workspace_bytes(B, T, In, p.hidden_size, p.num_layers, pk.ug)
    if frames is not None and not torch.cuda.is_current_stream_capturing():
        wsb = ws.scratch(tag, nbytes, _abi.BLSTM_WS_HEADER, x.device)      # a new longest utterance per batch: no per-shape buffers
        flags |= _abi.BLSTM_WS_DIRTY
    else:
        wsb = ws.get((tag, B, T), nbytes, x.device)
    y = torch.empty(T, B, 2, pk.Hp, device=x.device, dtype=torch.float32)
    images = bool(flags & _abi.BLSTM_XCD) and bool(flags & _abi.BLSTM_BF16X3)     # activations travel as x3 images
    wih = pk.wih_img if images else pk.wih_x3 if flags & _abi.BLSTM_BF16X3 else pk.wih
    bias = pk.bias
    # measured (dc / chimera, H=600; round 3, with the x products formed in the window after the barrier): fused is 1.7 %
    # faster at B=64, 1.0 % at B=32 (it lost 2.4 % there before), 1.6 % SLOWER at B=16 -- the fused MFMAs cost every
    # time step the same, the GEMM (and the 246 MB of G) they replace shrink with the batch
    fuse_env = options.get("fuse_first_layer")
    if frames is None and images and pk.wih_frag0 is not None and (fuse_env == "1" or (fuse_env == "auto" and B > 16)):
        flags |= _abi.BLSTM_FUSE_IN0                     # first layer's x W_ih^T inside its recurrence launch
        wih = [pk.wih_frag0] + list(wih[1:])
        if pk.bias0_tail is not None:
            flags |= _abi.BLSTM_FUSE_TAIL
            bias = [pk.bias0_tail] + list(pk.bias[1:])
    if flags & _abi.BLSTM_XCD:
        _XcdSerial.before(x.device)
    lib.blstm_forward(x.data_ptr(), x.stride(0), x.stride(1), B, T, In, p.hidden_size, p.num_layers, pk.ug,
                      [t.data_ptr() for t in wih],
                      [t.data_ptr() for t in (pk.whh_x3 if flags & _abi.BLSTM_BF16X3 else pk.whh)],
                      [t.data_ptr() for t in bias], y.data_ptr() if need_y or not images else None,
                      wsb.data_ptr(), wsb.numel(), flags, _stream(), frames=frames.data_ptr() if frames is not None else None)
    if persistent_capable(p.hidden_size) and options.get("recurrence") == "1":
        _XcdPolicy.note_launch(bool(flags & _abi.BLSTM_XCD))
    y.x3_image = None
    y.fp32_valid = bool(need_y or not images)
    if images:
        # the last layer's output also sits in the workspace as an x3 image: the heads' GEMM operand
        off, _ = lib.blstm_y_image(B, T, In, p.hidden_size, p.num_layers, pk.ug)
        y.x3_image = (wsb, off)                        # keeps the workspace alive with y
    if flags & _abi.BLSTM_XCD:
        _XcdSerial.after(x.device)
        _XcdStatus.post(wsb)
        if options.get("check") == "1":      # debug / tests: synchronise and examine now
            _XcdStatus.flush()
    return y


def run_head(head: PackedHead, y, B, T, mode, group=0, eps=1e-12, resid=None, b_off=0, b_total=None, resid_mod=None):
    """y (T, Btot, 2, Hp) time-major -> (B, T, N) batch-major with the fused
    epilogue.  ``b_off``/``b_total`` select a batch slice of y (phase net); ``resid_mod`` = R: batch row b adds residual row
    b % R (phase net, both speakers in one launch: ``resid`` is (R, T, N))."""
    lib = get_lib()
    Btot = y.shape[1] if b_total is None else b_total
    Hp = y.shape[3]
    hd = head.get(Hp)
    out = torch.empty(B, T, hd.N, device=y.device, dtype=torch.float32)
    a_ptr = y.data_ptr() + b_off * 2 * Hp * 4
    rp = resid.data_ptr() if resid is not None else None
    img = getattr(y, "x3_image", None)
    if (img is not None and mode == EPI_L2NORM and resid is not None and resid_mod is not None and b_off == 0 and Btot == B
            and (group == 2 or group % 4 == 0) and hd.N % group == 0):
        wsb, off = img                               # pre-split activations, residual + pair / group normalisation in the epilogue
        lib.linear_x3p_resid(wsb.data_ptr() + off, T * B, 2 * Hp, hd.img.data_ptr(), hd.b.data_ptr(), hd.N, group, eps, rp,
                             resid_mod, out.data_ptr(), B, hd.N, T * hd.N, precision() == "bf16", _stream())
        return out
    if resid_mod is not None:
        raise RuntimeError("run_head: resid_mod needs the x3 image of the recurrence output (XCD form)")
    if (img is not None and resid is None and b_off == 0 and Btot == B
            and (mode != EPI_L2NORM or (group % 4 == 0 and 80 % group == 0 and 80 // group <= 4))):
        wsb, off = img                               # pre-split activations straight from the recurrence epilogue
        if precision() == "bf16":
            mode |= _abi.EPI_BF16
        lib.linear_x3p(wsb.data_ptr() + off, T * B, 2 * Hp, hd.img.data_ptr(), hd.b.data_ptr(), hd.N, mode, group, eps,
                       out.data_ptr(), B, hd.N, T * hd.N, _stream())
    elif not getattr(y, "fp32_valid", True):
        raise RuntimeError("run_head: this head cannot read the x3 image but the recurrence was run with need_y=False")
    elif _split_bf16() and (mode != EPI_L2NORM or 160 % group == 0):
        lib.linear_bf16x3(a_ptr, Btot * 2 * Hp, 2 * Hp, B, T * B, 2 * Hp, hd.planes.data_ptr(), hd.ld3, hd.b.data_ptr(),
                          hd.N, mode, group, eps, rp, out.data_ptr(), hd.N, T * hd.N, _stream())
    else:
        lib.linear(a_ptr, Btot * 2 * Hp, 2 * Hp, B, T * B, 2 * Hp, hd.w.data_ptr(), 2 * Hp, hd.b.data_ptr(), hd.N,
                   mode, group, eps, rp, out.data_ptr(), hd.N, T * hd.N, _stream())
    return out


def run_head_pair(head_a: PackedHead, head_b: PackedHead, y, B, T, group, eps=1e-12):
    """Two heads over the same recurrence output in ONE launch (chimera: fc_dc + L2 norm over ``group`` | fc_mi + sigmoid):
    returns ((B, T, Na), (B, T, Nb)), or None when this forward cannot take the pre-split-operand GEMM (no x3 image, a group
    the register epilogue does not hold) -- the caller then runs the heads one by one."""
    img = getattr(y, "x3_image", None)
    if img is None or not (group % 4 == 0 and 80 % group == 0 and 80 // group <= 4):
        return None
    lib = get_lib()
    Hp = y.shape[3]
    ha, hb = head_a.get(Hp), head_b.get(Hp)
    if ha.N % group != 0 or y.shape[1] != B:
        return None
    key = (ha.key, hb.key)
    if getattr(head_a, "_pair_key", None) != key:
        head_a._pair_img = torch.cat([ha.img, hb.img], 0).contiguous()          # rows of both layers: one B operand
        head_a._pair_b = torch.cat([ha.b, hb.b]).contiguous()
        head_a._pair_key = key
    out_a = torch.empty(B, T, ha.N, device=y.device, dtype=torch.float32)
    out_b = torch.empty(B, T, hb.N, device=y.device, dtype=torch.float32)
    wsb, off = img
    lib.linear_x3p_pair(wsb.data_ptr() + off, T * B, 2 * Hp, head_a._pair_img.data_ptr(), head_a._pair_b.data_ptr(), ha.N + hb.N,
                        ha.N, group, eps, out_a.data_ptr(), B, ha.N, T * ha.N, out_b.data_ptr(), hb.N, T * hb.N,
                        precision() == "bf16", _stream())
    return out_a, out_b


def use_hip_path(module):
    """Inference (eval mode, no autograd graph needed) -> HIP kernels.
    Anything that needs autograd or train-mode BatchNorm/dropout takes the
    training path (HIP forward/backward of the BLSTM stack and the heads, nn/_train.py; the rest on ATen autograd)."""
    if module.training:
        return False
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        return False
    return True


def needs_graph(*inputs):
    """True when autograd is on and an INPUT wants a gradient (frozen parameters, e.g. input saliency): the HIP inference
    path builds no graph, so such a forward must take the training path too."""
    return torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in inputs)


EPI_BIAS, EPI_L2NORM, EPI_SIGMOID, EPI_RELU = _abi.EPI_BIAS, _abi.EPI_L2NORM, _abi.EPI_SIGMOID, _abi.EPI_RELU
