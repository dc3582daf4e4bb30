"""One table of every run-time switch of the package.

The reference's recipes are driven by a config JSON (egs/wsj0-2mix/deep_clustering/config.json:1-27; SURVEY section 5 asks
for optional keys such as ``precision`` / ``world_size`` on that surface).  Every switch below can be given

  * in a recipe config: an optional ``"hip_options": {...}`` object at the top level, or the same keys inside
    ``"model_options"`` (the model constructors accept and strip them) -- ``apply_config(args)`` reads both;
  * in code: ``options.configure(precision="f32", recurrence="steps")``;
  * in the environment (``ONSSEN_*``): an environment variable that is SET always wins -- it is the operator's override for
    one run, and what the tools and tests use.

A reference config without any of these keys loads unchanged (tests/test_config_surface.py).  The settings are per process
(one process drives one GPU), not per model.  Debug-build knobs of the native library (``ONSSEN_KNOB_INT`` in
csrc/onssen_hip.hip; compiled out of the shipped build) are not listed here.
"""
import os

_BOOL = {True: "1", False: "0", "1": "1", "0": "0", "true": "1", "false": "0", "on": "1", "off": "0"}


def _choice(*allowed):
    def conv(v):
        v = str(v).lower() if not isinstance(v, bool) else _BOOL[v]
        if v not in allowed:
            raise ValueError(f"must be one of {allowed}, got {v!r}")
        return v
    return conv


def _flag(v):
    try:
        return _BOOL[v if isinstance(v, bool) else str(v).lower()]
    except KeyError:
        raise ValueError(f"must be a boolean, got {v!r}") from None


def _alias(table):
    """Config spelling -> the value the code compares against (``"persistent"`` -> ``"1"``)."""
    def conv(v):
        k = _BOOL.get(v, None) if isinstance(v, bool) else str(v).lower()
        if k in table:
            return table[k]
        if k in table.values():
            return k
        raise ValueError(f"must be one of {sorted(table)}, got {v!r}")
    return conv


def _int(v):
    return str(int(v))


# key: (environment variable, default, converter, one line of documentation)
TABLE = {
    "precision": ("ONSSEN_PRECISION", "bf16x3", _choice("f32", "bf16x3", "bf16"),
                  "arithmetic of the BLSTM / head contractions: f32 = exact-fp32 MFMA, bf16x3 = fp32 as three bf16 MFMAs (default, "
                  "inside the 1e-4 contract), bf16 = plain bf16 products (opt-in, outside it)"),
    "recurrence": ("ONSSEN_XCD", "1", _alias({"persistent": "1", "steps": "0"}),
                   "persistent = one XCD-local launch per layer; steps = one launch per time step"),
    "fuse_first_layer": ("ONSSEN_FUSE_IN0", "auto", _alias({"auto": "auto", "1": "1", "0": "0"}),
                         "first layer's input projection inside the recurrence launch (auto: batches above 16 rows)"),
    "recurrence_unit_group": ("ONSSEN_XCD_UG", "0", _int,
                              "hidden units per member of the persistent recurrence (0 = 4*ceil(H/128); 24 at H <= 640 is the "
                              "25-member A/B form of round 5)"),
    "step_unit_group": ("ONSSEN_UG", "8", _int, "hidden units per workgroup of the launch-per-step recurrence"),
    "split_rows": ("ONSSEN_SPLIT_ROWS", "0", _flag, "launch-per-step recurrence: split the batch rows over two workgroups"),
    "ablate": ("ONSSEN_ABLATE", "0", _int, "profiling-only ablation bits of the recurrence kernels"),
    "serialize_persistent": ("ONSSEN_XCD_SERIALIZE", "0", _flag, "one persistent launch in flight per device across streams"),
    "nonfinite": ("ONSSEN_NONFINITE", "propagate", _choice("raise", "propagate"),
                  "non-finite activations seen by a persistent launch (they cannot pass its tagged exchange): propagate (default since round 6) = the "
                  "call is re-run on the launch-per-step recurrence, which gives nn.LSTM's NaN in / NaN out -- the reference's behaviour --; "
                  "raise = stop with an error instead"),
    "check": ("ONSSEN_CHECK", "0", _flag, "synchronise and examine the persistent launches' status words after every forward"),
    "check_weights": ("ONSSEN_CHECK_WEIGHTS", "0", _flag, "verify the packed weight images against the parameters before every forward (one synchronisation each: debugging)"),
    "weight_guard": ("ONSSEN_WEIGHT_GUARD", "1", _flag, "every inference forward that reuses cached weight images compares a device-side sampled checksum of the "
                     "live parameters with the one taken when the images were built (one small launch, no synchronisation); a mismatch drops the "
                     "images and raises StalePackedWeights at the next status poll (separate_* / tester.eval re-run the call by themselves)"),
    "dc_cluster": ("ONSSEN_DC_PERSISTENT", "1", _alias({"persistent": "1", "steps": "0"}),
                   "deep-clustering 2-means: all Lloyd passes in one launch, or one launch per pass"),
    "dc_compact": ("ONSSEN_DC_COMPACT", "1", _flag, "fc_dc stores only the active bins' embeddings into the clustering's array"),
    "train_blstm": ("ONSSEN_TRAIN_HIP", "1", _alias({"hip": "1", "aten": "0"}), "training BLSTM / head / BatchNorm on the HIP kernels"),
    "train_backward": ("ONSSEN_BWD_XCD", "1", _alias({"persistent": "1", "steps": "0"}), "backward recurrence form"),
    "train_gemm": ("ONSSEN_TRAIN_GEMM", "x3", _choice("x3", "blas"), "weight / input gradient contractions: package GEMM or library fp32"),
    "train_dp_image": ("ONSSEN_TRAIN_DP_IMAGE", "1", _flag, "the backward recurrence writes dP as the gradient GEMMs' x3 image instead of fp32"),
    "train_wgrad_rows": ("ONSSEN_TRAIN_WGRAD_ROWS", "1", _flag, "weight gradients from the row-major images the forward left"),
    "train_fused_loss": ("ONSSEN_TRAIN_FUSED_LOSS", "1", _flag, "fc_dc + normalise + loss_dc as one autograd node in train_step"),
    "train_fused_norm": ("ONSSEN_TRAIN_FUSED_NORM", "1", _flag, "fc_dc + normalise as one autograd node"),
    "loss": ("ONSSEN_LOSS_HIP", "1", _alias({"hip": "1", "torch": "0"}), "loss kernels on the device or PyTorch ops"),
    "fused_adam": ("ONSSEN_FUSED_ADAM", "1", _flag, "build_optimizer returns utils.ClipAdam (clipping + Adam on onssen_clip_adam_f32) for device parameters"),
    "cpu_autograd": ("ONSSEN_CPU_AUTOGRAD", "0", _flag, "TEST SCAFFOLDING, off in the product: let a training forward on CPU tensors run on ATen's LSTM so that "
                     "the multi-process logic (gradient buckets, replica agreement) can be exercised over gloo without a GPU; otherwise a CPU tensor raises"),
    "synthetic_data": ("ONSSEN_SYNTHETIC_DATA", "0", _flag, "wsj0_2mix_dataloader falls back to the synthetic corpus when data_path is empty"),
    "loader_workers": ("ONSSEN_LOADER_WORKERS", "4", _int, "file-reading threads of the wsj0-2mix file loader (0 = read on the calling thread)"),
    "loader_prefetch": ("ONSSEN_LOADER_PREFETCH", "3", _int, "batches the file loader keeps in flight ahead of the training step"),
    "world_size": ("WORLD_SIZE", "1", _int, "data-parallel ranks (torch.distributed.run exports it; a config may state it for checking)"),
}

_configured = {}


def get(key):
    """Current value of ``key`` as a string: environment variable if set, else ``configure``d value, else the default."""
    env, default, _, _ = TABLE[key]
    v = os.environ.get(env)
    if v is not None:
        return v
    return _configured.get(key, default)


def flag(key):
    return get(key) == "1"


def configure(**kw):
    """Set switches for this process (validated).  ``None`` removes a setting.  Returns the previous values."""
    old = {}
    for k, v in kw.items():
        if k not in TABLE:
            raise KeyError(f"onssen_amd.options: unknown option {k!r} (known: {sorted(TABLE)})")
        old[k] = _configured.get(k)
        if v is None:
            _configured.pop(k, None)
            continue
        try:
            _configured[k] = TABLE[k][2](v)
        except ValueError as e:
            raise ValueError(f"onssen_amd.options: {k} {e}") from None
    return old


def constructor_options(cls_name, kw):
    """``**hip_options`` of a model constructor: TABLE keys are applied, anything else is the TypeError Python would raise."""
    bad = [k for k in kw if k not in TABLE]
    if bad:
        raise TypeError(f"{cls_name}.__init__() got an unexpected keyword argument {bad[0]!r}")
    configure(**kw)


def split_model_options(model_options):
    """(constructor kwargs, hip options) of a recipe's ``model_options``: the keys of TABLE are taken out so that
    ``onn.deep_clustering(**kwargs)`` keeps the reference's signature (onssen/nn/deep_clustering.py:6-13)."""
    kwargs = {k: v for k, v in dict(model_options).items() if k not in TABLE}
    return kwargs, {k: v for k, v in dict(model_options).items() if k in TABLE}


def apply_config(args):
    """Apply the optional switches of a recipe config (a dict / AttrDict as ``json.load`` gives it): top-level
    ``hip_options`` and any TABLE key found in ``model_options`` / ``feature_options``.  Returns ``model_options`` without
    them.  A stated ``world_size`` that disagrees with the launcher's is an error (the reference has no such key: silent
    single-GPU training of an 8-GPU recipe is the failure this catches)."""
    g = args.get if hasattr(args, "get") else (lambda k, d=None: getattr(args, k, d))
    hip = dict(g("hip_options", None) or {})
    kwargs, extra = split_model_options(g("model_options", None) or {})
    hip.update(extra)
    hip.update(split_model_options(g("feature_options", None) or {})[1])
    ws = hip.pop("world_size", None)
    if ws is not None and int(ws) != int(os.environ.get("WORLD_SIZE", "1")):
        raise RuntimeError(f"config states world_size = {ws} but this process was launched with WORLD_SIZE = "
                           f"{os.environ.get('WORLD_SIZE', '1')} (python -m torch.distributed.run --nproc-per-node {ws} ...)")
    configure(**hip)
    return kwargs


def describe():
    """{key: {env, value, default, source, doc}} -- what a run actually used (bench.py prints the non-defaults)."""
    out = {}
    for k, (env, default, _, doc) in TABLE.items():
        src = "env" if os.environ.get(env) is not None else "config" if k in _configured else "default"
        out[k] = {"env": env, "value": get(k), "default": default, "source": src, "doc": doc}
    return out
