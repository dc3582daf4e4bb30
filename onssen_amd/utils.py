"""The small pieces of onssen.utils that the recipes' run.py files touch next to the hot path (SURVEY 5 "Config" row):
the config container and the optimizer factory.  The trainer / tester LOOPS stay the reference's own host code (they are
out of scope: control flow around the path); ``onssen_amd.dist.train_step`` and ``onssen_amd.evaluate.tester`` are the
counterparts of their bodies."""
import os

import torch

from . import options


class AttrDict(dict):
    """Dictionary whose keys are also attributes, nested -- what ``attrdict.AttrDict`` gives the reference's run.py
    (egs/wsj0-2mix/deep_clustering/run.py:19-23: ``args = AttrDict(json.load(f))``, then both ``args['model_options']`` and
    ``args.feature_options.batch_size``).  The ``attrdict`` package is dead on Python >= 3.10 (collections.Mapping)."""

    def __getattr__(self, name):
        try:
            v = self[name]
        except KeyError:
            raise AttributeError(name) from None
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        try:
            del self[name]
        except KeyError:
            raise AttributeError(name) from None


def build_optimizer(params, optimizer_options):
    """onssen/utils/basic.py:5-11: ``{"name": "adam" | "sgd" | "rmsprop", "lr": ...}``.  Every optimizer returned drops the
    package's packed weight images after ``step()`` (a fused step moves the parameters without bumping their versions:
    ``nn._core.invalidate_packed_weights``)."""
    name, lr = optimizer_options["name"], optimizer_options["lr"]
    if name == "adam":
        # same update rule; on a GPU the whole step is ONE multi-tensor kernel instead of ~10 (ONSSEN_FUSED_ADAM=0: torch's default)
        params = list(params)
        fused = bool(params) and all(p.is_cuda for p in params) and options.get("fused_adam") == "1"
        opt = torch.optim.Adam(params, lr=lr, fused=True) if fused else torch.optim.Adam(params, lr=lr)
    elif name == "sgd":
        opt = torch.optim.SGD(params, lr=lr, momentum=0.9)
    elif name == "rmsprop":
        opt = torch.optim.RMSprop(params, lr=lr)
    else:
        raise ValueError(f"unknown optimizer {name!r}")
    from .nn._core import invalidate_packed_weights
    opt.register_step_post_hook(lambda optimizer, args, kwargs: invalidate_packed_weights())
    return opt
