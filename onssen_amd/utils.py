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


class ClipAdam(torch.optim.Optimizer):
    """Adam (torch.optim.Adam's update, no weight decay / amsgrad) whose step can take the gradient-norm clipping that
    precedes it in the reference's loop (onssen/utils/train.py:83-84) into the same two passes over the parameters:
    ``step_clipped(max_norm)`` = ``clip_grad_norm_(params, max_norm); step()`` on ``onssen_clip_adam_f32`` -- no scaling pass
    over the gradients, every parameter and moment read once and written once.  ``step()`` is the plain update on the same
    kernel.  CUDA float32 parameters only; the state (``step``, ``exp_avg``, ``exp_avg_sq``) has torch.optim.Adam's layout, so
    state_dicts move between the two.  After ``step_clipped`` ``p.grad`` still holds the UNclipped gradients unless
    ``write_clipped_grads=True``."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, write_clipped_grads=False, weight_decay=0, amsgrad=False,
                 maximize=False):
        if weight_decay != 0 or amsgrad or maximize:
            raise ValueError("ClipAdam implements plain Adam only: weight_decay / amsgrad / maximize are not supported "
                             "(use torch.optim.Adam: fused_adam = 0)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False))
        self.write_clipped_grads = write_clipped_grads
        self._ws = None
        for group in self.param_groups:
            for p in group["params"]:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise TypeError("ClipAdam: contiguous float32 parameters on a ROCm device only")

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._update(0.0)
        return loss

    @torch.no_grad()
    def step_clipped(self, max_norm):
        """Clip the global gradient norm to ``max_norm`` and take the Adam step.  Returns the norm before clipping (a device
        scalar, like ``clip_grad_norm_``)."""
        return self._update(float(max_norm))

    def load_state_dict(self, state_dict):
        for g in state_dict.get("param_groups", []):     # a torch.optim.Adam state with options this kernel does not implement
            if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
                raise ValueError("ClipAdam.load_state_dict: weight_decay / amsgrad / maximize are not supported")
        super().load_state_dict(state_dict)

    def _update(self, max_norm):
        try:
            return self._update_impl(max_norm)
        finally:
            # the kernel moves the parameters through raw pointers: tensor._version does not change, and step_clipped() does not
            # pass through Optimizer.step's hook wrapper -- drop the packed weight images here, whoever built this optimizer
            from .nn._core import invalidate_packed_weights
            invalidate_packed_weights()
            self._opt_called = True              # LR schedulers check that an optimizer step came before scheduler.step()

    def _update_impl(self, max_norm):
        from .hip import get_lib
        lib = get_lib()
        groups = []
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise ValueError("ClipAdam: weight_decay / amsgrad / maximize are not supported")
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            # one kernel call takes ONE bias-correction step: parameters are bucketed by their own step count (torch.optim.Adam
            # keeps it per parameter; a head that only gets a gradient now and then has a smaller one than the rest)
            by_step = {}
            for p in live:
                st = self.state[p]
                by_step.setdefault(int(st["step"]) if st else 0, []).append(p)
            for _, ps in sorted(by_step.items()):
                groups.append((group, ps))
        for group, ps in groups:
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
                st["step"] += 1
                if not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                    p.grad = p.grad.float().contiguous()
        if not groups:
            return None
        # one call per parameter group (its own lr / betas / step); the clip coefficient is the GLOBAL norm's: with several
        # groups the norm pass runs over all of them first
        dev = groups[0][1][0].device
        stream = torch.cuda.current_stream(dev).cuda_stream
        allp = [p for _, ps in groups for p in ps]
        numel = [p.numel() for p in allp]
        norm = None
        if max_norm > 0.0 and max_norm != float("inf"):
            nbytes = lib.clip_adam_workspace_bytes(numel)
            if self._ws is None or self._ws.numel() * 4 < nbytes or self._ws.device != dev:
                self._ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        if len(groups) == 1:
            group, ps = groups[0]
            b1, b2 = group["betas"]
            lib.clip_adam([p.data_ptr() for p in ps], [p.grad.data_ptr() for p in ps], [self.state[p]["exp_avg"].data_ptr() for p in ps],
                          [self.state[p]["exp_avg_sq"].data_ptr() for p in ps], numel, max_norm, group["lr"], b1, b2, group["eps"],
                          int(self.state[ps[0]]["step"]), self._ws.data_ptr() if self._ws is not None else None,
                          self._ws.numel() * 4 if self._ws is not None else 0, stream, write_grads=self.write_clipped_grads)
            if max_norm > 0.0 and max_norm != float("inf"):
                norm = self._ws[0]
        else:      # rare: clip with torch (global norm over all groups), then the plain fused update per group
            if max_norm > 0.0 and max_norm != float("inf"):
                norm = torch.nn.utils.clip_grad_norm_(allp, max_norm)
            for group, ps in groups:
                b1, b2 = group["betas"]
                lib.clip_adam([p.data_ptr() for p in ps], [p.grad.data_ptr() for p in ps], [self.state[p]["exp_avg"].data_ptr() for p in ps],
                              [self.state[p]["exp_avg_sq"].data_ptr() for p in ps], [p.numel() for p in ps], 0.0, group["lr"], b1, b2,
                              group["eps"], int(self.state[ps[0]]["step"]), None, 0, stream)
        return norm


def build_optimizer(params, optimizer_options):
    """onssen/utils/basic.py:5-11: ``{"name": "adam" | "sgd" | "rmsprop", "lr": ...}``.  Every optimizer returned drops the
    package's packed weight images after ``step()`` (a fused step moves the parameters without bumping their versions:
    ``nn._core.invalidate_packed_weights``)."""
    name, lr = optimizer_options["name"], optimizer_options["lr"]
    if name == "adam":
        # same update rule; on a GPU the whole step is ONE multi-tensor kernel instead of ~10 (ONSSEN_FUSED_ADAM=0: torch's default)
        params = list(params)
        # on a GPU: clipping + Adam in two passes on the package's kernel (ClipAdam; fused_adam / ONSSEN_FUSED_ADAM=0: torch's default)
        fused = (bool(params) and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params)
                 and options.get("fused_adam") == "1")
        opt = ClipAdam(params, lr=lr) if fused else torch.optim.Adam(params, lr=lr)
    elif name == "sgd":
        opt = torch.optim.SGD(params, lr=lr, momentum=0.9)
    elif name == "rmsprop":
        opt = torch.optim.RMSprop(params, lr=lr)
    else:
        raise ValueError(f"unknown optimizer {name!r}")
    from .nn._core import invalidate_packed_weights
    opt.register_step_post_hook(lambda optimizer, args, kwargs: invalidate_packed_weights())
    return opt
