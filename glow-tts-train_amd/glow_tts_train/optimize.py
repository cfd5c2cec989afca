"""Adam with the Noam schedule behind the reference's `optimize.Adam` surface (reference: glow_tts_train/optimize.py),
restructured for MI355X: every parameter, gradient and moment lives in ONE flat fp32 buffer each, so

  * `clip_grad_value_` is one streaming kernel instead of 519 `.item()` host syncs (reference utils.py:126-128),
  * the update is one streaming kernel over 4 x 114.5 MB (12 blocks) instead of ~519 x 4 small launches,
  * gradient buckets for data-parallel all-reduce are plain slices of the flat gradient buffer (parallel.py),
  * the learning rate is derived on device from a step counter, so a captured hipGraph replays correctly.
"""
from __future__ import annotations

import typing
import weakref
from operator import is_ as _is

import numpy as np
import torch

from ._hip import call, ptr

_ALIGN = 64  # elements: every parameter starts on a 256-byte boundary of the flat buffers


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam arithmetic (no amsgrad / weight decay) over flat buffers, launched through the C ABI."""

    def __init__(self, params, lr=1.0, betas=(0.9, 0.98), eps=1e-9, dim_model: float = 0.0, warmup_steps: float = 0.0,
                 base_lr: typing.Optional[float] = None):
        params = [p for p in params]
        # the group carries every key torch.optim.Adam's does (amsgrad, weight_decay, foreach, ... at their defaults), so
        # state_dict() has torch's layout for this torch version and the reference loads it unchanged
        template = torch.optim.Adam([torch.zeros(1)], lr=lr, betas=betas, eps=eps).param_groups[0]
        super().__init__(params, {k: v for k, v in template.items() if k != "params"})
        self.base_lr = float(lr if base_lr is None else base_lr)
        self.dim_model, self.warmup = float(dim_model), float(warmup_steps)
        self._build_flat()

    # -- layout -------------------------------------------------------------------------------------------------
    def _build_flat(self):
        ps = [p for g in self.param_groups for p in g["params"]]
        if not ps:
            raise ValueError("FlatAdam: no parameters")
        dev, dt = ps[0].device, torch.float32
        offs, total = [], 0
        for p in ps:
            if p.dtype != dt or p.device != dev:
                raise RuntimeError("FlatAdam: all parameters must be fp32 on one device")
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.offsets, self.numel_padded = offs, total
        self.numel = sum(p.numel() for p in ps)
        self.flat_p = torch.zeros(total, device=dev, dtype=dt)
        self.flat_g = torch.zeros(total, device=dev, dtype=dt)
        self.flat_m = torch.zeros(total, device=dev, dtype=dt)
        self.flat_v = torch.zeros(total, device=dev, dtype=dt)
        # device state: [adam step t (1-based for the NEXT update), noam step_num, lr of the next update,
        #                lr imposed on the next update only (0 = follow the schedule)]
        self.dev_state = torch.tensor([1.0, 1.0, 0.0, 0.0], device=dev, dtype=dt)
        with torch.no_grad():
            for p, o in zip(ps, offs):
                n = p.numel()
                self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                if p.grad is not None:
                    self.flat_g[o:o + n].copy_(p.grad.reshape(-1))
                p.grad = self.flat_g[o:o + n].view(p.shape)
                p._glowtts_flat_grad = self.flat_g
                p._glowtts_flat_numel = self.numel
                p._glowtts_flat_owner = weakref.ref(self)
        self._params = ps
        # zero_grad() has not run yet: the views just installed are the ones grads_in_place() compares against
        self._views = [p.grad for p in ps]

    def slices(self):
        """(offset, numel) of every parameter inside the flat buffers, in construction order."""
        return [(o, p.numel()) for p, o in zip(self._params, self.offsets)]

    # -- torch.optim surface ------------------------------------------------------------------------------------
    def _grad_views(self):
        """The gradient views handed out by zero_grad, kept so that "is .grad still ours?" is an identity test per
        parameter (this runs three times per step over ~500 parameters: pointer arithmetic there cost milliseconds)."""
        views = getattr(self, "_views", None)
        if views is None:
            views = self._views = [None] * len(self._params)
        return views

    def zero_grad(self, set_to_none: bool = False):
        # the gradients must stay views of the flat buffer: one memset, never `grad = None`
        self.flat_g.zero_()
        views = self._grad_views()
        for i, p in enumerate(self._params):
            if p.grad is not views[i] or views[i] is None:
                o = self.offsets[i]
                views[i] = self.flat_g[o:o + p.numel()].view(p.shape)
                p.grad = views[i]

    def grads_in_place(self) -> bool:
        views = self._grad_views()
        return all(map(_is, [p.grad for p in self._params], views))        # (C-level loops: called twice per step)

    def clip_grad_value_(self, clip_value: float):
        """utils.clip_grad_value_ over the whole flat gradient buffer in one launch; None if a gradient has been replaced by
        a foreign tensor (the caller then takes the general path)."""
        if not self.grads_in_place():
            return None
        from .utils import _FlatGradView
        sumsq = torch.zeros(1, device=self.flat_g.device, dtype=torch.float32)
        call("glowtts_clip_grad_value", ptr(self.flat_g), self.flat_g.numel(), float(clip_value), ptr(sumsq))
        return _FlatGradView(sumsq, 2.0)

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("FlatAdam.step: closures are not supported")
        if not self.grads_in_place():                 # a foreign hook may have replaced .grad: fold it back in
            views = self._grad_views()
            for i, (p, o) in enumerate(zip(self._params, self.offsets)):
                if p.grad is not None and p.grad is not views[i] and p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                    self.flat_g[o:o + p.numel()].copy_(p.grad.reshape(-1))
                if p.grad is not views[i]:
                    views[i] = self.flat_g[o:o + p.numel()].view(p.shape)
                    p.grad = views[i]
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        call("glowtts_adam_noam", ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_m), ptr(self.flat_v),
             self.numel_padded, ptr(self.dev_state), self.base_lr, float(b1), float(b2), float(g["eps"]),
             self.dim_model, self.warmup)
        call("glowtts_adam_advance", ptr(self.dev_state), self.base_lr, self.dim_model, self.warmup)

    def state_dict(self):
        """torch.optim.Adam-compatible layout: state[i] = {step, exp_avg, exp_avg_sq} (what checkpoint.py:44 saves)."""
        t = (self.dev_state[0] - 1.0).detach().cpu()        # ONE device read for the step every entry shares
        state = {}
        for i, (p, o) in enumerate(zip(self._params, self.offsets)):
            n = p.numel()
            state[i] = {
                "step": t.clone(),
                "exp_avg": self.flat_m[o:o + n].view(p.shape).clone(),
                "exp_avg_sq": self.flat_v[o:o + n].view(p.shape).clone(),
            }
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        groups[0]["params"] = list(range(len(self._params)))
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, d):
        """Like torch.optim.Optimizer.load_state_dict: a state whose parameter count or moment shapes differ from this
        optimizer's raises ValueError; an EMPTY state (a file saved before the first update) is accepted, and so is a
        state without an entry for some parameter — torch.optim.Adam only creates a parameter's state at its first
        gradient, so a reference checkpoint of a model with a frozen / never-used parameter has none: zero moments here,
        with a warning."""
        st = d.get("state", {})
        saved_groups = d.get("param_groups", [])
        if saved_groups and "params" in saved_groups[0]:
            n_saved = sum(len(g["params"]) for g in saved_groups)
            if n_saved != len(self._params):
                raise ValueError(f"loaded state dict has {n_saved} parameters, the optimizer has {len(self._params)}")
        missing = [i for i in range(len(self._params)) if st and i not in st and str(i) not in st]
        if missing:
            import warnings
            warnings.warn(f"FlatAdam.load_state_dict: no state for parameters {missing[:8]}{'...' if len(missing) > 8 else ''} "
                          "(never updated when the file was written): their moments start at zero.  All parameters share ONE "
                          "device step counter here, so these parameters' bias correction continues at the file's step N — "
                          "torch.optim.Adam would restart theirs at step 1 (their first updates are ~1 / (1 - beta1^N) times "
                          "smaller than the reference's)")
        steps = []
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self._params, self.offsets)):
                s = st.get(i, st.get(str(i)))
                n = p.numel()
                if s is None:
                    if st:
                        self.flat_m[o:o + n].zero_()
                        self.flat_v[o:o + n].zero_()
                    continue
                for name in ("exp_avg", "exp_avg_sq"):
                    if tuple(s[name].shape) != tuple(p.shape):
                        raise ValueError(f"optimizer state {name}[{i}] has shape {tuple(s[name].shape)}, "
                                         f"the parameter has {tuple(p.shape)}")
                self.flat_m[o:o + n].copy_(s["exp_avg"].reshape(-1))
                self.flat_v[o:o + n].copy_(s["exp_avg_sq"].reshape(-1))
                steps.append(float(s["step"]))
            if steps:
                if max(steps) != min(steps):
                    import warnings
                    warnings.warn("FlatAdam.load_state_dict: per-parameter steps differ; using the largest")
                self.dev_state[0] = max(steps) + 1.0
        for g_new, g in zip(d.get("param_groups", []), self.param_groups):
            for k in ("lr", "betas", "eps"):
                if k in g_new:
                    g[k] = g_new[k]
        # torch.optim.Adam.load_state_dict leaves the stored group lr in force until the schedule next writes it, i.e.
        # for exactly one update under "noam" (reference optimize.py:43-48, :60-61) and for good otherwise
        groups = d.get("param_groups", [])
        if groups and "lr" in groups[0]:
            lr = float(groups[0]["lr"])
            if self.warmup > 0.0:
                self.dev_state[3] = lr
            else:
                self.base_lr = lr


class Adam:
    """Reference surface (optimize.py:8-64): `Adam(params, scheduler, dim_model, warmup_steps, lr, betas, eps)` with
    `.step() .zero_grad() .get_lr() .state_dict() .load_state_dict() .cur_lr .step_num ._optim`."""

    def __init__(self, params, scheduler, dim_model, warmup_steps: int = 4000, lr: float = 1e0,
                 betas: typing.Tuple[float, float] = (0.9, 0.98), eps: float = 1e-9):
        self.params = list(params)
        self.scheduler, self.dim_model, self.warmup_steps = scheduler, dim_model, warmup_steps
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_num = 1
        self.cur_lr = lr * self._get_lr_scale()
        noam = scheduler == "noam"
        self._optim = FlatAdam(self.params, lr=self.cur_lr, betas=betas, eps=eps, base_lr=lr,
                               dim_model=dim_model if noam else 0.0, warmup_steps=warmup_steps if noam else 0.0)

    def _get_lr_scale(self):
        if self.scheduler == "noam":
            return np.power(self.dim_model, -0.5) * np.min(
                [np.power(self.step_num, -0.5), self.step_num * np.power(self.warmup_steps, -1.5)])
        return 1

    def _update_learning_rate(self):
        # host mirror of what glowtts_adam_advance did on device (no synchronisation)
        self.step_num += 1
        if self.scheduler == "noam":
            self.cur_lr = self.lr * self._get_lr_scale()
            for group in self._optim.param_groups:
                group["lr"] = self.cur_lr

    def get_lr(self):
        return self.cur_lr

    def step(self):
        self._optim.step()
        self._update_learning_rate()

    def zero_grad(self):
        self._optim.zero_grad()

    def load_state_dict(self, d):
        self._optim.load_state_dict(d)

    def state_dict(self):
        return self._optim.state_dict()


OptimizerType = Adam
