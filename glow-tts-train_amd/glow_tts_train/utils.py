"""Losses, layout helpers and gradient clipping with the reference's names and signatures
(reference: glow_tts_train/utils.py), running on the HIP kernels.
"""
from __future__ import annotations

import torch
from torch.nn import functional as F

from . import ops
from ._hip import call, ptr


def intersperse(lst, item):
    out = [item] * (2 * len(lst) + 1)
    out[1::2] = lst
    return out


def mle_loss(z, m, logs, logdet, mask):
    """Negative log-likelihood per (frame, channel) incl. the log-Jacobian (reference utils.py:14-23): one streaming
    reduction kernel forward, one elementwise kernel backward."""
    return ops.MleLossFn.apply(z, m, logs, logdet, ops.mask2d(mask))


def duration_loss(logw, logw_, lengths):
    """reference utils.py:26-28 — (B, 1, T_text) tensors; on the GPU one small kernel each way (ops.DurationLossFn)."""
    if logw.is_cuda and logw.shape == logw_.shape:
        return ops.DurationLossFn.apply(logw, logw_, lengths)
    return torch.sum((logw - logw_) ** 2) / torch.sum(lengths)


def fused_add_tanh_sigmoid_multiply(input_a, input_b, n_channels=None):
    """reference utils.py:31-38.  `input_b` is the conditioning: (B, 2H, 1), broadcastable zeros, or None."""
    if input_b is not None and input_b.shape[-1] != 1:
        input_a = input_a + input_b            # general (B, 2H, T) second operand: add first, gate with no bias
        input_b = None
    return ops.GateFn.apply(input_a, input_b)


def convert_pad_shape(pad_shape):
    return [v for pair in reversed(pad_shape) for v in pair]


def shift_1d(x):
    return F.pad(x, (1, 0))[:, :, :-1]


def sequence_mask(length, max_length=None):
    if max_length is None:
        max_length = length.max()
    return torch.arange(max_length, dtype=length.dtype, device=length.device)[None, :] < length[:, None]


def generate_path(duration, mask):
    """Durations -> hard monotonic alignment (reference utils.py:99-115); inference only."""
    b, t_x, t_y = mask.shape
    ends = torch.cumsum(duration, 1)                                     # [b, t_x]
    frames = torch.arange(t_y, device=duration.device, dtype=ends.dtype)
    upto = (frames[None, None, :] < ends[:, :, None]).to(mask.dtype)    # frame < end of token
    return (upto - F.pad(upto, (0, 0, 1, 0))[:, :-1]) * mask


class _FlatGradView:
    """Lazily-evaluated global gradient norm: float() forces the one host sync the reference pays per TENSOR."""

    def __init__(self, sumsq: torch.Tensor, norm_type: float):
        self._sumsq, self._p = sumsq, norm_type

    def tensor(self) -> torch.Tensor:
        return self._sumsq.sqrt() if self._p == 2.0 else self._sumsq ** (1.0 / self._p)

    def __float__(self):
        return float(self.tensor())

    def __repr__(self):
        return f"{float(self):.6g}"


def clip_grad_value_(parameters, clip_value, norm_type=2):
    """Elementwise clamp of all gradients to +-clip_value; returns the (pre-clamp) global L2 norm
    (reference utils.py:118-132).  No host synchronisation: the norm is accumulated on device and only read if the
    caller converts the return value to float (train.py:145 ignores it).  One launch per gradient buffer; with the
    flat-buffer optimizer (optimize.py) all gradients live in ONE buffer -> one launch per step."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    params = [p for p in parameters if p.grad is not None]
    if float(norm_type) != 2.0:
        raise NotImplementedError("clip_grad_value_: only the L2 norm the reference uses is implemented")
    if not params:
        return 0.0
    dev = params[0].grad.device
    sumsq = torch.zeros(1, device=dev, dtype=torch.float32)
    flat = getattr(params[0], "_glowtts_flat_grad", None)
    owner = getattr(params[0], "_glowtts_flat_owner", None)
    owner = owner() if owner is not None else None
    # one launch over the flat buffer only while EVERY .grad still is the optimizer's view of it: a hook or a wrapper that
    # replaced a .grad leaves the flat slice stale, and clamping it would skip the live gradient (the per-tensor path below)
    if flat is not None and owner is not None and all(getattr(p, "_glowtts_flat_grad", None) is flat for p in params) \
            and sum(p.numel() for p in params) == getattr(params[0], "_glowtts_flat_numel", -1) \
            and owner.flat_g is flat and owner.grads_in_place():
        call("glowtts_clip_grad_value", ptr(flat), flat.numel(), float(clip_value), ptr(sumsq))
    else:
        for p in params:
            g = p.grad.data
            if not g.is_contiguous():
                g = g.contiguous()
                p.grad.data = g
            call("glowtts_clip_grad_value", ptr(g), g.numel(), float(clip_value), ptr(sumsq))
    return _FlatGradView(sumsq, float(norm_type))


def squeeze(x, x_mask=None, n_sqz=2, io_bf16=False):
    """Fold n_sqz consecutive frames into channels (reference utils.py:135-147).  `io_bf16`: the squeezed tensor leaves as
    bf16 (the flow decoder's bf16-tensor mode; not part of the reference signature)."""
    b, c, t = x.size()
    if x_mask is None:
        x_mask = torch.ones(b, 1, t, device=x.device, dtype=torch.float32)
    xs, ms = ops.SqueezeFn.apply(x, ops.mask2d(x_mask), n_sqz, bool(io_bf16))
    return xs, ms.unsqueeze(1)


def unsqueeze(x, x_mask=None, n_sqz=2, io_bf16=False):
    """Inverse of squeeze (reference utils.py:150-160).  `io_bf16`: the squeezed input is bf16; the result is fp32."""
    b, c, t = x.size()
    if x_mask is None:
        x_mask = torch.ones(b, 1, t, device=x.device, dtype=torch.float32)
    xu, mu = ops.UnsqueezeFn.apply(x, ops.mask2d(x_mask), n_sqz, bool(io_bf16))
    return xu, mu.unsqueeze(1)


def to_gpu(x: torch.Tensor) -> torch.Tensor:
    return x.contiguous().cuda(non_blocking=True)
