"""Autograd operators over the HIP kernels (hand-written backward for every flow; the reference relies on autograd
over stock ops).  Every operator takes contiguous fp32 device tensors in the reference layout
``(B, C, T)`` and launches on PyTorch's current stream through the C ABI in ``_hip.py``.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _hip
from ._hip import call, f32, ptr


def _c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.contiguous()


def mask2d(x_mask: torch.Tensor) -> torch.Tensor:
    """(B, 1, T) float mask -> contiguous fp32 (B, T)."""
    m = x_mask.reshape(x_mask.shape[0], x_mask.shape[-1])
    if m.dtype != torch.float32:
        m = m.float()
    return m.contiguous()


def mask_len(m2: torch.Tensor) -> torch.Tensor:
    """x_len[b] = sum_t mask[b, t]  (layers.py:187,245)."""
    out = torch.empty(m2.shape[0], device=m2.device, dtype=torch.float32)
    call("glowtts_mask_len", ptr(m2), ptr(out), m2.shape[0], m2.shape[1])
    return out


_TORCH_MASKS = __import__("os").environ.get("GLOWTTS_KEEP_MASK", "1") == "0"      # tuning / A-B switch: torch's generator

# Dropout decisions are data, not part of the model (reference: F.dropout, layers.py:58,147; attentions.py:67,71,251,379;
# models.py:45,49).  EVERY keep-mask of the package is drawn by keep_mask() below, which names the site it is drawn for, so a
# parity test can (a) read the decisions of a step back (`keep_mask_tap`) and feed them to its checker, or (b) make the step
# use decisions recorded from the reference (`keep_mask_inject`).  Both are None in production.
#   keep_mask_tap(site, mask, p_drop)                 called with every mask drawn (or injected)
#   keep_mask_inject(site, shape, p_drop) -> uint8 device tensor of `shape`, or None to draw as usual
keep_mask_tap = None
keep_mask_inject = None

# Seeds of the Philox launches are drawn the way torch's own CUDA dropout draws them: from the device's default CUDA generator
# — (its seed, its Philox offset), the offset then advanced — NOT from the global CPU generator (that perturbed the
# stream DataLoader shuffling and worker seeding use).  torch.manual_seed(s) resets that generator's seed and offset, so it
# fixes the masks of the steps that follow it, again and again for the same s.
def _next_seed(device) -> int:
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    gen = torch.cuda.default_generators[idx]
    seed, off = gen.initial_seed(), gen.get_offset()
    gen.set_offset(off + 4)                              # (Philox offsets move in fours)
    return (seed * 0x9E3779B97F4A7C15 + off * 0xD1B54A32D192ED03 + 0x632BE59BD9B4E019) % (2 ** 62)


def seed_keep_masks(seed: int) -> None:
    """Restart the keep-mask sequence of the current device from `seed` (= torch.cuda.manual_seed: the masks follow the device's
    default generator): two runs that call this with the same value draw the same masks, whatever was drawn before."""
    torch.cuda.manual_seed(int(seed))


def dropout(x: torch.Tensor, p_drop: float, site: str = "") -> torch.Tensor:
    """F.dropout(x, p, training=True) with the package's keep-masks (the per-operator paths of the text encoder; the executors
    apply their masks inside kernels): x * keep / (1 - p), differentiable through torch's multiply."""
    if p_drop <= 0.0:
        return x
    keep = keep_mask(tuple(x.shape), p_drop, x.device, site)
    return x * (keep.to(x.dtype) * (1.0 / (1.0 - p_drop)))


def keep_mask(shape, p_drop: float, device, site: str = "") -> torch.Tensor:
    """uint8 dropout keep-mask (1 = keep with probability 1 - p_drop) of `shape`, in ONE launch whatever its size
    (`glowtts_keep_mask`: Philox4x32-7, two bytes of randomness per decision, so p is quantised to 1/65536).  The seed comes
    from the device's default CUDA generator (seed + Philox offset, as torch's own dropout): `torch.manual_seed` makes the masks
    repeatable and drawing them does not consume the global CPU stream; inside a graph capture the seed would be frozen into the graph, so
    there the mask is drawn by torch's graph-aware `bernoulli_`.  `site` names what the mask is for (see the hooks above)."""
    shape = tuple(int(d) for d in shape)
    n = 1
    for d in shape:
        n *= d
    out = None
    if keep_mask_inject is not None:
        out = keep_mask_inject(site, shape, float(p_drop))
        if out is not None and (tuple(out.shape) != shape or out.dtype != torch.uint8 or not out.is_cuda
                                or not out.is_contiguous()):
            raise RuntimeError(f"keep_mask_inject({site!r}): need a contiguous uint8 device tensor of shape {shape}")
    if out is None:
        out = torch.empty(shape, device=device, dtype=torch.uint8)
        if n == 0:
            return out
        if torch.cuda.is_current_stream_capturing() or _TORCH_MASKS:
            out.bernoulli_(1.0 - p_drop)
        else:
            call("glowtts_keep_mask", ptr(out), n, _next_seed(device), float(p_drop))
    if keep_mask_tap is not None:
        keep_mask_tap(site, out, float(p_drop))
    return out


# ----------------------------------------------------------------------------------------------------------------
class ActNormFn(Function):
    """z = (bias + exp(logs) x) mask, logdet = sum(logs) x_len   (layers.py:196-197)."""

    @staticmethod
    def forward(ctx, x, m2, logs, bias, x_len):
        x = f32(_c(x))
        B, C, T = x.shape
        z = torch.empty_like(x)
        logdet = torch.empty(B, device=x.device, dtype=torch.float32)
        lg, bs = _c(logs.reshape(-1)), _c(bias.reshape(-1))
        call("glowtts_actnorm_fwd", ptr(x), ptr(m2), ptr(lg), ptr(bs), ptr(x_len), ptr(z), ptr(logdet), B, C, T, 0)
        ctx.save_for_backward(x, m2, lg, x_len)
        ctx.pshape = logs.shape
        return z, logdet

    @staticmethod
    @once_differentiable
    def backward(ctx, dz, dlogdet):
        x, m2, lg, x_len = ctx.saved_tensors
        B, C, T = x.shape
        dz = _c(dz) if dz is not None else torch.zeros_like(x)
        dx = torch.empty_like(x)
        dpar = torch.zeros(2, C, device=x.device, dtype=torch.float32)
        call("glowtts_actnorm_bwd", ptr(x), ptr(m2), ptr(lg), ptr(dz), ptr(_c(dlogdet)), ptr(x_len), ptr(dx),
             ptr(dpar[0]), ptr(dpar[1]), B, C, T)
        return dx, None, dpar[0].view(ctx.pshape), dpar[1].view(ctx.pshape), None


def actnorm_reverse(x, m2, logs, bias):
    x = f32(_c(x))
    B, C, T = x.shape
    z = torch.empty_like(x)
    call("glowtts_actnorm_fwd", ptr(x), ptr(m2), ptr(_c(logs.reshape(-1))), ptr(_c(bias.reshape(-1))), None, ptr(z),
         None, B, C, T, 1)
    return z


def actnorm_stats(x, m2):
    """Masked per-channel sums for the data-dependent init (layers.py:209-211): returns (sum_x, sum_x2, count)."""
    x = f32(_c(x))
    B, C, T = x.shape
    s = torch.zeros(2, C, device=x.device, dtype=torch.float32)
    call("glowtts_actnorm_stats", ptr(x), ptr(m2), ptr(s[0]), ptr(s[1]), B, C, T)
    return s[0], s[1], mask_len(m2).sum()


# ----------------------------------------------------------------------------------------------------------------
def invconv_prepare(weight: torch.Tensor):
    """(W^-1, log det W) of the n x n mixing matrix on one wavefront (replaces torch.inverse / torch.logdet)."""
    w = f32(_c(weight.detach()))
    n = w.shape[0]
    w_inv = torch.empty_like(w)
    logdet_w = torch.empty(1, device=w.device, dtype=torch.float32)
    call("glowtts_invconv_prepare", ptr(w), ptr(w_inv), ptr(logdet_w), n)
    return w_inv, logdet_w


class InvConvFn(Function):
    """layers.py:247-272 without the two permute copies: 4x4 (n_split) mix per (b, group, t) in registers."""

    @staticmethod
    def forward(ctx, x, m2, weight, x_len, n_split):
        x = f32(_c(x))
        B, C, T = x.shape
        w = f32(_c(weight))
        w_inv, logdet_w = invconv_prepare(w)
        z = torch.empty_like(x)
        logdet = torch.empty(B, device=x.device, dtype=torch.float32)
        call("glowtts_invconv_fwd", ptr(x), ptr(m2), ptr(w), ptr(logdet_w), ptr(x_len), ptr(z), ptr(logdet), B, C, T,
             n_split)
        ctx.save_for_backward(x, m2, w, w_inv, x_len)
        ctx.n_split = n_split
        return z, logdet

    @staticmethod
    @once_differentiable
    def backward(ctx, dz, dlogdet):
        x, m2, w, w_inv, x_len = ctx.saved_tensors
        B, C, T = x.shape
        dz = _c(dz) if dz is not None else torch.zeros_like(x)
        dx = torch.empty_like(x)
        dw = torch.zeros_like(w)
        call("glowtts_invconv_bwd", ptr(x), ptr(m2), ptr(w), ptr(w_inv), ptr(dz), ptr(_c(dlogdet)), ptr(x_len), ptr(dx),
             ptr(dw), B, C, T, ctx.n_split)
        return dx, None, dw, None, None


class ActNormInvConvFn(Function):
    """ActNorm followed by InvConvNear (flows 3i, 3i+1 of a decoder block, models.py:176-179) in one pass each way."""

    @staticmethod
    def forward(ctx, x, m2, logs, bias, weight, x_len, n_split):
        x = f32(_c(x))
        B, C, T = x.shape
        w = f32(_c(weight))
        w_inv, logdet_w = invconv_prepare(w)
        lg, bs = _c(logs.reshape(-1)), _c(bias.reshape(-1))
        z = torch.empty_like(x)
        logdet = torch.empty(B, device=x.device, dtype=torch.float32)
        call("glowtts_actnorm_invconv_fwd", ptr(x), ptr(m2), ptr(lg), ptr(bs), ptr(w), ptr(logdet_w), ptr(x_len), ptr(z),
             ptr(logdet), B, C, T, n_split)
        ctx.save_for_backward(x, m2, lg, bs, w, w_inv, x_len)
        ctx.n_split, ctx.pshape = n_split, logs.shape
        return z, logdet

    @staticmethod
    @once_differentiable
    def backward(ctx, dz, dlogdet):
        x, m2, lg, bs, w, w_inv, x_len = ctx.saved_tensors
        B, C, T = x.shape
        dz = _c(dz) if dz is not None else torch.zeros_like(x)
        dx = torch.empty_like(x)
        n = ctx.n_split
        dpar = torch.zeros(2 * C + n * n, device=x.device, dtype=torch.float32)
        dlogs, dbias, dw = dpar[:C], dpar[C:2 * C], dpar[2 * C:]
        call("glowtts_actnorm_invconv_bwd", ptr(x), ptr(m2), ptr(lg), ptr(bs), ptr(w), ptr(w_inv), ptr(dz),
             ptr(_c(dlogdet)), ptr(x_len), ptr(dx), ptr(dlogs), ptr(dbias), ptr(dw), B, C, T, n)
        return dx, None, dlogs.view(ctx.pshape), dbias.view(ctx.pshape), dw.view(n, n), None, None


def invconv_apply(x, m2, w, n_split):
    """Mix with an explicit matrix and no log-det (the reverse path passes the stored inverse, layers.py:254-258)."""
    x = f32(_c(x))
    B, C, T = x.shape
    z = torch.empty_like(x)
    call("glowtts_invconv_fwd", ptr(x), ptr(m2), ptr(f32(_c(w))), None, None, ptr(z), None, B, C, T, n_split)
    return z


# ----------------------------------------------------------------------------------------------------------------
class CouplingFn(Function):
    """attentions.py:128-142: z = [x_0 ; (m + exp(logs) x_1) mask], logdet = sum logs mask; (m, logs) = `out` halves."""

    @staticmethod
    def forward(ctx, x, out, m2, sigmoid_scale, link=None):
        x, out = f32(_c(x)), f32(_c(out))
        B, C, T = x.shape
        z = torch.empty_like(x)
        logdet = _hip.scratch_zeros((B,), x.device)
        call("glowtts_coupling_fwd", ptr(x), ptr(out), ptr(m2), ptr(z), ptr(logdet), B, C, T, int(sigmoid_scale), 0)
        ctx.save_for_backward(x, out, m2)
        ctx.sig = int(sigmoid_scale)
        ctx.link = link
        return z, logdet

    @staticmethod
    @once_differentiable
    def backward(ctx, dz, dlogdet):
        x, out, m2 = ctx.saved_tensors
        B, C, T = x.shape
        dz = _c(dz) if dz is not None else torch.zeros_like(x)
        dx = torch.empty_like(x)
        dout = torch.empty_like(out)
        call("glowtts_coupling_bwd", ptr(x), ptr(out), ptr(m2), ptr(dz), ptr(_c(dlogdet)), ptr(dx), ptr(dout), B, C, T,
             ctx.sig)
        if ctx.link is not None and ctx.link.armed:
            # x also feeds the start conv (through its first half); that conv's backward runs later, adds its part into
            # this buffer in place and hands the sum to autograd — no zeros / slice-copy / add launches for the fan-out
            ctx.link.buf = dx
            return None, dout, None, None, None
        return dx, dout, None, None, None


class GradLink:
    """Side channel between the two consumers of a coupling block's input (the affine apply and the start conv)."""

    def __init__(self):
        self.buf = None
        self.armed = False


def coupling_reverse(x, out, m2, sigmoid_scale):
    x, out = f32(_c(x)), f32(_c(out))
    B, C, T = x.shape
    z = torch.empty_like(x)
    call("glowtts_coupling_fwd", ptr(x), ptr(out), ptr(m2), ptr(z), None, B, C, T, int(sigmoid_scale), 1)
    return z


# ----------------------------------------------------------------------------------------------------------------
class GateFn(Function):
    """utils.py:31-38 with the conditioning row broadcast over time; backward recomputes tanh/sigmoid from `a`."""

    @staticmethod
    def forward(ctx, a, g):
        a = f32(_c(a))
        B, H2, T = a.shape
        H = H2 // 2
        g2 = None if g is None else f32(_c(g.reshape(B, H2)))
        acts = torch.empty(B, H, T, device=a.device, dtype=torch.float32)
        call("glowtts_gate_fwd", ptr(a), ptr(g2), ptr(acts), B, H, T)
        ctx.save_for_backward(a, g2)
        ctx.gshape = None if g is None else g.shape
        return acts

    @staticmethod
    @once_differentiable
    def backward(ctx, dacts):
        a, g2 = ctx.saved_tensors
        B, H2, T = a.shape
        da = torch.empty_like(a)
        call("glowtts_gate_bwd", ptr(a), ptr(g2), ptr(_c(dacts)), ptr(da), B, H2 // 2, T)
        dg = None if g2 is None else da.sum(-1).view(ctx.gshape)
        return da, dg


class ResSkipFn(Function):
    """layers.py:157-162: x <- (x + rs[:, :H]) mask, skip <- skip + rs[:, H:]; the last layer folds `output * mask`."""

    @staticmethod
    def forward(ctx, x, rs, m2, skip_in, last):
        rs = f32(_c(rs))
        B, _, T = rs.shape
        H = rs.shape[1] if last else rs.shape[1] // 2
        skip_in = _c(skip_in)
        skip_out = torch.empty(B, H, T, device=rs.device, dtype=torch.float32)
        if last:
            call("glowtts_res_skip_fwd", None, ptr(rs), ptr(m2), ptr(skip_in), None, ptr(skip_out), B, H, T, 1)
            x_out = None
        else:
            x = f32(_c(x))
            x_out = torch.empty_like(x)
            call("glowtts_res_skip_fwd", ptr(x), ptr(rs), ptr(m2), ptr(skip_in), ptr(x_out), ptr(skip_out), B, H, T, 0)
        ctx.save_for_backward(m2)
        ctx.last, ctx.has_skip, ctx.H = bool(last), skip_in is not None, H
        if last:
            return skip_out
        return x_out, skip_out

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        (m2,) = ctx.saved_tensors
        B, T = m2.shape
        H = ctx.H
        if ctx.last:
            dskip = _c(grads[0])
            drs = torch.empty_like(dskip)
            call("glowtts_res_skip_bwd", None, ptr(dskip), ptr(m2), None, ptr(drs), B, H, T, 1)
            return None, drs, None, (drs if ctx.has_skip else None), None
        dx_out, dskip = grads
        dev = m2.device
        dx_out = _c(dx_out) if dx_out is not None else torch.zeros(B, H, T, device=dev)
        dskip = _c(dskip) if dskip is not None else torch.zeros(B, H, T, device=dev)
        drs = torch.empty(B, 2 * H, T, device=dev, dtype=torch.float32)
        call("glowtts_res_skip_bwd", ptr(dx_out), ptr(dskip), ptr(m2), None, ptr(drs), B, H, T, 0)
        return drs[:, :H], drs, None, (dskip if ctx.has_skip else None), None


# ----------------------------------------------------------------------------------------------------------------
class SqueezeFn(Function):
    """utils.py:135-147 (time -> channel fold, masked); backward is the unsqueeze kernel with the same mask.
    io: the squeezed tensor is bf16 in HBM (the flow decoder's bf16-tensor mode); the un-squeezed side is always fp32."""

    @staticmethod
    def forward(ctx, x, m2, n, io=False):
        x = f32(_c(x))
        B, C, T = x.shape
        Ts = T // n
        xs = torch.empty(B, C * n, Ts, device=x.device, dtype=torch.bfloat16 if io else torch.float32)
        ms = torch.empty(B, Ts, device=x.device, dtype=torch.float32)
        call("glowtts_squeeze_io", ptr(x), ptr(m2), ptr(xs), ptr(ms), B, C, T, n, int(io))
        ctx.save_for_backward(ms)
        ctx.n, ctx.T, ctx.io = n, T, bool(io)
        ctx.mark_non_differentiable(ms)
        return xs, ms

    @staticmethod
    @once_differentiable
    def backward(ctx, dxs, _dms):
        (ms,) = ctx.saved_tensors
        dxs = _c(dxs)
        if ctx.io and dxs.dtype != torch.bfloat16:
            dxs = dxs.to(torch.bfloat16)
        B, Cn, Ts = dxs.shape
        n = ctx.n
        if ctx.T == Ts * n:
            dx = torch.empty(B, Cn // n, ctx.T, device=dxs.device, dtype=torch.float32)
        else:  # frames cut off by the floor division get zero gradient
            dx = torch.zeros(B, Cn // n, ctx.T, device=dxs.device, dtype=torch.float32)
        if ctx.T == Ts * n:
            call("glowtts_unsqueeze_io", ptr(dxs), ptr(ms), ptr(dx), None, B, Cn // n, Ts, n, int(ctx.io))
        else:
            tmp = torch.empty(B, Cn // n, Ts * n, device=dxs.device, dtype=torch.float32)
            call("glowtts_unsqueeze_io", ptr(dxs), ptr(ms), ptr(tmp), None, B, Cn // n, Ts, n, int(ctx.io))
            dx[:, :, : Ts * n] = tmp
        return dx, None, None, None


class UnsqueezeFn(Function):
    """utils.py:150-160; backward is the squeeze kernel (mask value per squeezed column).  io: the squeezed INPUT is bf16."""

    @staticmethod
    def forward(ctx, xs, ms, n, io=False):
        xs = _c(xs)
        if xs.dtype != (torch.bfloat16 if io else torch.float32):
            raise RuntimeError(f"glow_tts_train: unsqueeze got {xs.dtype} with io={io}")
        B, Cn, Ts = xs.shape
        x = torch.empty(B, Cn // n, Ts * n, device=xs.device, dtype=torch.float32)
        mo = torch.empty(B, Ts * n, device=xs.device, dtype=torch.float32)
        call("glowtts_unsqueeze_io", ptr(xs), ptr(ms), ptr(x), ptr(mo), B, Cn // n, Ts, n, int(io))
        ctx.save_for_backward(mo)
        ctx.n, ctx.io = n, bool(io)
        ctx.mark_non_differentiable(mo)
        return x, mo

    @staticmethod
    @once_differentiable
    def backward(ctx, dx, _dmo):
        (mo,) = ctx.saved_tensors
        dx = f32(_c(dx))
        B, C, T = dx.shape
        n = ctx.n
        dxs = torch.empty(B, C * n, T // n, device=dx.device, dtype=torch.bfloat16 if ctx.io else torch.float32)
        call("glowtts_squeeze_io", ptr(dx), ptr(mo), ptr(dxs), None, B, C, T, n, int(ctx.io))
        return dxs, None, None, None


# ----------------------------------------------------------------------------------------------------------------
_HALF_LOG_2PI = 0.5 * math.log(2 * math.pi)


class MleLossFn(Function):
    """utils.py:14-23 as one streaming reduction and a one-workgroup finish (no one-element torch launches)."""

    @staticmethod
    def forward(ctx, z, m, logs, logdet, m2):
        z, m, logs = f32(_c(z)), f32(_c(m)), f32(_c(logs))
        B, C, T = z.shape
        acc = _hip.scratch_zeros((2,), z.device)
        out = torch.empty(2, device=z.device, dtype=torch.float32)          # loss, denominator
        call("glowtts_mle_loss_fwd", ptr(z), ptr(m), ptr(logs), ptr(m2), ptr(f32(_c(logdet))), ptr(acc), ptr(out), B, C, T)
        ctx.save_for_backward(z, m, logs, out)
        ctx.B = B
        return out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss):
        z, m, logs, out = ctx.saved_tensors
        dloss = f32(_c(dloss)).reshape(1)
        dz, dm, dlogs = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        dlogdet = torch.empty(ctx.B, device=z.device, dtype=torch.float32)
        call("glowtts_mle_loss_bwd", ptr(z), ptr(m), ptr(logs), ptr(dloss), ptr(out[1:]), ptr(dz), ptr(dm), ptr(dlogs),
             ptr(dlogdet), ctx.B, z.numel())
        return dz, dm, dlogs, dlogdet, None


class DurationLossFn(Function):
    """utils.py:26-28: sum((logw - logw_)^2) / sum(lengths), one launch each way.  logw_ carries no gradient (the alignment
    is a constant of the step, models.py:383-392)."""

    @staticmethod
    def forward(ctx, logw, logw_, lengths):
        logw, logw_ = f32(_c(logw)), f32(_c(logw_.detach()))
        lengths = _c(lengths.to(device=logw.device, dtype=torch.int64))
        out = torch.empty(2, device=logw.device, dtype=torch.float32)       # loss, denominator
        call("glowtts_duration_loss_fwd", ptr(logw), ptr(logw_), ptr(lengths), ptr(out), lengths.numel(), logw.numel())
        ctx.save_for_backward(logw, logw_, out)
        return out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss):
        logw, logw_, out = ctx.saved_tensors
        dloss = f32(_c(dloss)).reshape(1)
        dlogw = torch.empty_like(logw)
        call("glowtts_duration_loss_bwd", ptr(logw), ptr(logw_), ptr(dloss), ptr(out[1:]), ptr(dlogw), logw.numel())
        return dlogw, None, None


def span_logw(first: torch.Tensor, t_x: torch.Tensor) -> torch.Tensor:
    """log(1e-8 + frames per token) * x_mask from the span table of mas_path_spans: (B, 1, Tx) fp32 (models.py:392)."""
    B, tx1 = first.shape
    out = torch.empty(B, 1, tx1 - 1, device=first.device, dtype=torch.float32)
    t_x = t_x.to(device=first.device, dtype=torch.int32).contiguous()
    call("glowtts_span_logw", ptr(first), ptr(t_x), ptr(out), B, tx1 - 1)
    return out


# ----------------------------------------------------------------------------------------------------------------
def mas_path(value: torch.Tensor, t_x: torch.Tensor, t_y: torch.Tensor) -> torch.Tensor:
    """Monotonic alignment search on device; value (B, Tx, Ty) fp32, lengths int32 (B). Returns 0/1 fp32 path."""
    value = f32(_c(value.detach()))
    B, Tx, Ty = value.shape
    path = torch.empty_like(value)
    t_x = t_x.to(device=value.device, dtype=torch.int32).contiguous()
    t_y = t_y.to(device=value.device, dtype=torch.int32).contiguous()
    call("glowtts_mas_path", ptr(value), ptr(path), ptr(t_x), ptr(t_y), B, Tx, Ty)
    return path


def mas_path_spans(value: torch.Tensor, t_x: torch.Tensor, t_y: torch.Tensor):
    """The search plus its by-products: (path (B,Tx,Ty) 0/1 fp32, first (B,Tx+1) int32, tok (B,Ty) int32).
    Where the multi-wave search takes the lattice (`glowtts_mas_spans_supported`) the kernel hands out the spans only and the
    path is expanded from them by a second launch — inside a training step (`_hip.side_stream_enabled()`: the step joins its
    side streams before the optimizer) on the text encoder's stream, off the serial stretch between forward and backward:
    nothing in the step reads the path, the reference returns it (models.py:394-399)."""
    value = f32(_c(value.detach()))
    B, Tx, Ty = value.shape
    dev = value.device
    path = torch.empty_like(value)
    first = torch.empty(B, Tx + 1, device=dev, dtype=torch.int32)
    tok = torch.empty(B, Ty, device=dev, dtype=torch.int32)
    t_x = t_x.to(device=dev, dtype=torch.int32).contiguous()
    t_y = t_y.to(device=dev, dtype=torch.int32).contiguous()
    if B * Tx * Ty > 0 and _hip.side_stream_enabled() and _hip.load().glowtts_mas_spans_supported(Tx, Ty):
        call("glowtts_mas_path_spans", ptr(value), None, ptr(first), ptr(tok), ptr(t_x), ptr(t_y), B, Tx, Ty)
        cur = torch.cuda.current_stream(dev)
        side = _hip.side_stream(dev, "encoder")
        side.wait_stream(cur)
        _hip.call_on(side.cuda_stream, "glowtts_mas_path_from_spans", ptr(first), ptr(path), B, Tx, Ty)
        for t in (first, path):
            t.record_stream(side)
        return path, first, tok
    call("glowtts_mas_path_spans", ptr(value), ptr(path), ptr(first), ptr(tok), ptr(t_x), ptr(t_y), B, Tx, Ty)
    return path, first, tok


def align_logp(x_m: torch.Tensor, x_logs: Optional[torch.Tensor], z: torch.Tensor) -> torch.Tensor:
    """log N(z_t'; x_m_t, exp(x_logs_t)) for every (token, frame) pair -> (B, Tx, Ty) (reference models.py:362-376), one
    kernel.  x_logs None = log-std 0 (mean_only)."""
    x_m, z = f32(_c(x_m.detach())), f32(_c(z.detach()))
    xl = None if x_logs is None else f32(_c(x_logs.detach()))
    B, C, Tx = x_m.shape
    Ty = z.shape[2]
    logp = torch.empty(B, Tx, Ty, device=z.device, dtype=torch.float32)
    call("glowtts_align_logp", ptr(x_m), ptr(xl), ptr(z), ptr(logp), B, C, Tx, Ty)
    return logp


class AlignExpandFn(Function):
    """z_stats = attn^T stats for a hard monotonic path (reference models.py:383-392) as a gather by the frame -> token map
    the search kernel wrote; backward = segment sums over the tokens' spans."""

    @staticmethod
    def forward(ctx, stats, tok, first):
        stats = f32(_c(stats))
        B, D, Tx = stats.shape
        Ty = tok.shape[1]
        out = torch.empty(B, D, Ty, device=stats.device, dtype=torch.float32)
        call("glowtts_align_expand_fwd", ptr(stats), ptr(tok), ptr(out), B, D, Tx, Ty)
        ctx.save_for_backward(first)
        ctx.shape = (B, D, Tx, Ty)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (first,) = ctx.saved_tensors
        B, D, Tx, Ty = ctx.shape
        dout = f32(_c(dout))
        dstats = torch.empty(B, D, Tx, device=dout.device, dtype=torch.float32)
        call("glowtts_align_expand_bwd", ptr(dout), ptr(first), ptr(dstats), B, D, Tx, Ty)
        return dstats, None, None


class RelAttnFn(Function):
    """Windowed relative-position self-attention (attentions.py:214-264) on the MFMA kernels of csrc/attention.hip.
    q, k, v: (B, H*dk, T); emb_k / emb_v: (1|H, 2w+1, dk) or None; m2: (B, T) sequence mask.  Returns (out, p_attn).
    `bf16_mma`: the contractions on the bf16 matrix pipe with fp32 accumulation (tensors and softmax stay fp32)."""

    @staticmethod
    def forward(ctx, q, k, v, emb_k, emb_v, m2, n_heads, window, block_length, p_drop, bf16_mma=False, site="attn"):
        q, k, v = f32(_c(q)), f32(_c(k)), f32(_c(v))
        B, C, T = q.shape
        dk = C // n_heads
        has_rel = emb_k is not None
        ek = f32(_c(emb_k.detach())) if has_rel else None
        ev = f32(_c(emb_v.detach())) if has_rel else None
        share = int((not has_rel) or emb_k.shape[0] == 1)
        drop = None
        if p_drop > 0.0:
            drop = keep_mask((B, n_heads, T, T), p_drop, q.device, site)   # keep = 1
        p_attn = torch.empty(B, n_heads, T, T, device=q.device, dtype=torch.float32)
        out = torch.empty_like(q)
        scale = 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0
        call("glowtts_rel_attn_fwd_ex", ptr(q), ptr(k), ptr(v), ptr(ek), ptr(ev), ptr(m2), ptr(drop), scale, ptr(p_attn),
             ptr(out), B, n_heads, T, dk, window if has_rel else 0, share, -1 if block_length is None else block_length,
             int(bool(bf16_mma)))
        ctx.save_for_backward(q, k, v, ek, ev, m2, drop, p_attn)
        ctx.cfg = (n_heads, window if has_rel else 0, share, -1 if block_length is None else block_length, scale,
                   int(bool(bf16_mma)))
        ctx.eshape = None if not has_rel else emb_k.shape
        ctx.mark_non_differentiable(p_attn)
        return out, p_attn

    @staticmethod
    @once_differentiable
    def backward(ctx, dout, _dp):
        q, k, v, ek, ev, m2, drop, p_attn = ctx.saved_tensors
        n_heads, window, share, blk, scale, bf16_mma = ctx.cfg
        B, C, T = q.shape
        dk = C // n_heads
        dout = _c(dout)
        ds = torch.empty_like(p_attn)
        dq, dk_, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        dek = dev = None
        if ek is not None:
            dek, dev = torch.zeros_like(ek), torch.zeros_like(ev)
        call("glowtts_rel_attn_bwd_ex", ptr(dout), ptr(q), ptr(k), ptr(v), ptr(ek), ptr(ev), ptr(m2), ptr(drop), scale,
             ptr(p_attn), ptr(ds), ptr(dq), ptr(dk_), ptr(dv), ptr(dek), ptr(dev), B, n_heads, T, dk, window, share, blk,
             bf16_mma)
        return dq, dk_, dv, dek, dev, None, None, None, None, None, None, None


def library_loaded() -> bool:
    return _hip._lib is not None
