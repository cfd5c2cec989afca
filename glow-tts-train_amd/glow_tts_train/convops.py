"""Autograd operators over the fp32-MFMA convolution kernels (csrc/convgemm.hip): a weight-normed 1-D convolution and
the whole WaveNet-style WN stack with a hand-composed backward.

The reference leaves these contractions to cuDNN/MIOpen through `F.conv1d` and autograd (layers.py:138-162,
attentions.py:124-126).  Here every contraction is one launch of the implicit-GEMM kernel family with bias / mask /
gate / residual-skip fused into the epilogue; parameter gradients are accumulated straight into `param.grad` when it
already exists (the flat-buffer optimizer keeps it allocated), which removes one `add` launch per parameter per step.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as _dist
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _hip, ops
from ._hip import call, conv_bind_planes, conv_math, f32, ptr, scratch_zeros

# parameters whose .grad was written directly by a backward (no AccumulateGrad node runs for them); a data-parallel
# reducer subscribes here to learn that a gradient is complete (parallel.FlowBlockReducer)
_grad_ready_listeners: List = []

# Operators add weight gradients straight into an existing `.grad` (no autograd add, no AccumulateGrad node).  A wrapper
# that relies on AccumulateGrad hooks for EVERY parameter — torch's DistributedDataParallel, which the reference's
# unchanged `__main__.py:268-271` puts around the model — never sees such a gradient, so the choice is made per step:
#   * a gradient listener is attached (`parallel.FlowBlockReducer` listens to `_notify`)      -> in place;
#   * a process group exists and nobody listens (some other wrapper owns the gradients: DDP)  -> through autograd;
#   * no process group                                                                         -> in place.
# `set_direct_grads(True|False)` or GLOWTTS_DIRECT_GRADS=1|0 pins the choice; `set_direct_grads(None)` returns to auto.
_DIRECT_GRADS: Optional[bool] = {"0": False, "1": True}.get(os.environ.get("GLOWTTS_DIRECT_GRADS", ""))


def set_direct_grads(enabled: Optional[bool]) -> None:
    global _DIRECT_GRADS
    _DIRECT_GRADS = None if enabled is None else bool(enabled)


def direct_grads_enabled() -> bool:
    """May operators write parameter gradients in place this step?  (see the comment above `_DIRECT_GRADS`)"""
    if _DIRECT_GRADS is not None:
        return _DIRECT_GRADS
    if _grad_ready_listeners:
        return True
    return not (_dist.is_available() and _dist.is_initialized())


def _mark_direct(params, flag: bool) -> None:
    """Tell gradient listeners which signal to trust for a parameter.  Autograd runs a parameter's AccumulateGrad node —
    and its post-accumulate hooks — even when an operator returns None for it, i.e. right after that operator's backward
    returns and possibly BEFORE the in-place gradient is complete (a ConvGroup un-packs later, on another stream).  For a
    marked parameter only `_notify` counts (parallel.FlowBlockReducer skips the hook)."""
    for p in params:
        if p is not None:
            p._glowtts_direct = flag


def add_grad_ready_listener(fn) -> None:
    _grad_ready_listeners.append(fn)


def remove_grad_ready_listener(fn) -> None:
    if fn in _grad_ready_listeners:
        _grad_ready_listeners.remove(fn)


def _notify(params) -> None:
    """The gradients of `params` are complete on the current stream.  A listener is called ONCE per group with the list (a flow
    block announces 56 tensors at a time: one call, not 56, on the backward thread)."""
    for fn in _grad_ready_listeners:
        fn(params)


class ParamPack(tuple):
    """The parameters of a whole-stack node handed to `Function.apply` as ONE opaque argument (autograd looks at top-level tensor
    arguments only).  The stack nodes exist in in-place gradient mode only: their backward writes every parameter gradient into
    `.grad` itself and hands autograd None — yet with the 400-odd parameters as tensor inputs the engine still ran an
    AccumulateGrad node for each of them, ~1 us apiece on the backward thread (0.5 ms per step at the default model:
    tools/dp_probe.py's host profile).  One parameter stays a real input (the `anchor`: the node needs an input that requires
    grad to be part of the graph at all — the mel frames do not)."""
    __slots__ = ()


def _unpack_params(params):
    """(parameter tensors, number of apply() arguments they took): `*params` is either the tensors themselves or (anchor, ParamPack)."""
    if len(params) == 2 and isinstance(params[1], ParamPack):
        return tuple(params[1]), 2
    return params, len(params)


def _rows_ok(x: torch.Tensor) -> bool:
    """(B, C, T) tensor whose rows are dense and T-contiguous inside each utterance (a channel slice qualifies)."""
    return x.dim() == 3 and x.stride(2) == 1 and x.stride(1) == x.shape[2]


def _dense(x: torch.Tensor) -> torch.Tensor:
    return x if _rows_ok(x) else x.contiguous()


def pack_weight(v: torch.Tensor, g: Optional[torch.Tensor], want_bwd: bool = True):
    """weight (Cout, Cin, taps) [+ weight-norm gain (Cout,1,1)] -> k-packed forward / backward-data layouts
    wp_f[tap][ceil(Cin/16)][Cout][16], wp_b[tap][ceil(Cout/16)][Cin][16] (taps flipped), and 1/||v|| per output channel."""
    v = f32(v.detach().contiguous())
    cout, cin, taps = v.shape
    dev = v.device
    gi, go = (cin + 15) // 16, (cout + 15) // 16
    new_f = torch.empty if cin % 16 == 0 else torch.zeros       # k slots past the channel count must read as zero
    new_b = torch.empty if cout % 16 == 0 else torch.zeros
    wp_f = new_f(taps, gi, cout, 16, device=dev, dtype=torch.float32)
    wp_b = new_b(taps, go, cin, 16, device=dev, dtype=torch.float32) if want_bwd else None
    inv = torch.empty(cout, device=dev, dtype=torch.float32) if g is not None else None
    gg = None if g is None else f32(g.detach().reshape(-1).contiguous())
    call("glowtts_pack_weight", ptr(v), ptr(gg), ptr(wp_f), ptr(wp_b), ptr(inv), cout, cin, taps)
    return wp_f, wp_b, inv


def conv_fwd(x, wp, bias, m2, y, cin, cout, taps, dil, pad, mask_in=False, mask_out=False, addend=None, mask_add=False):
    B, _, T = x.shape
    call("glowtts_conv_fwd", ptr_rows(x), x.stride(0), ptr(wp), ptr(bias), ptr(m2),
         None if addend is None else ptr_rows(addend), 0 if addend is None else addend.stride(0),
         ptr_rows(y), y.stride(0), B, cin, cout, T, taps, dil, pad, int(mask_in), int(mask_out), int(mask_add),
         tag=("M%d K%dx%d N%dx%d", cout, cin, taps, B, T))
    return y


def ptr_rows(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("glow_tts_train (MI355X build): operator received a CPU tensor")
    if not _rows_ok(t):
        raise RuntimeError("glow_tts_train: tensor rows are not dense")
    return t.data_ptr()


# Tuning knob, OFF by default: GLOWTTS_WGRAD_PRIO=-1 makes the weight-gradient stream a high-priority HIP stream.  On one
# GPU without collectives it shortens the step a little (17.60 -> 17.47 ms: its queue is the one every backward ends with), but
# with RCCL collectives on another stream ANY high-priority stream in the process nearly doubles the step (bench.py
# --rccl-self: 18.2 -> 32 ms, whichever of the weight-gradient and communication streams has the priority).
_WGRAD_PRIO = int(os.environ.get("GLOWTTS_WGRAD_PRIO", "0"))
# ... and the LAST block of the backward (the decoder's first) keeps its weight gradients on the chain's stream: by then the
# weight-gradient stream has a backlog and the chain nothing else to do (0 / 1 / 2 blocks: 17.68 / 17.40 / 17.60 ms per step).
_WGRAD_MAIN_BLOCKS = int(os.environ.get("GLOWTTS_WGRAD_MAIN_BLOCKS", "1"))


class _WgradStream:
    """Weight-gradient launches of a backward pass go to a second HIP stream: they read tensors that already exist (layer
    input, output gradient) and nothing on the dx chain waits for them, so their workgroups fill the prologue / epilogue /
    tail bubbles of the backward-data kernels running on the main stream (-1.1 ms per step at config 2).  `join()` before
    the packed gradients are consumed.  Inactive outside a training step and while per-launch timings are collected."""

    def __init__(self, device):
        self.enabled = _hip.side_stream_enabled()
        if self.enabled:
            self.main = torch.cuda.current_stream(device)
            self.side = _hip.side_stream(device, "wgrad", priority=_WGRAD_PRIO)

    def run(self, fn, *reads):
        if not self.enabled:
            fn()
            return
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side):
            fn()
        for t in reads:                       # allocated on the main stream, read on the side stream: the caching allocator
            if t is not None:                 # must not hand the block out again (saved tensors are freed as soon as this
                t.record_stream(self.side)    # backward returns) before the side stream is done with it

    def join(self):
        if self.enabled:
            self.main.wait_stream(self.side)


class _GradSink:
    """Where a parameter gradient goes: in place into param.grad (no autograd add) or into a fresh zero tensor."""

    def __init__(self, params):
        self.params = list(params)
        self.direct = direct_grads_enabled() and all(
            p is None or (p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32)
            for p in self.params)
        _mark_direct(self.params, self.direct)
        self.bufs = [None if p is None else (p.grad if self.direct else torch.zeros_like(p)) for p in self.params]

    def buf(self, i):
        return self.bufs[i]

    def results(self):
        if self.direct:
            _notify([p for p in self.params if p is not None])
            return [None] * len(self.params)
        return self.bufs


def _weight_grads(x, d, m2_for_d, dwp_shape, v, g, inv, dv_buf, dg_buf, db_buf, taps, dil, pad, dwp=None, unpack=True,
                  m2_for_x=None):
    """dW (through the weight norm) and dbias of y = conv(x): one wrw launch (bias row sums ride along), one unpack."""
    B, cin, T = x.shape
    cout = d.shape[1]
    if dwp is None:
        dwp = scratch_zeros(dwp_shape, d.device)
    call("glowtts_conv_wrw", ptr_rows(x), x.stride(0), ptr_rows(d), d.stride(0), ptr(m2_for_d), ptr(m2_for_x), ptr(dwp),
         None if db_buf is None else ptr(db_buf), B, cin, cout, T, taps, dil, pad, tag=("M%d K%dx%d N%dx%d", cout, cin, taps, B, T))
    if not unpack:
        return
    gg = None if g is None else g.detach().reshape(-1).contiguous()
    call("glowtts_unpack_weight_grad", ptr(dwp), ptr(v.detach().contiguous()), ptr(gg), ptr(inv), ptr(dv_buf),
         None if dg_buf is None else ptr(dg_buf), cout, cin, taps)


class Conv1dFn(Function):
    """y = conv1d(x [* mask]; weight_norm(v, g) or v, bias, dilation, 'same' padding) [* mask] on the MFMA kernels.
    `mask_in` / `mask_out` fold the reference's `conv(x * x_mask)` / `conv(...) * x_mask` into the kernel."""

    @staticmethod
    def forward(ctx, x, v, g, bias, m2, mask_in, mask_out, dil, group=None, gidx=0, link=None):
        # link (ops.GradLink): x is the FULL tensor of a coupling block, the conv reads its first Cin channels; in backward
        # the input gradient is added in place into the buffer the affine apply left in the link
        ctx.link = link
        ctx.x_full_shape = None
        if link is not None:
            ctx.x_full_shape = tuple(x.shape)
            x = x[:, : v.shape[1]]
        x = _dense(f32(x))
        B, cin, T = x.shape
        cout, _, taps = v.shape
        pad = (taps * dil - dil) // 2
        if group is not None:
            wp_f, wp_b, inv = group.packed(gidx)         # (conv1d counted this use in group.pending: grad mode is off in here)
        else:
            wp_f, wp_b, inv = pack_weight(v, g)
        ctx.group = (group, gidx)
        y = torch.empty(B, cout, T, device=x.device, dtype=torch.float32)
        b1 = None if bias is None else f32(bias.detach().contiguous())
        bound = group.plan.bind() if group is not None else False          # (a group with bf16 planes: the selected conv arithmetic)
        try:
            conv_fwd(x, wp_f, b1, m2, y, cin, cout, taps, dil, pad, mask_in=bool(mask_in), mask_out=bool(mask_out))
        finally:
            WNPackPlan.unbind(bound)
        ctx.save_for_backward(x, wp_b, inv, m2 if (mask_out or mask_in) else None)
        ctx.params = (v, g, bias)
        ctx.cfg = (taps, dil, pad, bool(mask_in), bool(mask_out))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, wp_b, inv, m2 = ctx.saved_tensors
        v, g, bias = ctx.params
        taps, dil, pad, mask_in, mask_out = ctx.cfg
        dy = _dense(dy)
        B, cin, T = x.shape
        cout = dy.shape[1]
        group, gidx = ctx.group
        dx = None
        bound = group.plan.bind() if group is not None else False
        try:
            if ctx.needs_input_grad[0]:
                link = ctx.link
                if link is not None:
                    full = link.buf                     # gradient of the other consumer of x (None if it did not run)
                    link.buf = None
                    if full is None:
                        full = torch.zeros(ctx.x_full_shape, device=dy.device, dtype=torch.float32)
                    part = full[:, :cin]
                    conv_fwd(dy, wp_b, None, m2, part, cout, cin, taps, dil, (taps - 1) * dil - pad, mask_in=mask_out,
                             mask_out=mask_in, addend=part)
                    dx = full
                else:
                    dx = torch.empty(B, cin, T, device=dy.device, dtype=torch.float32)
                    conv_fwd(dy, wp_b, None, m2, dx, cout, cin, taps, dil, (taps - 1) * dil - pad, mask_in=mask_out,
                             mask_out=mask_in)
        finally:
            WNPackPlan.unbind(bound)
        if group is not None:
            # packed weight gradient into the group's accumulator (un-packed once, by the group's last backward); the bias
            # gradient rides along with the same launch, straight into bias.grad
            _weight_grads(x, dy, m2 if mask_out else None, (taps, cin, cout), v, g, inv, None, None,
                          None if bias is None else bias.grad, taps, dil, pad, dwp=group.dwp(gidx), unpack=False,
                          m2_for_x=m2 if mask_in else None)
            if bias is not None:
                _notify([bias])
            group.finish_one()
            return dx, None, None, None, None, None, None, None, None, None, None
        sink = _GradSink([v, g, bias])
        _weight_grads(x, dy, m2 if mask_out else None, (taps, cin, cout), v, g, inv, sink.buf(0), sink.buf(1), sink.buf(2),
                      taps, dil, pad, m2_for_x=m2 if mask_in else None)
        dv, dg, db = sink.results()
        return dx, dv, dg, db, None, None, None, None, None, None, None


_conv1d_apply = _hip.direct_apply(Conv1dFn)


def conv1d(conv: torch.nn.Module, x: torch.Tensor, m2: Optional[torch.Tensor] = None, mask_in: bool = False,
           mask_out: bool = False, link=None) -> torch.Tensor:
    """Run an nn.Conv1d (optionally weight-normed, 'same' padding) through the MFMA kernels."""
    if x.dim() == 3 and x.shape[2] == 1 and x.shape[0] > 1 and m2 is None and conv.kernel_size[0] == 1:
        # One frame per utterance (the speaker rows g (B, gin, 1) through WN.cond_layer, reference layers.py:142-143): a 1x1 convolution
        # treats every frame alike, so the B single-frame utterances are run as ONE utterance of B frames — the kernels tile frames,
        # and B one-frame tiles made the input gradient of the conditioning layer (M = gin, K = 2H * n_layers) a 192 us launch
        return conv1d(conv, x.transpose(0, 2).contiguous(), link=link).transpose(0, 2).contiguous()
    if hasattr(conv, "weight_v"):
        v, g = conv.weight_v, conv.weight_g
    else:
        v, g = conv.weight, None
    group, gidx = getattr(conv, "_glowtts_group", (None, 0))
    if group is not None and not group.active:
        group = None
    if group is not None and torch.is_grad_enabled() and (v.requires_grad or x.requires_grad):
        group.pending += 1                               # one backward call to wait for before the group un-packs
    return _conv1d_apply(x, v, g, conv.bias, m2, mask_in, mask_out, conv.dilation[0], group, gidx, link)


class ChanLayerNormFn(Function):
    """y = post(LayerNorm_over_channels(pre(x) + res)) on (B, C, T) (reference layers.py:19-28 + the residual add before it).
    pre = ReLU when `relu_in` (duration predictor: conv -> ReLU -> LayerNorm, models.py:45-46); post = Dropout(ReLU(.)) by
    `relu_out` / `p_drop` (pre-net: LayerNorm -> ReLU -> Dropout, layers.py:73-80; duration predictor: LayerNorm -> Dropout):
    the element-wise neighbours of a norm ride in its kernels (csrc/norm.hip) — no launch, no tensor of their own."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, relu_in=False, relu_out=False, p_drop=0.0, site="ln"):
        if relu_in and res is not None:
            # the kernel gates dx by x > 0 and has no separate residual gradient in this form; no caller needs it
            raise RuntimeError("LayerNorm: relu_in together with a residual input is not supported")
        x = f32(x.contiguous())
        res = None if res is None else f32(res.contiguous())
        B, C, T = x.shape
        y = torch.empty_like(x)
        stats = torch.empty(B, 2, T, device=x.device, dtype=torch.float32)
        keep, scale = None, 1.0
        if p_drop > 0.0:
            keep = ops.keep_mask((B, C, T), p_drop, x.device, site)
            scale = 1.0 / (1.0 - p_drop)
        call("glowtts_chan_layernorm_fwd_act", ptr(x), ptr(res), None, None, 1.0, ptr(gamma.detach().contiguous()),
             ptr(beta.detach().contiguous()), ptr(y), ptr(stats), int(relu_in), int(relu_out), ptr(keep), scale, B, C, T, float(eps))
        ctx.save_for_backward(x, res, stats, *([y] if relu_out else []), *([keep] if keep is not None else []))
        ctx.params = (gamma, beta)
        ctx.opts = (bool(relu_in), bool(relu_out), keep is not None, scale)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        relu_in, relu_out, has_keep, scale = ctx.opts
        saved = list(ctx.saved_tensors)
        x, res, stats = saved[:3]
        y = saved[3] if relu_out else None
        keep = saved[-1] if has_keep else None
        gamma, beta = ctx.params
        B, C, T = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        sink = _GradSink([gamma, beta])
        call("glowtts_chan_layernorm_bwd_act", ptr(x), ptr(res), None, None, 1.0, ptr(gamma.detach().contiguous()), ptr(stats),
             ptr(y), ptr(dy), int(relu_in), int(relu_out), ptr(keep), scale, ptr(dx), None, ptr(sink.buf(0)), ptr(sink.buf(1)),
             B, C, T)
        dg, db = sink.results()
        return dx, (dx if res is not None else None), dg, db, None, None, None, None, None


class EmbedFn(Function):
    """h = weight[ids] * scale as (B, H, T) (reference models.py:121-122: `self.emb(x) * sqrt(hidden)` then the transpose): a
    gather kernel, and in the backward one workgroup per vocabulary entry sums the gradient columns of its positions
    (csrc/train_ops.hip) — torch's gather + mul + transpose copy and its sort-based embedding backward are gone."""

    @staticmethod
    def forward(ctx, ids, weight, scale):
        ids = ids.contiguous()
        if ids.dtype != torch.int64:
            ids = ids.long()
        B, T = ids.shape
        V, H = weight.shape
        out = torch.empty(B, H, T, device=weight.device, dtype=torch.float32)
        call("glowtts_embed_fwd", ptr(ids), ptr(f32(weight.detach().contiguous())), float(scale), ptr(out), B, T, H, V)
        ctx.save_for_backward(ids)
        ctx.weight, ctx.scale = weight, float(scale)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        weight = ctx.weight
        B, T = ids.shape
        V, H = weight.shape
        sink = _GradSink([weight])
        call("glowtts_embed_bwd", ptr(ids), ptr(f32(dout.contiguous())), ctx.scale, ptr(sink.buf(0)), B, T, H, V)
        (dw,) = sink.results()
        return None, dw, None


# Arithmetic of the WN-stack convolutions (see _hip.conv_math / include/glowtts_hip.h).  GLOWTTS_CONV_MATH =
#   bf16x6+wrw (default) — each fp32 operand split exactly into three bf16 planes, the six products above 2^-24 on the bf16
#       matrix pipe, fp32 accumulation, for forward, backward-data and weight gradient: fp32-equivalent results (error
#       against fp64 no larger than the native kernels', tests/test_conv_math.py; every golden-vector and CPU-restatement GPU test runs in
#       both forms at the same tolerances, tests/test_hip_parity.py `conv_mode`), 17 % less step time at config 2;
#   fp32 — the native fp32 MFMA kernels; bf16x6 | bf16x3 | bf16 [+wrw] — the other plane counts (bf16: plain bf16 operands).
# The C library itself starts in native fp32 (glowtts_conv_math(0)); this package selects its default at import.
DEFAULT_CONV_MATH = "bf16x6+wrw"
_SPLIT_MATH = [False]
_MATH_NAME = ["fp32"]


def set_conv_math(mode) -> str:
    """Select the arithmetic of the WN-stack convolutions; returns the previous setting (a name accepted here)."""
    before = _MATH_NAME[0]
    if isinstance(mode, int):
        mode = {v: k for k, v in _hip.CONV_MATH_MODES.items()}[mode & 3] + ("+wrw" if (mode >> 2) & 3 else "")
    conv_math(mode)
    _SPLIT_MATH[0] = (conv_math(None) & 3) != 0
    _MATH_NAME[0] = mode
    return before


def conv_math_name() -> str:
    return _MATH_NAME[0]


try:
    set_conv_math(os.environ.get("GLOWTTS_CONV_MATH", DEFAULT_CONV_MATH))
except _hip.HipLibraryMissing:          # importable without the library (CPU-only host logic); any launch still raises
    pass


class WNPackPlan:
    """Persistent packed-weight / packed-gradient buffers and device descriptor tables of one WN stack, so that weight
    norm + packing of all 2*n_layers convolutions is ONE launch per forward and their un-packing ONE launch per backward
    (the tables hold raw device pointers, valid as long as the parameter storage does not move: checked by key)."""

    def __init__(self, want_planes: bool = False, wino: bool = False):
        self.key = None
        self.gkey = None
        self.want_planes = want_planes       # only a WN stack's convolutions have bf16-plane kernels (convgemm_split.hip)
        self.wino = wino                     # a flow block's plan: its gated in-convs also get Winograd-domain planes (convwino.hip)
        self.shared = None                   # StackArena this plan's packed weights live in (one pack launch per stack), or None

    @staticmethod
    def make_key(params):
        return tuple(0 if p is None else p.data_ptr() for p in params) + tuple(None if p is None else tuple(p.shape) for p in params)

    @staticmethod
    def arena_floats(params, n_convs):
        shapes = [tuple(params[3 * i].shape) for i in range(n_convs)]
        return sum(t * ((ci + 15) // 16) * co * 16 + t * ((co + 15) // 16) * ci * 16 for co, ci, t in shapes)

    def ensure(self, params, n_layers=None, n_convs=None, arena=None, shared=None):
        """params: (v, g, bias) per convolution; a WN stack passes n_layers (2 convolutions per layer).  `arena` / `shared`: the
        slice of a StackArena's buffer this plan packs into (StackArena.ensure) instead of a buffer of its own."""
        n_convs = 2 * n_layers if n_convs is None else n_convs
        key = self.make_key(params)
        if key == self.key and arena is None:
            return
        self.shared = shared                 # (a plan whose parameters moved leaves its stack's buffer: the stack rebuilds)
        dev = params[0].device
        self.convs = []          # (v, g, wp_f, wp_b, inv, cout, cin, taps, dwp_offset)
        rows, off = [0], 0
        # all packed weights of the stack live in ONE buffer (zero-filled once: k positions beyond a channel count stay
        # zero), so the optional split into bf16 planes (glowtts_conv_math) is one launch over it
        shapes = [tuple(params[3 * i].shape) for i in range(n_convs)]
        sizes = [(t * ((ci + 15) // 16) * co * 16, t * ((co + 15) // 16) * ci * 16) for co, ci, t in shapes]
        self.wp_arena = torch.zeros(sum(a + b for a, b in sizes), device=dev, dtype=torch.float32) if arena is None else arena
        assert self.wp_arena.numel() == sum(a + b for a, b in sizes)
        self.wp_planes = None                # bf16 planes of wp_arena (3 x n uint16), made on first use of a split mode
        cursor = 0
        for i in range(n_convs):
            v, g = params[3 * i], params[3 * i + 1]
            cout, cin, taps = v.shape
            gi, go = (cin + 15) // 16, (cout + 15) // 16
            wp_f = self.wp_arena[cursor: cursor + sizes[i][0]].view(taps, gi, cout, 16)
            wp_b = self.wp_arena[cursor + sizes[i][0]: cursor + sizes[i][0] + sizes[i][1]].view(taps, go, cin, 16)
            cursor += sizes[i][0] + sizes[i][1]
            inv = torch.empty(cout, device=dev) if g is not None else None
            self.convs.append((v, g, wp_f, wp_b, inv, cout, cin, taps, off))
            off += taps * cin * cout
            rows.append(rows[-1] + cout)
        self.dwp = torch.empty(off, device=dev, dtype=torch.float32)
        self.total_rows = rows[-1]
        self.prefix = torch.tensor(rows, dtype=torch.int32).to(dev)
        self.rows = rows
        self.desc = torch.tensor(
            [[v.data_ptr(), 0 if g is None else g.data_ptr(), f.data_ptr(), b.data_ptr(), 0 if inv is None else inv.data_ptr(),
              cout, cin, taps] for (v, g, f, b, inv, cout, cin, taps, _) in self.convs], dtype=torch.int64).to(dev)
        self.key, self.gkey = key, None
        self.version = getattr(self, "version", 0) + 1

    def _plane_home(self, make: bool):
        """(packed buffer, its bf16 planes) this plan's planes are addressed in: its own, or its stack's."""
        if self.shared is not None:
            return self.shared.arena, self.shared.planes_buffer(make)
        if make and self.wp_planes is None:       # (zeros: the k positions no packing covers must read as zero in every plane)
            self.wp_planes = torch.zeros(3 * self.wp_arena.numel(), device=self.wp_arena.device, dtype=torch.int16)
        return self.wp_arena, self.wp_planes

    def pack(self):
        self.pack_count = getattr(self, "pack_count", 0) + 1      # (planes made from the packed weights go stale here)
        if _SPLIT_MATH[0] and self.want_planes:   # bf16-plane arithmetic is on: the planes of the new weights in the same pass
            arena, planes = self._plane_home(True)
            call("glowtts_pack_weight_planes_multi", ptr(self.desc), ptr(self.prefix), len(self.convs), self.total_rows,
                 ptr(arena), arena.numel(), ptr(planes))
            if self.shared is None and self.wino and _WINO and _hip.get_knob("WINO"):
                _wino_update(self, arena, [self])     # (a block with its own packed buffer: the per-block path, e.g. configs[4])
        else:
            call("glowtts_pack_weight_multi", ptr(self.desc), ptr(self.prefix), len(self.convs), self.total_rows)

    def bind(self) -> bool:
        """Hand this stack's planes to the calling thread's next convolution launches (no-op in native fp32 mode)."""
        if _SPLIT_MATH[0]:
            arena, planes = self._plane_home(False)
            if planes is not None:
                conv_bind_planes(arena, planes)
                owner = self.shared if self.shared is not None else self
                wino = getattr(owner, "wino_planes", None)
                if wino is not None and getattr(owner, "wino_key", None) is arena:
                    _hip.conv_bind_wino(arena, wino)     # (used by the gated in-conv while the switch GLOWTTS_WINO is on)
                return True
        return False

    @staticmethod
    def unbind(bound: bool):
        if bound:
            conv_bind_planes(None)
            _hip.conv_bind_wino(None)

    def dwp_view(self, i):
        cout, cin, taps, off = self.convs[i][5:9]
        return self.dwp[off: off + taps * cin * cout]

    def layer_table(self, params, n_layers):
        """Host-side `glowtts_wn_layer[n_layers]` for the native executor (csrc/wn_stack.hip): packed weights from this
        plan, biases and gradient targets from the parameters (which must have contiguous fp32 .grad).  Cached."""
        key = (self.version, tuple(0 if p is None else p.data_ptr() for p in params),
               tuple(0 if (p is None or p.grad is None) else p.grad.data_ptr() for p in params))
        if getattr(self, "_ltab_key", None) != key:
            tab = (_hip.WnLayer * n_layers)()
            for i in range(n_layers):
                in_b, rs_b = params[6 * i + 2], params[6 * i + 5]
                ci, cr = self.convs[2 * i], self.convs[2 * i + 1]
                tab[i].wf_in, tab[i].wb_in, tab[i].b_in = ci[2].data_ptr(), ci[3].data_ptr(), in_b.data_ptr()
                tab[i].wf_rs, tab[i].wb_rs, tab[i].b_rs = cr[2].data_ptr(), cr[3].data_ptr(), rs_b.data_ptr()
                tab[i].dwp_in, tab[i].dwp_rs = self.dwp_view(2 * i).data_ptr(), self.dwp_view(2 * i + 1).data_ptr()
                tab[i].db_in = 0 if in_b.grad is None else in_b.grad.data_ptr()
                tab[i].db_rs = 0 if rs_b.grad is None else rs_b.grad.data_ptr()
            self._ltab, self._ltab_key = tab, key
        return self._ltab

    def unpack_tables(self, params):
        """Device descriptor table of `unpack_into_grads` without launching it (the native executor launches it)."""
        gkey = tuple(0 if p is None else p.grad.data_ptr() for p in params)
        if gkey != self.gkey:
            self._build_gdesc(params, gkey)
        return self.gdesc, self.prefix

    def _build_gdesc(self, params, gkey):
        rows = []
        for i, (v, g, _, _, inv, cout, cin, taps, off) in enumerate(self.convs):
            pv, pg = params[3 * i], params[3 * i + 1]
            rows.append([self.dwp.data_ptr() + 4 * off, v.data_ptr(), 0 if g is None else g.data_ptr(),
                         0 if inv is None else inv.data_ptr(), pv.grad.data_ptr(), 0 if pg is None else pg.grad.data_ptr(),
                         cout, cin, taps])
        self.gdesc = torch.tensor(rows, dtype=torch.int64).to(self.dwp.device)
        self.gkey = gkey

    def unpack_into_grads(self, params):
        gkey = tuple(0 if p is None else p.grad.data_ptr() for p in params)
        if gkey != self.gkey:
            self._build_gdesc(params, gkey)
        call("glowtts_unpack_weight_grad_multi", ptr(self.gdesc), ptr(self.prefix), len(self.convs), self.total_rows)


def _process_group_active() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


class StackArena:
    """ONE packed-weight buffer and ONE set of bf16 planes for all blocks of a flow stack (round 4): weight norm + packing + plane
    split of the whole decoder is one launch per step where every block launched its own (17 us each, twelve of them on the
    decoder's chain at config 2).  The blocks' plans keep their tables; their buffers are slices of this one, their plane bindings
    point at it.  A plan whose parameters moved drops out (WNPackPlan.ensure) and the next `pack` rebuilds the stack."""

    def __init__(self):
        self.keys = None
        self.planes = None

    def planes_buffer(self, make: bool):
        if make and self.planes is None:          # (zeros: the k positions no packing covers must read as zero in every plane)
            self.planes = torch.zeros(3 * self.arena.numel(), device=self.arena.device, dtype=torch.int16)
        return self.planes

    def pack(self, plans, conv_params, n_convs):
        keys = tuple(WNPackPlan.make_key(cp) for cp in conv_params)
        if keys != self.keys or any(p.shared is not self for p in plans):
            sizes = [WNPackPlan.arena_floats(cp, n_convs) for cp in conv_params]
            if getattr(self, "arena", None) is not None and self.arena.is_cuda and not torch.cuda.is_current_stream_capturing():
                # a REBUILD (a parameter moved: rare) drops the old packed buffer and planes, which kernels queued on the forward
                # chains' and the weight-gradient streams may still read and whose addresses the cached tables hold: wait for the
                # device once before the allocator may hand that memory out again
                torch.cuda.synchronize(self.arena.device)
            self.arena = torch.zeros(sum(sizes), device=conv_params[0][0].device, dtype=torch.float32)
            self.planes = None
            off, rows = 0, [0]
            for p, cp, n in zip(plans, conv_params, sizes):
                p.ensure(cp, n_convs=n_convs, arena=self.arena[off: off + n], shared=self)
                off += n
                base = rows[-1]
                rows += [base + r for r in p.rows[1:]]
            self.desc = torch.cat([p.desc for p in plans]).contiguous()
            self.prefix = torch.tensor(rows, dtype=torch.int32).to(self.arena.device)
            self.n_conv, self.total_rows, self.keys = self.desc.shape[0], rows[-1], keys
        for p in plans:
            p.pack_count = getattr(p, "pack_count", 0) + 1
        if _SPLIT_MATH[0] and plans[0].want_planes:
            call("glowtts_pack_weight_planes_multi", ptr(self.desc), ptr(self.prefix), self.n_conv, self.total_rows, ptr(self.arena),
                 self.arena.numel(), ptr(self.planes_buffer(True)))
            if _WINO and _hip.get_knob("WINO"):
                self.wino_weights(plans)
        else:
            call("glowtts_pack_weight_multi", ptr(self.desc), ptr(self.prefix), self.n_conv, self.total_rows)

    def wino_weights(self, plans):
        """Winograd-domain planes (csrc/convwino.hip) of every gated 5-tap in-conv of the stack: one launch behind the weight pack."""
        _wino_update(self, self.arena, plans)


def _wino_update(owner, arena, plans):
    """(Re)make the Winograd-domain planes of the gated 5-tap in-convs packed in `arena` (a flow stack's buffer or one block's own):
    table of (offset, Cin / 16, M) rows built once per buffer, one `glowtts_wino_weights` launch per packing."""
    if getattr(owner, "wino_key", None) is not arena:
        rows = []
        for p in plans:
            for (_v, _g, wp_f, _b, _inv, cout, cin, taps, _off) in p.convs:
                if taps == 5 and cout == 2 * cin and cin % 64 == 0:
                    rows.append([(wp_f.data_ptr() - arena.data_ptr()) // 4, cin // 16, cout])
        owner.wino_table = torch.tensor(rows, dtype=torch.int64).to(arena.device) if rows else None
        owner.wino_planes = (torch.zeros(3 * _hip.wino_plane_elems(arena.numel()), device=arena.device, dtype=torch.int16)
                             if rows else None)
        owner.wino_key = arena
    if owner.wino_table is not None:
        call("glowtts_wino_weights", ptr(arena), arena.numel(), ptr(owner.wino_table), owner.wino_table.shape[0],
             ptr(owner.wino_planes), owner.wino_planes.numel() // 3)


# The gated 5-tap in-conv of the flow stack in its Winograd F(4, 5) form (csrc/convwino.hip): 35.4 us against 48.4 for the
# direct bf16x6 kernel alone at the benchmark's shape, 13.60 -> 13.41 ms per step with the two-chain forward (13.87 -> 13.48 with one
# chain).  Needs the stack's one packed buffer (FlowStackFn) and the bf16x6 arithmetic; GLOWTTS_WINO=0 keeps the direct kernels.
_WINO = os.environ.get("GLOWTTS_WINO", "1") != "0"

# FlowStackFn: one weight-pack launch and one W^-1 / log det W launch for the whole stack (0: one of each per block)
_STACK_PACK = os.environ.get("GLOWTTS_STACK_PACK", "1") != "0"


# WN stack executor (csrc/wn_stack.hip queues a stack's whole launch sequence from C): "both" (default) = forward and
# backward; "fwd" = native forward, backward driven layer by layer from Python; "off".  The step's host time matters: with
# ~1 000 launches it is within 20 % of the GPU time, and past it on a slower host or with the faster bf16-plane kernels.
_WN_NATIVE = os.environ.get("GLOWTTS_WN_NATIVE", "both")

_active_groups: List = []


class ConvGroup:
    """The plain (non-WN) convolutions of one module — a coupling block's start / end convs, the whole text encoder —
    share ONE weight-norm + packing launch per forward and ONE un-packing launch per backward (each conv otherwise costs
    a pack and an unpack launch of its own: ~140 launches per step at config 2).

    `begin()` at the top of the owner's forward packs everything; `conv1d` / `Conv1dFn` then take the packed weights from
    the group, accumulate the packed weight gradient into the group's buffer and count down; the last backward of the
    group un-packs into `param.grad` and announces the gradients.  Needs every gradient to exist already (the flat-buffer
    optimizer keeps them allocated); otherwise the group stays inactive and every conv packs for itself."""

    def __init__(self, convs, planes: bool = False):
        """`planes`: also keep bf16 planes of the packed weights (one split launch per pack) so that convolutions of this group
        with a bf16-plane kernel (round 3: the text encoder's 3-tap FFN convolutions) run in the selected conv arithmetic —
        whoever launches them binds the planes around the launch (`self.plan.bind()` / `WNPackPlan.unbind`)."""
        self.modules = list(convs)
        self.plan = WNPackPlan(want_planes=planes)
        self.index = {}
        self.active = False
        self.pending = 0
        self.touched = False
        for i, m in enumerate(self.modules):
            m._glowtts_group = (self, i)

    def _params(self):
        out = []
        for m in self.modules:
            if hasattr(m, "weight_v"):
                out += [m.weight_v, m.weight_g, m.bias]
            else:
                out += [m.weight, None, m.bias]
        return out

    def begin(self) -> None:
        params = self._params()
        tensors = [p for p in params if p is not None]
        self.active = all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in tensors)
        if self.active and torch.is_grad_enabled() and any(p.requires_grad for p in tensors):
            self.active = direct_grads_enabled() and all((not p.requires_grad) or (p.grad is not None and p.grad.is_contiguous()
                                                        and p.grad.dtype == torch.float32) for p in tensors)
        _mark_direct(tensors, self.active and torch.is_grad_enabled())
        if not self.active:
            return
        if self.pending > 0 and self.touched:            # a previous backward never completed: do not lose its gradients
            self.flush()
        self.params = params
        self.plan.ensure(params, n_convs=len(self.modules))
        self.plan.pack()
        self.pending, self.touched = 0, False

    def packed(self, i):
        return self.plan.convs[i][2:5]                   # wp_f, wp_b, inv

    def dwp(self, i):
        if not self.touched:                             # first weight gradient of this backward: clear the accumulator
            self.plan.dwp.zero_()
            self.touched = True
            if self not in _active_groups:
                _active_groups.append(self)
        return self.plan.dwp_view(i)

    def finish_one(self) -> None:
        self.pending -= 1
        if self.pending <= 0 and self.touched:
            # last backward of the group: un-pack behind the weight-gradient stream's queue instead of on the dx chain
            # (the stream first waits for the current one, i.e. for this group's own wrw launches)
            _WgradStream(self.plan.dwp.device).run(self.flush)

    def flush(self) -> None:
        self.plan.unpack_into_grads(self.params)
        _notify([p for j, p in enumerate(self.params) if p is not None and j % 3 != 2])
        self.pending, self.touched = 0, False
        if self in _active_groups:
            _active_groups.remove(self)


def flush_groups() -> None:
    """Un-pack whatever weight gradients are still waiting in a ConvGroup (a conv whose backward never ran leaves its
    group counting); `train.train_batch` calls this after `loss.backward()`."""
    for g in list(_active_groups):
        g.flush()


def _wn_layers_from_slabs(x, xs, acts, ts, n_layers):
    """Per-layer (x_i, acts_i, ts_i) views of the native executor's slabs, in the order the Python backward expects."""
    out = []
    for i in range(n_layers):
        out += [x if i == 0 else xs[i - 1], acts[i], ts[i]]
    return out


class WNFn(Function):
    """The whole WN stack (reference layers.py:138-162) as one autograd node.

    forward: weight norm + packing of every conv (one launch), dropout keep-masks of every layer (one generator call),
             then per layer i:  acts, ts = conv_gate(x_i)                 [k-tap dilated conv + bias + dropout + cond + gate]
                                x_{i+1}, skip = conv_res_skip(acts, x_i)  [1x1 conv + bias + residual*mask + skip accumulate]
    backward, per layer (reverse): d_rs = [dx_{i+1} mask ; dskip] -> weight grads (wrw, bias sums ride along)
             -> d_acts = W_rs^T d_rs -> gate backward from the saved tanh/sigmoid -> weight grads
             -> dx_i = d_rs[:H] + W_in^T (*) d_xin ; finally every packed gradient goes through the weight norm into
             param.grad in one launch.
    Saved per layer: x_i (H), acts (H), ts (2H) [+ dropout bytes].
    """

    @staticmethod
    def forward(ctx, x, m2, cond, p_drop, dil_rate, n_layers, plan, drop_pre, *params):
        # params: per layer (in_v, in_g, in_b, rs_v, rs_g, rs_b); plan: the module's WNPackPlan
        plan.ensure(params, n_layers)
        plan.pack()
        bound = plan.bind()
        try:
            return WNFn._forward(ctx, x, m2, cond, p_drop, dil_rate, n_layers, plan, drop_pre, *params)
        finally:
            plan.unbind(bound)

    @staticmethod
    def _forward(ctx, x, m2, cond, p_drop, dil_rate, n_layers, plan, drop_pre, *params):
        x = f32(x.contiguous())
        B, H, T = x.shape
        dev = x.device
        saved = []
        skip = None
        cur = x
        drop_all = None
        if p_drop > 0.0:
            if drop_pre is not None and tuple(drop_pre.shape) == (n_layers, B, 2 * H, T) and drop_pre.is_contiguous():
                drop_all = drop_pre                     # drawn by the caller (one generator launch for all coupling blocks)
            else:
                drop_all = ops.keep_mask((n_layers, B, 2 * H, T), p_drop, dev, "wn")   # keep = 1
        ctx.native = False
        if cond is None and _hip.timing_off() and _WN_NATIVE != "off" and all(params[3 * j + 2] is not None for j in range(2 * n_layers)):
            # the whole launch sequence of the stack in one native call (csrc/wn_stack.hip): four allocations, one ctypes
            # call, instead of two launches + four allocations per layer driven from here
            import ctypes
            taps = params[0].shape[2]
            acts = torch.empty(n_layers, B, H, T, device=dev, dtype=torch.float32)
            ts = torch.empty(n_layers, B, 2 * H, T, device=dev, dtype=torch.float32)
            xs = torch.empty(n_layers - 1, B, H, T, device=dev, dtype=torch.float32) if n_layers > 1 else None
            skip = torch.empty(B, H, T, device=dev, dtype=torch.float32)
            tab = plan.layer_table(params, n_layers)
            call("glowtts_wn_fwd", ctypes.addressof(tab), n_layers, ptr(x), ptr(m2), ptr(drop_all),
                 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0, ptr(xs), ptr(acts), ptr(ts), ptr(skip), B, H, T, taps, dil_rate)
            ctx.save_for_backward(m2, x, acts, ts, *([] if xs is None else [xs]), *([] if drop_all is None else [drop_all]))
            ctx.params, ctx.plan, ctx.native = params, plan, True
            ctx.cfg = (n_layers, dil_rate, float(p_drop), False, B, H, T)
            return skip
        for i in range(n_layers):
            in_v, in_g, in_b, rs_v, rs_g, rs_b = params[6 * i: 6 * i + 6]
            taps = in_v.shape[2]
            dil = dil_rate ** i
            pad = (taps * dil - dil) // 2
            wf_in, wf_rs = plan.convs[2 * i][2], plan.convs[2 * i + 1][2]
            drop = None if drop_all is None else drop_all[i]
            acts = torch.empty(B, H, T, device=dev, dtype=torch.float32)
            ts = torch.empty(B, 2 * H, T, device=dev, dtype=torch.float32)
            c_i = None if cond is None else f32(cond[:, 2 * H * i: 2 * H * (i + 1)].reshape(B, 2 * H).contiguous())
            call("glowtts_conv_gate_fwd", ptr(cur), ptr(wf_in), ptr(f32(in_b.detach().contiguous())), ptr(c_i), ptr(drop),
                 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0, ptr(acts), ptr(ts), B, H, T, taps, dil, pad,
                 tag=("M%d K%dx%d N%dx%d", 2 * H, H, taps, B, T))
            last = i == n_layers - 1
            skip_out = torch.empty(B, H, T, device=dev, dtype=torch.float32)
            nxt = None if last else torch.empty(B, H, T, device=dev, dtype=torch.float32)
            call("glowtts_conv_res_skip_fwd", ptr(acts), ptr(wf_rs), ptr(f32(rs_b.detach().contiguous())), ptr(m2),
                 None if last else ptr(cur), ptr(skip), None if last else ptr(nxt), ptr(skip_out), B, H, T, int(last),
                 tag=("M%d K%dx1 N%dx%d", H if last else 2 * H, H, B, T))
            saved += [cur, acts, ts]
            skip = skip_out
            if not last:
                cur = nxt
        ctx.save_for_backward(m2, *saved, *([] if drop_all is None else [drop_all]))
        ctx.params = params
        ctx.plan = plan
        ctx.cfg = (n_layers, dil_rate, float(p_drop), cond is not None, B, H, T)
        return skip

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        n_layers, dil_rate, p_drop, has_cond, B, H, T = ctx.cfg
        sv = ctx.saved_tensors
        m2 = sv[0]
        bound = ctx.plan.bind()              # (the backward runs on autograd's thread: the binding is per thread)
        try:
            if ctx.native:
                return WNFn._backward_native(ctx, dout)
            saved = sv[1: 1 + 3 * n_layers]
            drops = list(sv[1 + 3 * n_layers]) if p_drop > 0 else [None] * n_layers
            return WNFn._backward_layers(ctx, dout, m2, saved, drops)
        finally:
            ctx.plan.unbind(bound)

    @staticmethod
    def _backward_layers(ctx, dout, m2, saved, drops):
        n_layers, dil_rate, p_drop, has_cond, B, H, T = ctx.cfg
        params, plan = ctx.params, ctx.plan
        dev = dout.device
        sink = _GradSink(params)
        dskip = dout.contiguous()
        dx_next = None
        dconds = [None] * n_layers
        scale = 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0
        plan.dwp.zero_()                                # ONE memset for every packed weight gradient of the stack
        wgrad = _WgradStream(dev)
        # virtual concat of d_rs (no res_skip_bwd launch, no d_rs tensor): needs the frame-packed wrw and whole 96-channel
        # chunks per source, gradients written in place, no conditioning input and unit dilation
        two_src = (sink.direct and not has_cond and dil_rate == 1 and H % 192 == 0 and T % 4 == 0 and _hip.timing_off()
                   and all(params[6 * j + 5] is not None for j in range(n_layers)))
        for i in reversed(range(n_layers)):
            in_v, in_g, in_b, rs_v, rs_g, rs_b = params[6 * i: 6 * i + 6]
            x_i, acts, ts = saved[3 * i: 3 * i + 3]
            wb_in, inv_in = plan.convs[2 * i][3:5]
            wb_rs, inv_rs = plan.convs[2 * i + 1][3:5]
            taps = in_v.shape[2]
            dil = dil_rate ** i
            pad = (taps * dil - dil) // 2
            last = i == n_layers - 1
            m_rs = H if last else 2 * H
            d_xin = torch.empty(B, 2 * H, T, device=dev, dtype=torch.float32)
            if two_src and not last:
                # d_rs = [dx_{i+1} mask ; dskip] is never written: dx_{i+1} left the layer above already masked, and the two
                # kernels that read d_rs take its halves from the two tensors (conv_gate_bwd / conv_wrw2)
                half = dx_next

                def wg_rs(half=half, dskip=dskip, acts=acts, i=i):
                    call("glowtts_conv_wrw2", ptr(acts), acts.stride(0), ptr(half), H * T, ptr(dskip), H * T, H,
                         ptr(plan.dwp_view(2 * i + 1)), ptr(sink.buf(6 * i + 5)), B, H, 2 * H, T, 1, 1, 0,
                         tag=("M%d K%dx1 N%dx%d", 2 * H, H, B, T))
                wgrad.run(wg_rs, half, dskip, acts)
                call("glowtts_conv_gate_bwd", ptr(half), ptr(dskip), ptr(wb_rs), ptr(ts), ptr(drops[i]), scale, ptr(d_xin), B,
                     2 * H, H, T, tag=("M%d K%dx1 N%dx%d", H, 2 * H, B, T))
                wgrad.run(lambda: _weight_grads(x_i, d_xin, None, (taps, H, 2 * H), in_v, in_g, inv_in, sink.buf(6 * i),
                                                sink.buf(6 * i + 1), sink.buf(6 * i + 2), taps, dil, pad,
                                                dwp=plan.dwp_view(2 * i), unpack=False), d_xin, x_i)
                dx = torch.empty(B, H, T, device=dev, dtype=torch.float32)
                # dx_i = half (residual path) + W_in^T (*) d_xin, masked on the way out unless it is the stack's own input grad
                conv_fwd(d_xin, wb_in, None, m2 if i > 0 else None, dx, 2 * H, H, taps, dil, (taps - 1) * dil - pad,
                         addend=half, mask_out=i > 0)
                dx_next = dx
                continue
            d_rs = torch.empty(B, m_rs, T, device=dev, dtype=torch.float32)
            call("glowtts_res_skip_bwd", None if last else ptr(dx_next), ptr(dskip), ptr(m2), None, ptr(d_rs), B, H, T, int(last))
            if last:
                dskip = d_rs                          # d(skip_in) of the last layer carries the folded mask
            wgrad.run(lambda: _weight_grads(acts, d_rs, None, (1, H, m_rs), rs_v, rs_g, inv_rs, sink.buf(6 * i + 3),
                                            sink.buf(6 * i + 4), sink.buf(6 * i + 5), 1, 1, 0, dwp=plan.dwp_view(2 * i + 1),
                                            unpack=not sink.direct), d_rs, acts)
            if has_cond and drops[i] is not None:
                # conditioning is added after the dropout: its gradient is the un-dropped pre-activation gradient, so
                # d(acts) is needed twice and is materialised
                d_acts = torch.empty(B, H, T, device=dev, dtype=torch.float32)
                conv_fwd(d_rs, wb_rs, None, None, d_acts, m_rs, H, 1, 1, 0)
                call("glowtts_gate_bwd_ts", ptr(ts), ptr(d_acts), ptr(drops[i]), scale, ptr(d_xin), B, H, T)
                tmp = torch.empty_like(d_xin)
                call("glowtts_gate_bwd_ts", ptr(ts), ptr(d_acts), None, 1.0, ptr(tmp), B, H, T)
                dconds[i] = tmp.sum(-1)
            else:
                # d(pre-activation) = gate'(stored tanh / sigmoid) * (W_rs^T d_rs): one kernel, d(acts) stays on chip
                call("glowtts_conv_gate_bwd", ptr(d_rs), None, ptr(wb_rs), ptr(ts), ptr(drops[i]), scale, ptr(d_xin), B, m_rs, H, T,
                     tag=("M%d K%dx1 N%dx%d", H, m_rs, B, T))
                if has_cond:
                    dconds[i] = d_xin.sum(-1)
            wgrad.run(lambda: _weight_grads(x_i, d_xin, None, (taps, H, 2 * H), in_v, in_g, inv_in, sink.buf(6 * i),
                                            sink.buf(6 * i + 1), sink.buf(6 * i + 2), taps, dil, pad, dwp=plan.dwp_view(2 * i),
                                            unpack=not sink.direct), d_xin, x_i)
            dx = torch.empty(B, H, T, device=dev, dtype=torch.float32)
            # dx_i = (residual path) d_rs[:, :H] + (conv path) W_in^T (*) d_xin ; the last layer has no residual path.  With
            # the two-source scheme the layer below reads dx_i as the first half of ITS d_rs, so the mask goes on here.
            mask_here = two_src and i > 0
            conv_fwd(d_xin, wb_in, None, m2 if mask_here else None, dx, 2 * H, H, taps, dil, (taps - 1) * dil - pad,
                     addend=None if last else d_rs[:, :H], mask_out=mask_here)
            dx_next = dx
        dcond = None
        if has_cond:
            dcond = torch.cat(dconds, 1).unsqueeze(-1)
        if sink.direct:
            # every un-packing (through the weight norm) in one launch, queued BEHIND the weight-gradient kernels on their
            # stream: the main stream does not wait here (train_batch joins the side streams after backward; a gradient
            # listener — the DP reducer — is told inside the side stream's context, so its collective waits on that stream)
            results = []
            wgrad.run(lambda: (plan.unpack_into_grads(params), results.extend(sink.results())))
        else:
            wgrad.join()
            results = sink.results()
        return (dx_next, None, dcond, None, None, None, None, None, *results)

    @staticmethod
    def _backward_native(ctx, dout):
        import ctypes
        n_layers, dil_rate, p_drop, _, B, H, T = ctx.cfg
        sv = list(ctx.saved_tensors)
        m2, x, acts, ts = sv[:4]
        rest = sv[4:]
        xs = rest.pop(0) if n_layers > 1 else None
        drop_all = rest.pop(0) if p_drop > 0 else None
        params, plan = ctx.params, ctx.plan
        dev = dout.device
        sink = _GradSink(params)
        if not sink.direct or _WN_NATIVE != "both":
            # The backward driven layer by layer from here (GLOWTTS_WN_NATIVE=fwd, or gradients that do not exist yet).  The
            # native backward issues the same launch sequence from C and saves ~3.5 ms of host time per step.  (Its first
            # form deferred a block's weight gradients behind its whole dx chain and materialised d_rs: 200 MB of workspace
            # per block fell out of the 256 MB Infinity Cache and cost 0.5-1.1 ms of GPU time; the two-source, per-layer
            # form matches this path's GPU time.)
            ctx.native = False
            saved = _wn_layers_from_slabs(x, xs, acts, ts, n_layers)
            return WNFn._backward_layers(ctx, dout, m2, saved, [None] * n_layers if drop_all is None else list(drop_all))
        taps = params[0].shape[2]
        dskip = dout.contiguous()
        wgrad = _WgradStream(dev)
        # two-source form (the launch sequence of _backward_layers): no d_rs tensor, weight gradients interleaved per layer
        two_src = wgrad.enabled and dil_rate == 1 and H % 192 == 0 and T % 4 == 0
        d_rs = torch.empty((B, H, T) if two_src else (n_layers, B, 2 * H, T), device=dev, dtype=torch.float32)
        d_xin = torch.empty(n_layers, B, 2 * H, T, device=dev, dtype=torch.float32)
        dx = torch.empty(n_layers, B, H, T, device=dev, dtype=torch.float32)
        plan.dwp.zero_()
        tab = plan.layer_table(params, n_layers)
        gdesc, prefix = plan.unpack_tables(params)
        call("glowtts_wn_bwd", ctypes.addressof(tab), n_layers, ptr(x), ptr(xs), ptr(acts), ptr(ts), ptr(m2), ptr(drop_all),
             1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0, ptr(dskip), ptr(d_rs), ptr(d_xin), ptr(dx), ptr(gdesc), ptr(prefix),
             len(plan.convs), plan.total_rows, B, H, T, taps, dil_rate, int(two_src),
             wgrad.side.cuda_stream if wgrad.enabled else None)
        if wgrad.enabled:
            for t in (x, xs, acts, d_rs, d_xin, dx, dskip, m2):   # read by the weight-gradient stream after this call returns (m2: see FlowBlockFn)
                if t is not None:
                    t.record_stream(wgrad.side)
            with torch.cuda.stream(wgrad.side):         # listeners (the DP reducer) must wait on the stream that un-packs
                results = sink.results()
        else:
            results = sink.results()
        return (dx[0], None, None, None, None, None, None, None, *results)


# ----------------------------------------------------------------------------------------------------------------
class FlowBlockPlan:
    """Packed weights, gradient accumulators and the host-side `glowtts_flow_block` table of one decoder block
    ([ActNorm, InvConvNear, CouplingBlock]): its 2 + 2 n_layers convolutions share ONE pack launch per forward and ONE
    un-pack launch per backward (the per-op path used two of each: the coupling's ConvGroup and the WN stack's plan)."""

    def __init__(self):
        self.plan = WNPackPlan(want_planes=True, wino=True)
        self._key = None

    def bind(self, io: bool):
        """Planes for the calling thread's next convolution launches.  bf16 tensors (io): the three bf16 planes h + m + l of
        the freshly packed weights, made here (one elementwise launch over the block's packed buffer) — plane h alone is the
        weight rounded to bf16, which is what every convolution but the end conv reads; else whatever glowtts_conv_math asks."""
        plan = self.plan
        if not io:
            return ("math", plan.bind())
        n = plan.wp_arena.numel()
        if getattr(self, "_planes_io", None) is None or self._planes_io.numel() != 3 * n:
            self._planes_io = torch.empty(3 * n, device=plan.wp_arena.device, dtype=torch.int16)
        if getattr(self, "_planes_io_version", None) != (plan.version, plan.pack_count):
            call("glowtts_split_planes", ptr(plan.wp_arena), n, ptr(self._planes_io), 3)
            self._planes_io_version = (plan.version, plan.pack_count)
        _hip.conv_bind_planes_ns(plan.wp_arena, self._planes_io, 3)
        return ("io", True)

    def unbind(self, bound):
        if bound[0] == "io":
            _hip.conv_bind_planes_ns(None)
        else:
            self.plan.unbind(bound[1])

    @staticmethod
    def conv_params(params, n_layers):
        """(v, g, bias) triples in the plan's order: start, end, then (in, res/skip) per WN layer."""
        logs, bias, w, sv, sg, sb, ev, eb = params[:8]
        return [sv, sg, sb, ev, None, eb] + list(params[8: 8 + 6 * n_layers])

    def table(self, params, n_layers):
        import ctypes
        plan = self.plan
        convp = self.conv_params(params, n_layers)
        key = (plan.version, tuple(0 if p is None else p.data_ptr() for p in params),
               tuple(0 if (p is None or p.grad is None) else p.grad.data_ptr() for p in params))
        if key != self._key:
            logs, bias, w = params[:3]
            gdesc, prefix = plan.unpack_tables(convp)
            layers = (_hip.WnLayer * n_layers)()
            for i in range(n_layers):
                ci, cr = plan.convs[2 + 2 * i], plan.convs[3 + 2 * i]
                in_b, rs_b = convp[6 + 6 * i + 2], convp[6 + 6 * i + 5]
                L = layers[i]
                L.wf_in, L.wb_in, L.b_in = ci[2].data_ptr(), ci[3].data_ptr(), in_b.data_ptr()
                L.wf_rs, L.wb_rs, L.b_rs = cr[2].data_ptr(), cr[3].data_ptr(), rs_b.data_ptr()
                L.dwp_in, L.dwp_rs = plan.dwp_view(2 + 2 * i).data_ptr(), plan.dwp_view(3 + 2 * i).data_ptr()
                L.db_in, L.db_rs = in_b.grad.data_ptr(), rs_b.grad.data_ptr()
            t = _hip.FlowBlock()
            t.logs, t.bias, t.w = logs.data_ptr(), bias.data_ptr(), w.data_ptr()
            cs, ce = plan.convs[0], plan.convs[1]
            t.wf_start, t.wb_start, t.b_start = cs[2].data_ptr(), cs[3].data_ptr(), convp[2].data_ptr()
            t.wf_end, t.wb_end, t.b_end = ce[2].data_ptr(), ce[3].data_ptr(), convp[5].data_ptr()
            t.dwp_start, t.dwp_end = plan.dwp_view(0).data_ptr(), plan.dwp_view(1).data_ptr()
            t.db_start, t.db_end = convp[2].grad.data_ptr(), convp[5].grad.data_ptr()
            t.dlogs, t.dbias, t.dw = logs.grad.data_ptr(), bias.grad.data_ptr(), w.grad.data_ptr()
            t.layers = ctypes.addressof(layers)
            t.pack_desc = None                     # packing (and the optional bf16-plane split) is launched by plan.pack()
            t.unpack_desc, t.pack_prefix = gdesc.data_ptr(), prefix.data_ptr()
            t.dwp_all, t.dwp_floats = plan.dwp.data_ptr(), plan.dwp.numel()
            t.n_layers, t.n_conv, t.total_rows = n_layers, len(plan.convs), plan.total_rows
            self._tab, self._layers, self._key = t, layers, key       # (the layer array must outlive the struct)
        return self._tab


def _flow_block_cache(actnorm, invconv, coupling) -> dict:
    """Per-block host cache (on the coupling module): the block's parameter list in the executor's order and the checks that
    depend only on the modules' structure.  ~60 attribute look-ups through nn.Module.__getattr__ per block and call otherwise
    (0.7 ms of host time per step at config 2).  Revalidated by identity of four parameters from its two ends and the middle
    (weight norm removed by store_inverse, layers replaced, a module rebuilt: all change at least one of them)."""
    c = getattr(coupling, "_fb_cache", None)
    wn = coupling.wn
    if c is not None:
        ps = c["params"]
        if (len(wn.in_layers) == c["n_layers"] and actnorm._parameters.get("logs") is ps[0]
                and invconv._parameters.get("weight") is ps[2] and coupling.start._parameters.get("weight_v") is ps[3]
                and wn.res_skip_layers[-1]._parameters.get("bias") is ps[-1]
                and wn.in_layers[-1]._parameters.get("weight_v") is ps[-6]):
            return c
    convs = [coupling.start, coupling.end] + list(wn.in_layers) + list(wn.res_skip_layers)
    static_ok = (hasattr(coupling.start, "weight_v") and not hasattr(coupling.end, "weight_v")
                 and not any(c_.bias is None for c_ in convs) and all(hasattr(c_, "weight_v") for c_ in convs[2:]))
    params = None
    if static_ok:
        params = [actnorm.logs, actnorm.bias, invconv.weight, coupling.start.weight_v, coupling.start.weight_g,
                  coupling.start.bias, coupling.end.weight, coupling.end.bias]
        for in_layer, rs_layer in zip(wn.in_layers, wn.res_skip_layers):
            params += [in_layer.weight_v, in_layer.weight_g, in_layer.bias, rs_layer.weight_v, rs_layer.weight_g, rs_layer.bias]
    c = {"params": params, "static_ok": static_ok, "n_layers": len(wn.in_layers), "gkey_ok": None}
    if static_ok:
        coupling._fb_cache = c
    return c


def _grad_key(params):
    """(gradient address, requires_grad) of every parameter, 0 for a missing gradient: what the executors' tables depend on."""
    return tuple([(0 if p.grad is None else p.grad.data_ptr(), p.requires_grad) for p in params])


def flow_block_eligible(actnorm, invconv, coupling, x, g) -> bool:
    """Can [actnorm, invconv, coupling] run as ONE native call each way (FlowBlockFn)?  Training direction, no conditioning
    input, fused-flow sizes, every parameter gradient already allocated (the flat-buffer optimizer) and written in place."""
    if not x.is_cuda or not torch.is_grad_enabled() or not _hip.timing_off() or _WN_NATIVE != "both":
        return False
    if not actnorm.initialized or invconv.no_jacobian or invconv.n_split not in (2, 4):
        return False
    if (g is not None) != (coupling.gin_channels != 0):      # speaker conditioning: the block takes the cond rows as an input
        return False
    if not direct_grads_enabled():
        return False
    c = _flow_block_cache(actnorm, invconv, coupling)
    if not c["static_ok"]:
        return False
    # gradients are accumulated in place: every buffer must exist already (the flat-buffer optimizer keeps them allocated);
    # the full check runs when a gradient has moved, otherwise one pass over the addresses
    key = _grad_key(c["params"])
    if key == c["gkey_ok"]:
        return True
    ok = all(p.requires_grad and p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32
             for p in c["params"])
    c["gkey_ok"] = key if ok else None
    return ok


def flow_block_bf16_ok(coupling, x_shape) -> bool:
    """Shapes the bf16-tensor kernels are instantiated for (csrc/convgemm_split.hip, dispatch_bf16_io): the default coupling
    network (5 taps, dilation 1, hidden width 192) on a flow tensor whose halves are whole 16-channel groups, T % 4 == 0."""
    wn = coupling.wn
    B, C, T = x_shape
    return (wn.kernel_size[0] == 5 and wn.dilation_rate == 1 and wn.hidden_channels == 192 and C % 32 == 0 and T % 4 == 0
            and T > 0)


def flow_block_params(actnorm, invconv, coupling):
    c = _flow_block_cache(actnorm, invconv, coupling)
    if c["params"] is not None:
        return c["params"]
    wn = coupling.wn
    out = [actnorm.logs, actnorm.bias, invconv.weight, coupling.start.weight_v, coupling.start.weight_g, coupling.start.bias,
           coupling.end.weight, coupling.end.bias]
    for in_layer, rs_layer in zip(wn.in_layers, wn.res_skip_layers):
        out += [in_layer.weight_v, in_layer.weight_g, in_layer.bias, rs_layer.weight_v, rs_layer.weight_g, rs_layer.bias]
    return out


class FlowBlockFn(Function):
    """One decoder block — ActNorm, InvConvNear, CouplingBlock (reference models.py:176-190; layers.py:182-199, 238-272;
    attentions.py:119-142) — as ONE autograd node whose forward and backward are one native call each
    (csrc/wn_stack.hip: glowtts_flow_block_fwd / _bwd queue the block's whole launch sequence from C)."""

    @staticmethod
    def forward(ctx, x, m2, x_len, drop, cfg, bplan, cond, *params):
        # cond: None or the speaker conditioning rows (B, 2H * n_layers, 1) = wn.cond_layer(g) (reference layers.py:142-150)
        import ctypes
        n_split, sigmoid_scale, p_drop, dil_rate, n_layers, H, io = cfg[:7]
        # io (csrc: the `_io` entry points): 0 = fp32 tensors; 1 = the coupling network's hidden tensors bf16 in HBM, the
        # flow tensor fp32; 3 = the flow tensor bf16 as well
        fdt = torch.bfloat16 if io & 2 else torch.float32
        adt = torch.bfloat16 if io & 1 else torch.float32
        x = x.contiguous()
        if x.dtype != fdt:
            raise RuntimeError(f"FlowBlockFn: flow tensor is {x.dtype}, the block runs with {fdt} flow tensors")
        B, C, T = x.shape
        dev = x.device
        plan = bplan.plan
        plan.ensure(FlowBlockPlan.conv_params(params, n_layers), n_convs=2 + 2 * n_layers)
        plan.pack()
        bound = bplan.bind(io)
        try:
            new = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)                      # noqa: E731
            act = lambda *shape: torch.empty(shape, device=dev, dtype=adt)                                # noqa: E731
            y, z = torch.empty(B, C, T, device=dev, dtype=fdt), torch.empty(B, C, T, device=dev, dtype=fdt)
            out = new(B, C, T)
            y0h = act(B, C // 2, T) if io == 1 else None      # bf16 copy of y's first half for the start conv
            h0, skip = act(B, H, T), act(B, H, T)
            acts, ts = act(n_layers, B, H, T), act(n_layers, B, 2 * H, T)
            xs = act(n_layers - 1, B, H, T) if n_layers > 1 else None
            logdet = new(B)
            winv = new(n_split * n_split + 1)
            if p_drop > 0.0 and (drop is None or tuple(drop.shape) != (n_layers, B, 2 * H, T) or not drop.is_contiguous()):
                drop = ops.keep_mask((n_layers, B, 2 * H, T), p_drop, dev, f"decoder.block.{cfg[7] if len(cfg) > 7 else 0}")
            if p_drop <= 0.0:
                drop = None
            tab = bplan.table(params, n_layers)
            tab.w_inv, tab.logdet_w = winv.data_ptr(), winv.data_ptr() + 4 * n_split * n_split
            taps = params[8].shape[2]
            scale = 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0
            cond_l = None
            if cond is not None:                         # layer-major rows (n_layers, B, 2H) for the gate kernels
                cond_l = f32(cond.detach()).reshape(B, n_layers, 2 * H).permute(1, 0, 2).contiguous()
            call("glowtts_flow_block_fwd_io", ctypes.addressof(tab), ptr(x), ptr(m2), ptr(x_len), ptr(cond_l), ptr(drop), scale, ptr(y), ptr(y0h),
                 ptr(h0), ptr(xs), ptr(acts), ptr(ts), ptr(skip), ptr(out), ptr(z), ptr(logdet), B, C, H, T, taps, dil_rate, n_split,
                 int(sigmoid_scale), int(io))
        finally:
            bplan.unbind(bound)
        ctx.save_for_backward(x, m2, x_len, y, h0, acts, ts, skip, out, winv, *([] if xs is None else [xs]),
                              *([] if drop is None else [drop]), *([] if y0h is None else [y0h]))
        ctx.cfg, ctx.bplan, ctx.params, ctx.taps, ctx.scale = cfg, bplan, params, taps, scale
        ctx.cond_shape = None if cond is None else tuple(cond.shape)
        return z, logdet

    @staticmethod
    @once_differentiable
    def backward(ctx, dz, dlogdet):
        import ctypes
        n_split, sigmoid_scale, p_drop, dil_rate, n_layers, H, io = ctx.cfg[:7]
        block_index = ctx.cfg[7] if len(ctx.cfg) > 7 else 1 << 30
        fdt = torch.bfloat16 if io & 2 else torch.float32
        adt = torch.bfloat16 if io & 1 else torch.float32
        sv = list(ctx.saved_tensors)
        x, m2, x_len, y, h0, acts, ts, skip, out, winv = sv[:10]
        rest = sv[10:]
        xs = rest.pop(0) if n_layers > 1 else None
        drop = rest.pop(0) if p_drop > 0 else None
        y0h = rest.pop(0) if io == 1 else None
        params, bplan = ctx.params, ctx.bplan
        plan = bplan.plan
        B, C, T = x.shape
        dev = x.device
        if not all(p.grad is not None and p.grad.is_contiguous() for p in params):
            raise RuntimeError("FlowBlockFn.backward: a parameter gradient buffer disappeared between forward and backward")
        new = lambda *shape: torch.empty(shape, device=dev, dtype=adt)                                    # noqa: E731
        flow = lambda *shape: torch.empty(shape, device=dev, dtype=fdt)                                   # noqa: E731
        dz = dz.contiguous() if dz is not None else torch.zeros_like(x)
        if dz.dtype != fdt:
            dz = dz.to(fdt)
        dlogdet = dlogdet.contiguous().float() if dlogdet is not None else torch.zeros(B, device=dev)
        wgrad = _WgradStream(dev)
        if block_index < _WGRAD_MAIN_BLOCKS:
            wgrad.enabled = False                    # the backward's last blocks: their weight gradients stay on the chain's stream
        # the two-source launch sequence (d_rs never materialised, batched weight gradients) also where the weight gradients stay
        # on the chain's stream — the backward's last block, and every block of a one-stream run (rocprofv3 passes then time the
        # launches the step really makes): 14.83 -> 14.75 ms per step (tools/ab_flags.py, four alternating pairs)
        two_src = (not io) and dil_rate == 1 and H % 192 == 0 and T % 4 == 0
        dy, dout, dx = flow(B, C, T), new(B, C, T), flow(B, C, T)
        dskip = new(B, H, T)
        d_rs = new(B, H, T) if two_src else new(n_layers, B, 2 * H, T)
        d_xin, dx_wn = new(n_layers, B, 2 * H, T), new(n_layers, B, H, T)
        tab = bplan.table(params, n_layers)
        tab.w_inv, tab.logdet_w = winv.data_ptr(), winv.data_ptr() + 4 * n_split * n_split
        dcond_l = None
        if ctx.cond_shape is not None:
            dcond_l = scratch_zeros((n_layers, B, 2 * H), dev)
        bound = bplan.bind(io)
        try:
            call("glowtts_flow_block_bwd_io", ctypes.addressof(tab), ptr(x), ptr(m2), ptr(x_len), ptr(drop), ctx.scale, ptr(y),
                 ptr(y0h), ptr(h0), ptr(xs), ptr(acts), ptr(ts), ptr(skip), ptr(out), ptr(dz), ptr(dlogdet), ptr(dy), ptr(dout), ptr(dskip),
                 ptr(d_rs), ptr(d_xin), ptr(dx_wn), ptr(dx), ptr(dcond_l), B, C, H, T, ctx.taps, dil_rate, n_split, int(sigmoid_scale),
                 int(two_src), int(io), wgrad.side.cuda_stream if wgrad.enabled else None)
        finally:
            bplan.unbind(bound)
        live = [p for p in params if p is not None]
        _mark_direct(live, True)
        if wgrad.enabled:
            # ... the MASK included: the start conv's weight gradient multiplies its d operand by it (the only 1x1 weight gradient that
            # does).  It is a small tensor made in the forward, freed when the last block's backward returns — and the caching allocator
            # then hands its memory to the next small allocation on the main stream while the side stream, a block or two behind, has
            # not run that launch yet: the start conv of the last side-stream block(s) came out with a weight AND bias gradient of
            # zero, once in ~200 steps in native-fp32 arithmetic (where the side stream lags most), 3 of ~25 full test-suite runs
            # (round 5; tools/race_hunt_c5.py)
            for t in (y, y0h, h0, xs, acts, skip, dout, dskip, d_rs, d_xin, dx_wn, m2):     # read by the second stream after this returns
                if t is not None:
                    t.record_stream(wgrad.side)
            with torch.cuda.stream(wgrad.side):          # every gradient of the block is complete at this point of THAT stream
                _notify(live)
        else:
            _notify(live)
        dcond = None if dcond_l is None else dcond_l.permute(1, 0, 2).reshape(ctx.cond_shape)
        return (dx, None, None, None, None, None, dcond) + (None,) * len(params)


class FlowStackFn(Function):
    """ALL decoder blocks of FlowSpecDecoder (reference models.py:193-211: the loop over `self.flows`) as ONE autograd node
    (round 4).  The launch sequence is FlowBlockFn's, block by block — one `glowtts_flow_block_fwd_io` /
    `_bwd_io` call each — but the host does per STACK what it did per block: one allocation per kind of activation slab
    ((n_blocks, ...) tensors; a block's buffers are slices addressed by pointer arithmetic), one saved-tensor list, one autograd
    node instead of twelve.  Per block the Python side is the weight pack, the plane binding, the cached table and one C call.
    fp32 or bf16 tensors (`io` as in FlowBlockFn), no conditioning input; anything else takes the per-block nodes."""

    @staticmethod
    def forward(ctx, x, m2, x_len, drops, cfg, bplans, counts, *params):
        import ctypes
        params, ctx.n_param_args = _unpack_params(params)
        n_split, sigmoid_scale, p_drop, dil_rate, n_layers, H, io = cfg
        fdt = torch.bfloat16 if io & 2 else torch.float32
        adt = torch.bfloat16 if io & 1 else torch.float32
        eF, eA = (2 if io & 2 else 4), (2 if io & 1 else 4)
        nb = len(bplans)
        x = x.contiguous()
        if x.dtype != fdt:
            raise RuntimeError(f"FlowStackFn: flow tensor is {x.dtype}, the stack runs with {fdt} flow tensors")
        B, C, T = x.shape
        dev = x.device
        new = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)                      # noqa: E731
        act = lambda *shape: torch.empty(shape, device=dev, dtype=adt)                                # noqa: E731
        zs, y = torch.empty(nb, B, C, T, device=dev, dtype=fdt), torch.empty(nb, B, C, T, device=dev, dtype=fdt)
        out = new(nb, B, C, T)
        y0h = act(nb, B, C // 2, T) if io == 1 else None
        h0, skip = act(nb, B, H, T), act(nb, B, H, T)
        acts, ts = act(nb, n_layers, B, H, T), act(nb, n_layers, B, 2 * H, T)
        nx = max(n_layers - 1, 1)
        xs = act(nb, nx, B, H, T)
        logdets = new(nb, B)
        winv = new(nb, n_split * n_split + 1)
        if p_drop > 0.0 and (drops is None or tuple(drops.shape) != (nb, n_layers, B, 2 * H, T) or not drops.is_contiguous()):
            drops = ops.keep_mask((nb, n_layers, B, 2 * H, T), p_drop, dev, "decoder.wn")
        if p_drop <= 0.0:
            drops = None
        scale = 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0
        nC, nH = B * C * T, B * H * T                             # elements of a flow / hidden tensor
        px, pm, pl = ptr(x), ptr(m2), ptr(x_len)
        pz, py, po, ph, psk, py0 = ptr(zs), ptr(y), ptr(out), ptr(h0), ptr(skip), ptr(y0h)
        pa, pts, pxs, pld, pw = ptr(acts), ptr(ts), ptr(xs), ptr(logdets), ptr(winv)
        pdr = ptr(drops)
        taps = params[8].shape[2]
        # fp32 tensors: block k's affine apply runs fused with block k + 1's ActNorm + InvConv (one pass over the flow tensor
        # instead of two; z_k is never written) — the block executors are told to leave those launches out (io bits 8 / 9)
        fuse = _FUSE_FLOWS and io == 0 and nb > 1 and n_split in (2, 4)      # (the fused kernels keep a group in registers: N <= 4)
        # ... and, where the shapes allow, end conv(k) + those flows + start conv(k + 1) as ONE launch (csrc/flow_boundary.hip)
        boundary = fuse and _FLOW_BOUNDARY and T % 4 == 0 and C <= 192 and C // n_split <= 64 and H <= 192
        prev_tab = None
        stack_pack = _STACK_PACK and nb > 1
        if stack_pack:                                            # every block's weight norm + packing (+ planes) in ONE launch
            st = getattr(bplans[0], "_stack_arena", None)
            if st is None:
                st = bplans[0]._stack_arena = StackArena()
            cps, o2 = [], 0
            for k in range(nb):
                cps.append(FlowBlockPlan.conv_params(params[o2: o2 + counts[k]], n_layers))
                o2 += counts[k]
            st.pack([bp.plan for bp in bplans], cps, 2 + 2 * n_layers)
        # W^-1 and log det W of every block's invertible 1x1 convolution: one launch (a table of the weights' addresses, cached)
        stack_prep = _STACK_PACK and nb > 1
        if stack_prep:
            o2, wkey = 0, []
            for k in range(nb):
                wkey.append(params[o2 + 2].data_ptr())
                o2 += counts[k]
            wtab = getattr(bplans[0], "_w_table", None)
            if wtab is None or wtab[0] != wkey:
                wtab = bplans[0]._w_table = (wkey, torch.tensor(wkey, dtype=torch.int64).to(dev))
            call("glowtts_invconv_prepare_multi", ptr(wtab[1]), pw, n_split * n_split + 1, nb, n_split)
        # Two half-batch chains on two streams (forward only; fp32 tensors with the boundary launch): every utterance is independent in
        # the forward, and two DIFFERENT kernels sharing the CUs fill each other's prologues and epilogues where two workgroups of
        # one kernel run in lock-step (tools/halfbatch_probe.py: a WN stack's forward 272 -> 245 us).  Same kernels, same slabs — a
        # call covers B / 2 utterances of every slab (tab.reserved = utterances per layer slab) —, results bit for bit the same.
        halves = _FWD_CHAINS if (_HALF_BATCH_FWD and boundary and stack_prep and stack_pack and B % _FWD_CHAINS == 0
                                 and os.environ.get("GLOWTTS_SIDE_STREAM", "1") != "0"       # (one-stream profiling passes)
                                 ) else 1
        if halves > 1:
            main_s = torch.cuda.current_stream(dev)
            extra = [_hip.side_stream(dev, "fwd%d" % (j + 2)) for j in range(halves - 1)]
            for s2 in extra:
                s2.wait_stream(main_s)
                for t_ in (x, m2, x_len, zs, y, out, h0, skip, acts, ts, xs, logdets, winv, drops):
                    if t_ is not None:
                        t_.record_stream(s2)
            chains, Bh = [main_s.cuda_stream] + [s2.cuda_stream for s2 in extra], B // halves
        off = 0
        for k in range(nb):
            pk = params[off: off + counts[k]]
            off += counts[k]
            bplan = bplans[k]
            plan = bplan.plan
            if not stack_pack:
                plan.ensure(FlowBlockPlan.conv_params(pk, n_layers), n_convs=2 + 2 * n_layers)
                plan.pack()
            bound = bplan.bind(io)
            try:
                tab = bplan.table(pk, n_layers)
                tab.w_inv = pw + k * (n_split * n_split + 1) * 4
                tab.logdet_w = tab.w_inv + 4 * n_split * n_split
                tab.reserved = B if halves > 1 else 0    # utterances per layer slab when a call covers only a part of them
                flags = int(io) | (1024 if stack_prep else 0)
                if halves > 1:
                    flags |= (256 | 4096 if k > 0 else 0) | (512 | 2048 if k < nb - 1 else 0)
                    sig_i = int(sigmoid_scale)
                    for hh in range(halves):
                        st, hb = chains[hh], hh * Bh
                        oC, oH, oB, oM = hb * C * T * 4, hb * H * T * 4, hb * 4, hb * T * 4
                        if k > 0:
                            _hip.call_on(st, "glowtts_flow_boundary_fwd", psk + (k - 1) * nH * 4 + oH, prev_tab.wf_end, prev_tab.b_end,
                                         py + (k - 1) * nC * 4 + oC, pm + oM, ptr(pk[0]), ptr(pk[1]), ptr(pk[2]), tab.logdet_w, pl + oB,
                                         tab.wf_start, tab.b_start, po + (k - 1) * nC * 4 + oC, py + k * nC * 4 + oC, ph + k * nH * 4 + oH,
                                         pld + (k - 1) * B * 4 + oB, pld + k * B * 4 + oB, Bh, C, H, T, n_split, sig_i)
                        _hip.call_on(st, "glowtts_flow_block_fwd_io", ctypes.addressof(tab),
                                     (px if k == 0 else pz + (k - 1) * nC * 4) + oC, pm + oM, pl + oB, None,
                                     None if pdr is None else pdr + k * n_layers * 2 * nH + hb * 2 * H * T, scale, py + k * nC * 4 + oC, None,
                                     ph + k * nH * 4 + oH, pxs + k * nx * nH * 4 + oH if n_layers > 1 else None,
                                     pa + k * n_layers * nH * 4 + oH, pts + k * n_layers * 2 * nH * 4 + 2 * oH, psk + k * nH * 4 + oH,
                                     po + k * nC * 4 + oC, pz + k * nC * 4 + oC, pld + k * B * 4 + oB, Bh, C, H, T, taps, dil_rate, n_split,
                                     sig_i, flags)
                    tab.reserved = 0                 # (read by the calls above while they queued their launches; the table is cached)
                    prev_tab = tab
                    continue
                if fuse and k > 0 and not stack_prep:
                    call("glowtts_invconv_prepare", ptr(pk[2]), tab.w_inv, tab.logdet_w, n_split)
                if boundary and k > 0:
                    call("glowtts_flow_boundary_fwd", psk + (k - 1) * nH * 4, prev_tab.wf_end, prev_tab.b_end, py + (k - 1) * nC * 4, pm,
                         ptr(pk[0]), ptr(pk[1]), ptr(pk[2]), tab.logdet_w, pl, tab.wf_start, tab.b_start, po + (k - 1) * nC * 4,
                         py + k * nC * 4, ph + k * nH * 4, pld + (k - 1) * B * 4, pld + k * B * 4, B, C, H, T, n_split,
                         int(sigmoid_scale))
                    flags |= 256 | 4096
                elif fuse and k > 0:
                    call("glowtts_coupling_actnorm_invconv_fwd", py + (k - 1) * nC * 4, po + (k - 1) * nC * 4, pm, ptr(pk[0]), ptr(pk[1]),
                         ptr(pk[2]), tab.logdet_w, pl, py + k * nC * 4, pld + (k - 1) * B * 4, pld + k * B * 4, B, C, T, n_split,
                         int(sigmoid_scale))
                    flags |= 256
                if fuse and k < nb - 1:
                    flags |= 512 | (2048 if boundary else 0)
                prev_tab = tab
                call("glowtts_flow_block_fwd_io", ctypes.addressof(tab), px if k == 0 else pz + (k - 1) * nC * eF, pm, pl, None,
                     None if pdr is None else pdr + k * n_layers * 2 * nH, scale, py + k * nC * eF,
                     None if py0 is None else py0 + k * (nC // 2) * eA, ph + k * nH * eA,
                     pxs + k * nx * nH * eA if n_layers > 1 else None, pa + k * n_layers * nH * eA,
                     pts + k * n_layers * 2 * nH * eA, psk + k * nH * eA, po + k * nC * 4, pz + k * nC * eF, pld + k * B * 4, B, C, H, T,
                     taps, dil_rate, n_split, int(sigmoid_scale), flags)
            finally:
                bplan.unbind(bound)
        if halves > 1:
            for s2 in extra:
                main_s.wait_stream(s2)
        ctx.save_for_backward(x, m2, x_len, zs, y, h0, acts, ts, skip, out, winv, xs, *([] if drops is None else [drops]),
                              *([] if y0h is None else [y0h]))
        ctx.cfg, ctx.bplans, ctx.counts, ctx.params, ctx.taps, ctx.scale, ctx.fuse = cfg, bplans, counts, params, taps, scale, fuse
        ctx.boundary = boundary
        return zs[nb - 1], logdets.sum(0)

    @staticmethod
    @once_differentiable
    def backward(ctx, dz, dlogdet):
        import ctypes
        n_split, sigmoid_scale, p_drop, dil_rate, n_layers, H, io = ctx.cfg
        fdt = torch.bfloat16 if io & 2 else torch.float32
        adt = torch.bfloat16 if io & 1 else torch.float32
        eF, eA = (2 if io & 2 else 4), (2 if io & 1 else 4)
        sv = list(ctx.saved_tensors)
        x, m2, x_len, zs, y, h0, acts, ts, skip, out, winv, xs = sv[:12]
        rest = sv[12:]
        drops = rest.pop(0) if p_drop > 0 else None
        y0h = rest.pop(0) if io == 1 else None
        params, bplans, counts = ctx.params, ctx.bplans, ctx.counts
        nb = len(bplans)
        B, C, T = x.shape
        dev = x.device
        if not all(p.grad is not None and p.grad.is_contiguous() for p in params):
            raise RuntimeError("FlowStackFn.backward: a parameter gradient buffer disappeared between forward and backward")
        act = lambda *shape: torch.empty(shape, device=dev, dtype=adt)                                # noqa: E731
        flow = lambda *shape: torch.empty(shape, device=dev, dtype=fdt)                               # noqa: E731
        dz = dz.contiguous() if dz is not None else torch.zeros_like(x)
        if dz.dtype != fdt:
            dz = dz.to(fdt)
        dlogdet = dlogdet.contiguous().float() if dlogdet is not None else torch.zeros(B, device=dev)
        wgrad = _WgradStream(dev)
        two_src = (not io) and dil_rate == 1 and H % 192 == 0 and T % 4 == 0
        # per-block scratch as slices of one allocation per kind: the weight-gradient stream reads a block's scratch after the
        # chain has moved on to the next block, so nothing is shared between blocks
        dy, dout, dxs = flow(nb, B, C, T), act(nb, B, C, T), flow(nb, B, C, T)
        dskip = act(nb, B, H, T)
        d_rs = act(nb, B, H, T) if two_src else act(nb, n_layers, B, 2 * H, T)
        d_xin, dx_wn = act(nb, n_layers, B, 2 * H, T), act(nb, n_layers, B, H, T)
        nC, nH = B * C * T, B * H * T
        nx = max(n_layers - 1, 1)
        px, pm, pl = ptr(x), ptr(m2), ptr(x_len)
        pz, py, po, ph, psk, py0 = ptr(zs), ptr(y), ptr(out), ptr(h0), ptr(skip), ptr(y0h)
        pa, pts, pxs, pw = ptr(acts), ptr(ts), ptr(xs), ptr(winv)
        pdr = ptr(drops)
        pdy, pdo, pdx, pds, pdrs, pdxin, pdxw = ptr(dy), ptr(dout), ptr(dxs), ptr(dskip), ptr(d_rs), ptr(d_xin), ptr(dx_wn)
        pdz, pdl = ptr(dz), ptr(dlogdet)
        drs_stride = (nH if two_src else n_layers * 2 * nH) * eA
        side = wgrad.side.cuda_stream if wgrad.enabled else None
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + c)
        if wgrad.enabled:                                # read by the second stream after this returns
            for t in (y, y0h, h0, xs, acts, skip, dout, dskip, d_rs, d_xin, dx_wn, zs, m2):    # (m2: see FlowBlockFn.backward)
                if t is not None:
                    t.record_stream(wgrad.side)
        # between blocks (fp32 tensors): start conv backward-data(k) + ActNorm / InvConv backward(k) + coupling backward(k - 1) + end conv
        # backward-data(k - 1) as ONE launch on the chain (csrc/flow_boundary.hip); the ActNorm / InvConv parameter gradients come out
        # as per-workgroup partials, added up by a small launch behind the block's weight gradients
        boundary = ctx.fuse and ctx.boundary and _FLOW_BOUNDARY_BWD
        if boundary:
            n_part = B * ((T + 31) // 32) * (C // n_split) * (2 * n_split + n_split * n_split)
            part = torch.empty(nb - 1, n_part, device=dev, dtype=torch.float32)
            ppart = ptr(part)
            if wgrad.enabled:
                part.record_stream(wgrad.side)
        for k in range(nb - 1, -1, -1):
            pk = params[offs[k]: offs[k + 1]]
            bplan = bplans[k]
            on_side = wgrad.enabled and k >= _WGRAD_MAIN_BLOCKS      # the backward's last blocks keep their weight gradients on the chain
            tab = bplan.table(pk, n_layers)
            tab.w_inv = pw + k * (n_split * n_split + 1) * 4
            tab.logdet_w = tab.w_inv + 4 * n_split * n_split
            fuse = ctx.fuse
            flags = int(io) | (256 if fuse and k > 0 else 0) | (512 if fuse and k < nb - 1 else 0)
            if boundary:
                flags |= (4096 if k > 0 else 0) | (2048 if k < nb - 1 else 0)
            bound = bplan.bind(io)
            try:
                call("glowtts_flow_block_bwd_io", ctypes.addressof(tab), px if k == 0 else pz + (k - 1) * nC * eF, pm, pl,
                     None if pdr is None else pdr + k * n_layers * 2 * nH, ctx.scale, py + k * nC * eF,
                     None if py0 is None else py0 + k * (nC // 2) * eA, ph + k * nH * eA,
                     pxs + k * nx * nH * eA if n_layers > 1 else None, pa + k * n_layers * nH * eA,
                     pts + k * n_layers * 2 * nH * eA, psk + k * nH * eA, po + k * nC * 4,
                     pdz if k == nb - 1 else pdx + (k + 1) * nC * eF, pdl, pdy + k * nC * eF, pdo + k * nC * eA, pds + k * nH * eA,
                     pdrs + k * drs_stride, pdxin + k * n_layers * 2 * nH * eA, pdxw + k * n_layers * nH * eA, pdx + k * nC * eF, None,
                     B, C, H, T, ctx.taps, dil_rate, n_split, int(sigmoid_scale), int(two_src), flags, side if on_side else None)
            finally:
                bplan.unbind(bound)
            live = [p for p in pk if p is not None]
            _mark_direct(live, True)
            ai_late = fuse and k > 0                     # logs / bias / W of this block: their gradients come from the fused kernel below
            conv_live = live[3:] if ai_late else live
            if on_side:
                with torch.cuda.stream(wgrad.side):      # every gradient of the block is complete at this point of THAT stream
                    _notify(conv_live)
            else:
                _notify(conv_live)
            if ai_late and boundary:
                tab_prev = bplans[k - 1].table(params[offs[k - 1]: offs[k]], n_layers)
                pp = ppart + (k - 1) * n_part * 4
                call("glowtts_flow_boundary_bwd", pdxw + k * n_layers * nH * 4, tab.wb_start, pdy + k * nC * 4, py + (k - 1) * nC * 4,
                     po + (k - 1) * nC * 4, pm, ptr(pk[0]), ptr(pk[1]), ptr(pk[2]), pdl, tab_prev.wb_end, pdy + (k - 1) * nC * 4,
                     pdo + (k - 1) * nC * 4, pds + (k - 1) * nH * 4, pp, B, C, H, T, n_split, int(sigmoid_scale), int(two_src))
                if on_side:
                    wgrad.side.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(wgrad.side):
                        call("glowtts_flow_boundary_bwd_reduce", pp, tab.w_inv, pdl, pl, ptr(pk[0].grad), ptr(pk[1].grad),
                             ptr(pk[2].grad), B, C, T, n_split)
                        _notify(live[:3])
                else:
                    call("glowtts_flow_boundary_bwd_reduce", pp, tab.w_inv, pdl, pl, ptr(pk[0].grad), ptr(pk[1].grad), ptr(pk[2].grad),
                         B, C, T, n_split)
                    _notify(live[:3])
            elif ai_late:
                # block k's ActNorm + InvConv backward fused with block k - 1's coupling backward: dy_k -> dy_{k-1}, dout_{k-1}
                call("glowtts_coupling_actnorm_invconv_bwd", py + (k - 1) * nC * 4, po + (k - 1) * nC * 4, pm, ptr(pk[0]), ptr(pk[1]),
                     ptr(pk[2]), tab.w_inv, pdy + k * nC * 4, pdl, pl, pdy + (k - 1) * nC * 4, pdo + (k - 1) * nC * 4, ptr(pk[0].grad),
                     ptr(pk[1].grad), ptr(pk[2].grad), B, C, T, n_split, int(sigmoid_scale))
                _notify(live[:3])
        return (dxs[0], None, None, None, None, None, None) + (None,) * ctx.n_param_args


# ----------------------------------------------------------------------------------------------------------------
_ENC_WGRAD = os.environ.get("GLOWTTS_ENC_WGRAD", "0") == "1"      # tuning knob: encoder weight gradients on the "wgrad" stream
_FUSE_FLOWS = os.environ.get("GLOWTTS_FUSE_FLOWS", "1") != "0"    # FlowStackFn: coupling(k) fused with ActNorm + InvConv (k + 1)
_FLOW_BOUNDARY = os.environ.get("GLOWTTS_FLOW_BOUNDARY", "1") != "0"   # ... and with end conv(k) / start conv(k + 1): one launch
# FlowStackFn forward as two half-batch chains on two streams (even batches, fp32 tensors): 13.71 -> 13.45 ms per step for ~1 ms more
# host enqueue (twice the decoder's forward launches); GLOWTTS_HALF_BATCH_FWD=0 keeps one chain (a rank whose host is the bottleneck)
# Round 4 switched the two chains off under a process group (bench.py --rccl-self: 14.9 -> 16.8 ms, "cause not isolated").  Round 5
# isolated it (DESIGN.md lesson 38): the second chain's stream and the main stream had landed on the SAME hardware queue (HIP maps
# streams onto GPU_MAX_HW_QUEUES = 4 queues in order of first use, and a process group's streams come first), so the two chains ran one
# after the other and every event wait of one stalled the other.  The second chain now runs on the weight-gradient stream (idle during
# the forward; _hip._FWD2_ON_WGRAD) — one stream fewer, 13.87 against 15.72 ms with the reducer attached, 13.29 against 13.36 without —
# and the chains are on with or without a process group.
_HALF_BATCH_ENV = os.environ.get("GLOWTTS_HALF_BATCH_FWD", "")
_HALF_BATCH_FWD = _HALF_BATCH_ENV != "0"
_FWD_CHAINS = int(os.environ.get("GLOWTTS_FWD_CHAINS", "2"))          # (chains of B / n utterances; 2 measured best)
# ... and the same in the backward: OPT-IN.  The kernel is 32 us against 45 for the three launches alone, but in the step the chain
# waits for the weight gradients' compute units at every block boundary of the backward whatever it launches (13.73 -> 13.72 ms per
# step), and the extra stream hand-over for its parameter-gradient reduction costs 1.4 ms of host enqueue (DESIGN.md lesson 36)
_FLOW_BOUNDARY_BWD = os.environ.get("GLOWTTS_FLOW_BOUNDARY_BWD", "0") == "1"


def _enc_layer_table(group, attn, ffn, norm1, norm2):
    """Host-side `glowtts_enc_layer` of one transformer layer, cached on the layer's ConvGroup (whose plan owns the packed
    weights of conv_q, conv_k, conv_v, conv_o, conv_1, conv_2 in that order)."""
    plan = group.plan
    convs = [attn.conv_q, attn.conv_k, attn.conv_v, attn.conv_o, ffn.conv_1, ffn.conv_2]
    others = [attn.emb_rel_k, attn.emb_rel_v] if attn.window_size is not None else [None, None]
    others += [norm1.gamma, norm1.beta, norm2.gamma, norm2.beta]
    live = [c.weight for c in convs] + [c.bias for c in convs] + [p for p in others if p is not None]
    key = (plan.version, tuple(p.data_ptr() for p in live), tuple(p.grad.data_ptr() for p in live))
    if getattr(group, "_enc_key", None) != key:
        t = _hip.EncLayer()
        gdesc, prefix = plan.unpack_tables(group._params())
        for i, (name, c) in enumerate(zip("qkvo12", convs)):
            wf, wb = plan.convs[i][2], plan.convs[i][3]
            setattr(t, "wf_" + name, wf.data_ptr())
            setattr(t, "wb_" + name, wb.data_ptr())
            setattr(t, "b_" + name, c.bias.data_ptr())
            setattr(t, "dwp_" + name, plan.dwp_view(i).data_ptr())
            setattr(t, "db_" + name, c.bias.grad.data_ptr())
        ek, ev = others[0], others[1]
        t.emb_k, t.emb_v = (None, None) if ek is None else (ek.data_ptr(), ev.data_ptr())
        t.demb_k, t.demb_v = (None, None) if ek is None else (ek.grad.data_ptr(), ev.grad.data_ptr())
        t.gamma1, t.beta1, t.gamma2, t.beta2 = (p.data_ptr() for p in others[2:])
        t.dgamma1, t.dbeta1, t.dgamma2, t.dbeta2 = (p.grad.data_ptr() for p in others[2:])
        t.pack_desc = None                       # ConvGroup.begin() has packed this layer's weights already
        t.unpack_desc, t.pack_prefix = gdesc.data_ptr(), prefix.data_ptr()
        t.dwp_all, t.dwp_floats = plan.dwp.data_ptr(), plan.dwp.numel()
        t.n_conv, t.total_rows = len(plan.convs), plan.total_rows
        group._enc_tab, group._enc_key, group._enc_live = t, key, live
    group._enc_tab.attn_bf16 = int(bool(getattr(attn, "bf16_mma", False)))
    return group._enc_tab, group._enc_live


def encoder_layer_eligible(group, attn, ffn, norm1, norm2, x) -> bool:
    """Can this transformer layer run as ONE native call each way (EncoderLayerFn)?  Training step on the GPU with every
    gradient buffer allocated and written in place, the attention kernel's envelope, ReLU FFN, biased convolutions."""
    if not (x.is_cuda and torch.is_grad_enabled() and _hip.timing_off() and _WN_NATIVE == "both" and direct_grads_enabled()):
        return False
    if not group.active or group.modules != [attn.conv_q, attn.conv_k, attn.conv_v, attn.conv_o, ffn.conv_1, ffn.conv_2]:
        return False
    t = x.size(2)
    if not (attn.k_channels % 16 == 0 and attn.k_channels <= 128 and (attn.window_size is None or attn.window_size <= 7)
            and not attn.proximal_bias and ffn.activation is None and ffn.kernel_size % 2 == 1):
        return False
    if attn.conv_o.weight.shape[0] != attn.channels or any(c.bias is None for c in group.modules):
        return False
    params = [attn.emb_rel_k, attn.emb_rel_v] if attn.window_size is not None else []
    params += [norm1.gamma, norm1.beta, norm2.gamma, norm2.beta] + [c.bias for c in group.modules]
    return all(p.requires_grad and p.grad is not None and p.grad.is_contiguous() for p in params)


class EncoderStackFn(Function):
    """ALL post-LN transformer layers of the text encoder (reference attentions.py:63-73: the loop over the layers) as ONE
    autograd node (round 4): EncoderLayerFn's native call per layer, but one activation buffer, one workspace, one saved-tensor
    list and one node for the stack — the layers' slices are addressed by pointer arithmetic.  `keep`: the byte keep-masks of all
    four dropouts of all layers, laid out layer by layer as [attention (B,h,T,T) | branch 1 (B,H,T) | FFN (B,F,T) | branch 2
    (B,H,T)], or None."""

    @staticmethod
    def forward(ctx, x, m2, keep, cfg, layers, counts, *params):
        import ctypes
        params, ctx.n_param_args = _unpack_params(params)
        heads, taps, window, share, blk, eps, p_drop = cfg
        nl = len(layers)
        x = f32(x.contiguous())
        B, H, T = x.shape
        F_ = layers[0][2].filter_channels
        dev = x.device
        n_ht, n_ft, n_tt = B * H * T, B * F_ * T, B * heads * T * T
        per = 8 * n_ht + n_ft + n_tt + 4 * B * T                  # floats of one layer's activations (EncoderLayerFn's layout)
        buf = torch.empty(nl, per, device=dev, dtype=torch.float32)
        scale = 1.0 / (1.0 - p_drop) if keep is not None else 1.0
        pk_, pb, px, pm = ptr(keep), ptr(buf), ptr(x), ptr(m2)
        ksz = n_tt + n_ht + n_ft + n_ht                            # keep bytes of one layer
        o_q, o_k, o_v, o_ya, o_o, o_x1, o_y2, o_x2 = (4 * j * n_ht for j in range(8))
        o_h = 4 * 8 * n_ht
        o_p = o_h + 4 * n_ft
        o_s1 = o_p + 4 * n_tt
        o_s2 = o_s1 + 4 * 2 * B * T
        for l, (group, attn, ffn, norm1, norm2) in enumerate(layers):
            tab, _ = _enc_layer_table(group, attn, ffn, norm1, norm2)
            base = pb + 4 * l * per
            xin = px if l == 0 else pb + 4 * (l - 1) * per + o_x2
            kb = None if pk_ is None else pk_ + l * ksz
            bound = group.plan.bind()                    # the FFN convolutions take the group's bf16 planes (conv arithmetic)
            try:
                call("glowtts_encoder_layer_fwd", ctypes.addressof(tab), xin, pm, kb, None if kb is None else kb + n_tt,
                     None if kb is None else kb + n_tt + n_ht, None if kb is None else kb + n_tt + n_ht + n_ft, scale,
                     base + o_q, base + o_k, base + o_v, base + o_p, base + o_ya, base + o_o, base + o_x1, base + o_s1, base + o_h,
                     base + o_y2, base + o_x2, base + o_s2, B, H, F_, T, heads, taps, window, share, blk, float(eps))
            finally:
                WNPackPlan.unbind(bound)
            attn.attn = buf[l, (o_p // 4): (o_p // 4) + n_tt].view(B, heads, T, T).detach()
        ctx.save_for_backward(x, m2, buf, *([] if keep is None else [keep]))
        ctx.cfg, ctx.layers, ctx.counts, ctx.dims, ctx.scale, ctx.params = cfg, layers, counts, (B, H, F_, T), scale, params
        return buf[nl - 1, (o_x2 // 4): (o_x2 // 4) + n_ht].view(B, H, T)

    @staticmethod
    @once_differentiable
    def backward(ctx, dx_out):
        import ctypes
        heads, taps, window, share, blk, eps, p_drop = ctx.cfg
        layers, counts, params = ctx.layers, ctx.counts, ctx.params
        nl = len(layers)
        B, H, F_, T = ctx.dims
        sv = ctx.saved_tensors
        x, m2, buf = sv[:3]
        keep = sv[3] if len(sv) > 3 else None
        dev = x.device
        n_ht, n_ft, n_tt = B * H * T, B * F_ * T, B * heads * T * T
        per = 8 * n_ht + n_ft + n_tt + 4 * B * T
        wper = 10 * n_ht + n_ft + n_tt                            # floats of one layer's backward workspace
        ws = torch.empty(nl, wper, device=dev, dtype=torch.float32)
        dx_out = f32(dx_out.contiguous())
        wgrad = _WgradStream(dev)
        wgrad.enabled = wgrad.enabled and _ENC_WGRAD          # (see EncoderLayerFn.backward: the weight gradients stay on this stream)
        pk_, pb, px, pm, pw, pdo = ptr(keep), ptr(buf), ptr(x), ptr(m2), ptr(ws), ptr(dx_out)
        ksz = n_tt + n_ht + n_ft + n_ht
        o_q, o_k, o_v, o_ya, o_o, o_x1, o_y2, o_x2 = (4 * j * n_ht for j in range(8))
        o_h = 4 * 8 * n_ht
        o_p = o_h + 4 * n_ft
        o_s1 = o_p + 4 * n_tt
        o_s2 = o_s1 + 4 * 2 * B * T
        w_dx1a, w_dy2, w_dx1, w_dxa, w_do, w_dya, w_dq, w_dk, w_dv, w_dx = (4 * j * n_ht for j in range(10))
        w_dpre = 4 * 10 * n_ht
        w_ds = w_dpre + 4 * n_ft
        side = wgrad.side.cuda_stream if wgrad.enabled else None
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + c)
        if wgrad.enabled:
            for t in (x, buf, ws, dx_out, m2, keep):         # (masks too: see FlowBlockFn.backward)
                if t is not None:
                    t.record_stream(wgrad.side)
        for l in range(nl - 1, -1, -1):
            group, attn, ffn, norm1, norm2 = layers[l]
            tab, live = _enc_layer_table(group, attn, ffn, norm1, norm2)
            base, wb = pb + 4 * l * per, pw + 4 * l * wper
            xin = px if l == 0 else pb + 4 * (l - 1) * per + o_x2
            din = pdo if l == nl - 1 else pw + 4 * (l + 1) * wper + w_dx
            kb = None if pk_ is None else pk_ + l * ksz
            bound = group.plan.bind()
            try:
                call("glowtts_encoder_layer_bwd", ctypes.addressof(tab), xin, pm, kb, None if kb is None else kb + n_tt,
                     None if kb is None else kb + n_tt + n_ht, None if kb is None else kb + n_tt + n_ht + n_ft, ctx.scale,
                     base + o_q, base + o_k, base + o_v, base + o_p, base + o_ya, base + o_o, base + o_x1, base + o_s1, base + o_h,
                     base + o_y2, base + o_s2, din, wb + w_dx1a, wb + w_dy2, wb + w_dpre, wb + w_dx1, wb + w_dxa, wb + w_do,
                     wb + w_dya, wb + w_ds, wb + w_dq, wb + w_dk, wb + w_dv, wb + w_dx, B, H, F_, T, heads, taps, window, share,
                     blk, side)
            finally:
                WNPackPlan.unbind(bound)
            _mark_direct(live, True)
            if wgrad.enabled:
                with torch.cuda.stream(wgrad.side):
                    _notify(live)
            else:
                _notify(live)
        return (ws[0, (w_dx // 4): (w_dx // 4) + n_ht].view(B, H, T), None, None, None, None, None) + (None,) * ctx.n_param_args


class EncoderLayerFn(Function):
    """One post-LN transformer layer of the text encoder (reference attentions.py:63-73, 204-264, 373-381) as ONE autograd
    node: forward and backward are one native call each (csrc/wn_stack.hip: glowtts_encoder_layer_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x, m2, drops, cfg, mods, *params):
        import ctypes
        heads, taps, window, share, blk, eps, p_drop = cfg
        group, attn, ffn, norm1, norm2 = mods
        x = f32(x.contiguous())
        B, H, T = x.shape
        F_ = ffn.filter_channels
        dev = x.device
        tab, live = _enc_layer_table(group, attn, ffn, norm1, norm2)
        n_ht, n_ft, n_tt = B * H * T, B * F_ * T, B * heads * T * T
        buf = torch.empty(8 * n_ht + n_ft + n_tt + 4 * B * T, device=dev, dtype=torch.float32)
        off = [0]

        def take(n, *shape):
            v = buf[off[0]: off[0] + n].view(*shape)
            off[0] += n
            return v
        q, k, v, y_att, o, x1, y2, x2 = (take(n_ht, B, H, T) for _ in range(8))
        h = take(n_ft, B, F_, T)
        p_attn = take(n_tt, B, heads, T, T)
        stats1, stats2 = take(2 * B * T, B, 2, T), take(2 * B * T, B, 2, T)
        da, do_, dh, d2 = drops if drops is not None else (None, None, None, None)
        scale = 1.0 / (1.0 - p_drop) if drops is not None else 1.0
        bound = group.plan.bind()                        # the FFN convolutions take the group's bf16 planes (conv arithmetic)
        try:
            call("glowtts_encoder_layer_fwd", ctypes.addressof(tab), ptr(x), ptr(m2), ptr(da), ptr(do_), ptr(dh), ptr(d2), scale,
                 ptr(q), ptr(k), ptr(v), ptr(p_attn), ptr(y_att), ptr(o), ptr(x1), ptr(stats1), ptr(h), ptr(y2), ptr(x2),
                 ptr(stats2), B, H, F_, T, heads, taps, window, share, blk, float(eps))
        finally:
            WNPackPlan.unbind(bound)
        attn.attn = p_attn.detach()
        ctx.save_for_backward(x, m2, buf, *([] if drops is None else drops))
        ctx.cfg, ctx.mods, ctx.dims, ctx.scale, ctx.live = cfg, mods, (B, H, F_, T), scale, live
        return x2

    @staticmethod
    @once_differentiable
    def backward(ctx, dx2):
        import ctypes
        heads, taps, window, share, blk, eps, p_drop = ctx.cfg
        group, attn, ffn, norm1, norm2 = ctx.mods
        B, H, F_, T = ctx.dims
        sv = ctx.saved_tensors
        x, m2, buf = sv[:3]
        da, do_, dh, d2 = sv[3:7] if len(sv) > 3 else (None, None, None, None)
        dev = x.device
        n_ht, n_ft, n_tt = B * H * T, B * F_ * T, B * heads * T * T
        off = [0]

        def take(src, n, *shape):
            v = src[off[0]: off[0] + n].view(*shape)
            off[0] += n
            return v
        q, k, v, y_att, o, x1, y2, _x2 = (take(buf, n_ht, B, H, T) for _ in range(8))
        h = take(buf, n_ft, B, F_, T)
        p_attn = take(buf, n_tt, B, heads, T, T)
        stats1, stats2 = take(buf, 2 * B * T, B, 2, T), take(buf, 2 * B * T, B, 2, T)
        ws = torch.empty(10 * n_ht + n_ft + n_tt, device=dev, dtype=torch.float32)
        off[0] = 0
        dx1a, dy2, dx1, dxa, d_o, dy_att, dq, dk_, dv, dx = (take(ws, n_ht, B, H, T) for _ in range(10))
        d_pre1 = take(ws, n_ft, B, F_, T)
        ds = take(ws, n_tt, B, heads, T, T)
        dx2 = f32(dx2.contiguous())
        tab, live = _enc_layer_table(group, attn, ffn, norm1, norm2)
        # The layer's weight-gradient kernels stay on the stream this backward runs on (the text encoder's own side stream,
        # which has ~10 ms of decoder backward to hide ~3 ms of work): sent to the decoder's weight-gradient stream they queue
        # behind its 5-tap kernels and lengthen the tail every step ends with (A/B: +0.4 ms per step).
        wgrad = _WgradStream(dev)
        wgrad.enabled = wgrad.enabled and _ENC_WGRAD
        bound = group.plan.bind()
        try:
            call("glowtts_encoder_layer_bwd", ctypes.addressof(tab), ptr(x), ptr(m2), ptr(da), ptr(do_), ptr(dh), ptr(d2), ctx.scale,
                 ptr(q), ptr(k), ptr(v), ptr(p_attn), ptr(y_att), ptr(o), ptr(x1), ptr(stats1), ptr(h), ptr(y2), ptr(stats2),
                 ptr(dx2), ptr(dx1a), ptr(dy2), ptr(d_pre1), ptr(dx1), ptr(dxa), ptr(d_o), ptr(dy_att), ptr(ds), ptr(dq), ptr(dk_),
                 ptr(dv), ptr(dx), B, H, F_, T, heads, taps, window, share, blk,
                 wgrad.side.cuda_stream if wgrad.enabled else None)
        finally:
            WNPackPlan.unbind(bound)
        _mark_direct(live, True)
        if wgrad.enabled:
            for t in (x, buf, ws, dx2, m2):              # read by the second stream after this returns (the mask too: see FlowBlockFn.backward)
                if t is not None:
                    t.record_stream(wgrad.side)
            with torch.cuda.stream(wgrad.side):
                _notify(live)
        else:
            _notify(live)
        return (dx, None, None, None, None) + (None,) * len(live)
