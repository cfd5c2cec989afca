"""Flow and conv-stack operators with the reference's names, constructor arguments and state-dict keys
(reference: glow_tts_train/layers.py), computing through the HIP kernels in ``csrc/``.

Per-flow operator API: ``f(x, x_mask, g=None, reverse=False) -> (z, logdet | None)`` and
``f.store_inverse()``; ``x`` is ``(B, C, T)``, ``x_mask`` ``(B, 1, T)`` float 0/1, ``logdet`` ``(B,)``.
A private keyword ``x_len`` (``sum(x_mask)`` per utterance) may be passed by FlowSpecDecoder to avoid recomputing it.
"""
from __future__ import annotations

import os
import typing

import torch
from torch import nn
from torch.nn import functional as F

from . import convops, ops
from ._hip import direct_apply

_chan_ln_apply = direct_apply(convops.ChanLayerNormFn)
_wn_apply = direct_apply(convops.WNFn)


def _ones_mask(x: torch.Tensor) -> torch.Tensor:
    return torch.ones(x.size(0), 1, x.size(2), device=x.device, dtype=x.dtype)


class LayerNorm(nn.Module):
    """Normalisation over the CHANNEL axis of (B, C, T) with eps 1e-4 (reference layers.py:10-28)."""

    MAX_CHANNELS = 768          # csrc/norm.hip keeps a frame's channel column in registers (the reference has no limit)

    def __init__(self, channels, eps=1e-4):
        super().__init__()
        self.channels, self.eps = channels, eps      # any width builds and loads (checkpoints of wider models, CPU construction)
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))

    def forward(self, x, res=None, relu_in=False, relu_out=False, p_drop=0.0, site="ln"):
        """LayerNorm over channels of x (+ res: the residual add that precedes every norm in the encoder is fused in).
        relu_in / relu_out / p_drop: the ReLU before and the ReLU / dropout after the norm, inside its kernels (3-D input);
        `site` names the dropout's keep-mask (ops.keep_mask).  relu_in with a residual input is not supported (raises).
        (B, C, T) tensors of up to MAX_CHANNELS channels run on the register-resident kernels; wider norms and other ranks —
        neither occurs in the reference's model — take a composition of framework operators on the same device (never the CPU)."""
        if x.dim() == 3 and self.channels <= self.MAX_CHANNELS:      # (a CPU tensor raises in the operator: no fallback)
            return _chan_ln_apply(x, res, self.gamma, self.beta, self.eps, relu_in, relu_out, p_drop, site)
        if relu_in and res is not None:
            raise RuntimeError("LayerNorm: relu_in together with a residual input is not supported")
        # F.layer_norm normalises trailing dims: move channels last, normalise, move back
        v = x if res is None else x + res
        if relu_in:
            v = F.relu(v)
        y = F.layer_norm(v.transpose(1, -1), (self.channels,), self.gamma, self.beta, self.eps).transpose(1, -1)
        if relu_out:
            y = F.relu(y)
        if p_drop:
            from . import ops
            keep = ops.keep_mask(tuple(y.shape), p_drop, y.device, site)
            y = y * keep.to(y.dtype) * (1.0 / (1.0 - p_drop))
        return y


class ConvReluNorm(nn.Module):
    """Encoder pre-net (reference layers.py:31-80): n x [conv -> LayerNorm -> ReLU -> dropout], zero-init residual proj."""

    def __init__(self, in_channels, hidden_channels, out_channels, kernel_size, n_layers, p_dropout):
        super().__init__()
        assert n_layers > 1, "Number of layers should be larger than 0."
        self.in_channels, self.hidden_channels, self.out_channels = in_channels, hidden_channels, out_channels
        self.kernel_size, self.n_layers, self.p_dropout = kernel_size, n_layers, p_dropout
        widths = [in_channels] + [hidden_channels] * n_layers
        self.conv_layers = nn.ModuleList(
            nn.Conv1d(widths[i], widths[i + 1], kernel_size, padding=kernel_size // 2) for i in range(n_layers)
        )
        self.norm_layers = nn.ModuleList(LayerNorm(hidden_channels) for _ in range(n_layers))
        self.relu_drop = nn.Sequential(nn.ReLU(), nn.Dropout(p_dropout))
        self.proj = nn.Conv1d(hidden_channels, out_channels, 1)
        nn.init.zeros_(self.proj.weight)
        nn.init.zeros_(self.proj.bias)

    def forward(self, x, x_mask):
        m2 = ops.mask2d(x_mask)
        h = x
        drop = self.relu_drop[1]
        p = float(drop.p) if (self.training and drop.p > 0.0) else 0.0
        for i, (conv, norm) in enumerate(zip(self.conv_layers, self.norm_layers)):
            # conv(h * mask): the mask is folded into the conv; LayerNorm -> ReLU -> Dropout: ONE kernel (csrc/norm.hip)
            h = norm(convops.conv1d(conv, h, m2, mask_in=True), relu_out=True, p_drop=p, site=f"encoder.pre.{i}")
        return (x + convops.conv1d(self.proj, h)) * x_mask


class WN(nn.Module):
    """WaveNet-style gated conv stack of the coupling network (reference layers.py:83-170).

    Every convolution of the stack runs on the hand-written MFMA kernels (csrc/convgemm*.hip, convwrw_tr.hip): the
    k-tap in-convolution with the gate (bias + conditioning + dropout + tanh * sigmoid) as its epilogue, the 1x1 res/skip
    convolution with the residual / skip update as its epilogue, their backward-data and weight-gradient kernels; one
    native call queues the whole stack each way (csrc/wn_stack.hip).  No MIOpen / rocBLAS kernel is involved.
    """

    def __init__(self, in_channels, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0):
        super().__init__()
        assert kernel_size % 2 == 1
        assert hidden_channels % 2 == 0
        self.in_channels, self.hidden_channels = in_channels, hidden_channels
        self.kernel_size = (kernel_size,)
        self.dilation_rate, self.n_layers = dilation_rate, n_layers
        self.gin_channels, self.p_dropout = gin_channels, p_dropout
        wn = torch.nn.utils.weight_norm
        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        self.drop = nn.Dropout(p_dropout)
        if gin_channels != 0:
            self.cond_layer = wn(nn.Conv1d(gin_channels, 2 * hidden_channels * n_layers, 1), name="weight")
        for i in range(n_layers):
            d = dilation_rate ** i
            self.in_layers.append(
                wn(nn.Conv1d(hidden_channels, 2 * hidden_channels, kernel_size, dilation=d,
                             padding=(kernel_size * d - d) // 2), name="weight"))
            out_ch = 2 * hidden_channels if i < n_layers - 1 else hidden_channels
            self.res_skip_layers.append(wn(nn.Conv1d(hidden_channels, out_ch, 1), name="weight"))

    @staticmethod
    def _conv_params(conv):
        """(v, g, bias) of a conv with or without weight norm (store_inverse() strips it, reference layers.py:164-170)."""
        if hasattr(conv, "weight_v"):
            return conv.weight_v, conv.weight_g, conv.bias
        return conv.weight, None, conv.bias

    def forward(self, x, x_mask=None, g=None, **kwargs):
        if x_mask is None:
            x_mask = _ones_mask(x)
        m2 = kwargs.get("m2")
        if m2 is None:
            m2 = ops.mask2d(x_mask)
        # (B, 2H*n_layers, 1): a 1x1 convolution over ONE frame (layers.py:142-143).  On the device it runs on the library's own
        # convolution kernels like every other contraction of the step (round 5: left to the framework it was MIOpen's Winograd kernel
        # plus torch's weight-norm kernels, ~10 launches per block — the only library GEMM on the configs[4] training path)
        if g is None:
            cond = None
        elif g.is_cuda and g.dim() == 3:
            cond = convops.conv1d(self.cond_layer, g.contiguous())
        else:
            cond = self.cond_layer(g)
        flat = []
        for in_layer, rs_layer in zip(self.in_layers, self.res_skip_layers):
            flat.extend(self._conv_params(in_layer))
            flat.extend(self._conv_params(rs_layer))
        p_drop = float(self.p_dropout) if self.training else 0.0
        if not hasattr(self, "_pack_plan"):
            self._pack_plan = convops.WNPackPlan(want_planes=True)
        drop_pre, self._drop_pre = getattr(self, "_drop_pre", None), None     # keep-masks drawn ahead by FlowSpecDecoder
        return _wn_apply(x, m2, cond, p_drop, self.dilation_rate, self.n_layers, self._pack_plan, drop_pre, *flat)

    def remove_weight_norm(self):
        if self.gin_channels != 0:
            torch.nn.utils.remove_weight_norm(self.cond_layer)
        for layer in list(self.in_layers) + list(self.res_skip_layers):
            torch.nn.utils.remove_weight_norm(layer)


class ActNorm(nn.Module):
    """Per-channel affine flow with data-dependent initialisation (reference layers.py:173-221)."""

    def __init__(self, channels, ddi=False, **kwargs):
        super().__init__()
        self.channels = channels
        self.initialized = not ddi
        # data-parallel runs: all-reduce the first batch's masked sums over the ranks before initialising (off by default =
        # the reference: every rank uses its own batch and rank 0's result wins at the parameter broadcast)
        self.ddi_all_reduce = os.environ.get("GLOWTTS_DDI_ALLREDUCE", "0") == "1"
        self.logs = nn.Parameter(torch.zeros(1, channels, 1))
        self.bias = nn.Parameter(torch.zeros(1, channels, 1))

    def forward(self, x, x_mask=None, reverse=False, **kwargs):
        if x_mask is None:
            x_mask = _ones_mask(x)
        m2 = ops.mask2d(x_mask)
        if not self.initialized:
            self.initialize(x, x_mask)
            self.initialized = True
        if reverse:
            return ops.actnorm_reverse(x, m2, self.logs, self.bias), None
        x_len = kwargs.get("x_len")
        if x_len is None:
            x_len = ops.mask_len(m2)
        return ops.ActNormFn.apply(x, m2, self.logs, self.bias, x_len)

    def store_inverse(self):
        pass

    def set_ddi(self, ddi):
        self.initialized = not ddi

    def initialize(self, x, x_mask):
        """First-batch statistics -> (logs, bias) so the output is zero-mean / unit-variance per channel."""
        with torch.no_grad():
            s1, s2, count = ops.actnorm_stats(x, ops.mask2d(x_mask))
            if self.ddi_all_reduce and torch.distributed.is_available() and torch.distributed.is_initialized():
                packed = torch.cat([s1.flatten(), s2.flatten(), torch.as_tensor(count, dtype=s1.dtype, device=s1.device).flatten()])
                torch.distributed.all_reduce(packed)
                n = s1.numel()
                s1, s2, count = packed[:n].view_as(s1), packed[n:2 * n].view_as(s2), packed[2 * n:]
            mean = s1 / count
            var = s2 / count - mean * mean
            half_log_var = 0.5 * torch.log(torch.clamp_min(var, 1e-6))
            self.bias.data.copy_((-mean * torch.exp(-half_log_var)).view_as(self.bias))
            self.logs.data.copy_((-half_log_var).view_as(self.logs))


class InvConvNear(nn.Module):
    """Invertible 1x1 convolution shared by C / n_split channel groups (reference layers.py:224-275)."""

    def __init__(self, channels, n_split=4, no_jacobian=False, **kwargs):
        super().__init__()
        assert n_split % 2 == 0
        self.channels, self.n_split, self.no_jacobian = channels, n_split, no_jacobian
        self.weight_inv: typing.Optional[torch.Tensor] = None
        q, _ = torch.linalg.qr(torch.randn(n_split, n_split))
        if torch.det(q) < 0:
            q[:, 0] = -q[:, 0]
        self.weight = nn.Parameter(q.contiguous())

    def forward(self, x, x_mask=None, reverse=False, **kwargs):
        b, c, t = x.size()
        assert c % self.n_split == 0
        if x_mask is None:
            x_mask = _ones_mask(x)
        m2 = ops.mask2d(x_mask)
        if reverse:
            w_inv = self.weight_inv
            if w_inv is None:  # the reference crashes here; computing the inverse is the evident intent
                w_inv = ops.invconv_prepare(self.weight)[0]
            return ops.invconv_apply(x, m2, w_inv.to(x.dtype), self.n_split), None
        x_len = kwargs.get("x_len")
        if x_len is None:
            x_len = ops.mask_len(m2)
        z, logdet = ops.InvConvFn.apply(x, m2, self.weight, x_len, self.n_split)
        if self.no_jacobian:
            logdet = 0
        return z, logdet

    def store_inverse(self):
        self.weight_inv = ops.invconv_prepare(self.weight)[0].to(dtype=self.weight.dtype)
