"""Glow-TTS generator with the reference's public surface (reference: glow_tts_train/models.py):
`FlowGenerator(...)(x, x_lengths, y, y_lengths, g, gen, noise_scale, length_scale)` returning the same three
tuples, `FlowSpecDecoder`, `TextEncoder`, `DurationPredictor`, `setup_model`, `ModelType`, and the same
state-dict keys — computing the flow stack, the alignment search and the losses with the HIP kernels in `csrc/`.
"""
from __future__ import annotations

import logging
import os
import math
import typing

import torch
from torch import nn
from torch.nn import functional as F

from . import _hip, convops, monotonic_align, ops
from .attentions import CouplingBlock, Encoder
from .layers import ActNorm, ConvReluNorm, InvConvNear, LayerNorm
from .optimize import OptimizerType
from .utils import generate_path, sequence_mask, squeeze, unsqueeze

_actnorm_invconv_apply = _hip.direct_apply(ops.ActNormInvConvFn)
_flow_stack_apply = _hip.direct_apply(convops.FlowStackFn)
_FLOW_STACK = __import__("os").environ.get("GLOWTTS_FLOW_STACK", "1") != "0"
_flow_block_apply = _hip.direct_apply(convops.FlowBlockFn)
_align_expand_apply = _hip.direct_apply(ops.AlignExpandFn)
_embed_apply = _hip.direct_apply(convops.EmbedFn)

_LOGGER = logging.getLogger("glow_tts_train.models")


class DurationPredictor(nn.Module):
    """log-duration regressor on the detached encoder output (reference models.py:21-51)."""

    def __init__(self, in_channels, filter_channels, kernel_size, p_dropout):
        super().__init__()
        self.in_channels, self.filter_channels = in_channels, filter_channels
        self.kernel_size, self.p_dropout = kernel_size, p_dropout
        pad = kernel_size // 2
        self.drop = nn.Dropout(p_dropout)
        self.conv_1 = nn.Conv1d(in_channels, filter_channels, kernel_size, padding=pad)
        self.norm_1 = LayerNorm(filter_channels)
        self.conv_2 = nn.Conv1d(filter_channels, filter_channels, kernel_size, padding=pad)
        self.norm_2 = LayerNorm(filter_channels)
        self.proj = nn.Conv1d(filter_channels, 1, 1)

    def forward(self, x, x_mask):
        m2 = ops.mask2d(x_mask)
        p = float(self.drop.p) if (self.training and self.drop.p > 0.0) else 0.0
        for i, (conv, norm) in enumerate(((self.conv_1, self.norm_1), (self.conv_2, self.norm_2))):
            # conv -> ReLU -> LayerNorm -> Dropout: the ReLU and the dropout ride in the norm's kernels (csrc/norm.hip)
            x = norm(convops.conv1d(conv, x, m2, mask_in=True), relu_in=True, p_drop=p, site=f"encoder.proj_w.{i}")
        return convops.conv1d(self.proj, x, m2, mask_in=True, mask_out=True)


class TextEncoder(nn.Module):
    """phoneme ids -> (prior mean, prior log-std, log-durations, mask) (reference models.py:54-142)."""

    def __init__(self, n_vocab, out_channels, hidden_channels, filter_channels, filter_channels_dp, n_heads, n_layers,
                 kernel_size, p_dropout, window_size=None, block_length=None, mean_only=False, prenet=False,
                 gin_channels=0):
        super().__init__()
        self.n_vocab, self.out_channels, self.hidden_channels = n_vocab, out_channels, hidden_channels
        self.filter_channels, self.filter_channels_dp = filter_channels, filter_channels_dp
        self.n_heads, self.n_layers, self.kernel_size, self.p_dropout = n_heads, n_layers, kernel_size, p_dropout
        self.window_size, self.block_length = window_size, block_length
        self.mean_only, self.prenet, self.gin_channels = mean_only, prenet, gin_channels

        self.emb = nn.Embedding(n_vocab, hidden_channels)
        nn.init.normal_(self.emb.weight, 0.0, hidden_channels ** -0.5)
        if prenet:
            self.pre = ConvReluNorm(hidden_channels, hidden_channels, hidden_channels, kernel_size=5, n_layers=3,
                                    p_dropout=0.5)
        self.encoder = Encoder(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout,
                               window_size=window_size, block_length=block_length)
        self.proj_m = nn.Conv1d(hidden_channels, out_channels, 1)
        if not mean_only:
            self.proj_s = nn.Conv1d(hidden_channels, out_channels, 1)
        self.proj_w = DurationPredictor(hidden_channels + gin_channels, filter_channels_dp, kernel_size, p_dropout)
        # every convolution of the encoder (q/k/v/o, FFN, prenet, projections, duration predictor): a handful of weight-packing
        # launches per forward and gradient un-packing launches per backward instead of one each per conv
        # (one group per transformer layer + one for the rest: a layer's gradients are un-packed — and, under data
        # parallelism, its bucket reduced — as soon as its backward is through, not when the whole encoder's is)
        convs_of = lambda mod: [m for m in mod.modules() if isinstance(m, nn.Conv1d)]                     # noqa: E731
        layered = [convs_of(a) + convs_of(f) for a, f in zip(self.encoder.attn_layers, self.encoder.ffn_layers)]
        taken = {id(m) for grp in layered for m in grp}
        rest = [m for m in convs_of(self) if id(m) not in taken]
        # (the transformer layers' groups keep bf16 planes: their 3-tap FFN convolutions run in the selected conv arithmetic)
        enc_planes = os.environ.get("GLOWTTS_ENC_PLANES", "1") != "0"
        self._conv_groups = [convops.ConvGroup(grp, planes=enc_planes) for grp in layered if grp] + ([convops.ConvGroup(rest)] if rest else [])

    # GLOWTTS_CHECK_IDS=1: verify every batch's phoneme ids lie in [0, n_vocab) before the gather (one device -> host sync per
    # step; the reference's nn.Embedding asserts on the device, the gather kernel here clamps — ADVICE r3)
    _check_ids = os.environ.get("GLOWTTS_CHECK_IDS", "0") == "1"

    def forward(self, x, x_lengths, g=None):
        if self._check_ids and x.numel():
            lo, hi = int(x.min()), int(x.max())
            if lo < 0 or hi >= self.n_vocab:
                raise RuntimeError(f"TextEncoder: phoneme id out of range: [{lo}, {hi}] not inside [0, {self.n_vocab})")
        for grp in self._conv_groups:
            grp.begin()
        # [b, h, t]: gather straight into the transposed layout, scaled; segment-sum backward (csrc/train_ops.hip); a CPU tensor
        # raises in the operator like everywhere else on the path
        h = _embed_apply(x, self.emb.weight, math.sqrt(self.hidden_channels))
        x_mask = sequence_mask(x_lengths, h.size(2)).unsqueeze(1).to(h.dtype)
        if self.prenet:
            h = self.pre(h, x_mask)
        h = self.encoder(h, x_mask)
        h_dp = h.detach()
        if g is not None:
            h_dp = torch.cat([h_dp, g.expand(-1, -1, h.size(-1))], 1)
        m2 = ops.mask2d(x_mask)
        x_m = convops.conv1d(self.proj_m, h, m2, mask_out=True)
        x_logs = torch.zeros_like(x_m) if self.mean_only else convops.conv1d(self.proj_s, h, m2, mask_out=True)
        return x_m, x_logs, self.proj_w(h_dp, x_mask), x_mask


class FlowSpecDecoder(nn.Module):
    """squeeze -> n_blocks x [ActNorm, InvConvNear, CouplingBlock] -> unsqueeze (reference models.py:145-215).

    `self.flows` keeps the reference's flat ModuleList (state-dict keys `flows.{3i}`, `{3i+1}`, `{3i+2}`); the
    per-utterance lengths every flow needs for its log-determinant are computed once here and handed down."""

    def __init__(self, in_channels, hidden_channels, kernel_size, dilation_rate, n_blocks, n_layers, p_dropout=0.0,
                 n_split=4, n_sqz=2, sigmoid_scale=False, gin_channels=0):
        super().__init__()
        self.in_channels, self.hidden_channels, self.kernel_size = in_channels, hidden_channels, kernel_size
        self.dilation_rate, self.n_blocks, self.n_layers, self.p_dropout = dilation_rate, n_blocks, n_layers, p_dropout
        self.n_split, self.n_sqz, self.sigmoid_scale, self.gin_channels = n_split, n_sqz, sigmoid_scale, gin_channels
        c = in_channels * n_sqz
        self.flows = nn.ModuleList()
        for _ in range(n_blocks):
            self.flows.extend([
                ActNorm(channels=c),
                InvConvNear(channels=c, n_split=n_split),
                CouplingBlock(c, hidden_channels, kernel_size=kernel_size, dilation_rate=dilation_rate,
                              n_layers=n_layers, gin_channels=gin_channels, p_dropout=p_dropout,
                              sigmoid_scale=sigmoid_scale),
            ])

    # bf16-tensor modes (BASELINE configs[2]; reference: the autocast branch, train.py:116-121).  `io_bf16`:
    #   "all" (or True) — every activation of the flow stack lives in HBM as bf16: the squeezed flow tensor between the flows
    #       and the hidden tensors of every coupling network (start-conv output, the WN stack's layer inputs, gate outputs,
    #       stored tanh / sigmoid, skip sum, and all their gradients);
    #   "hidden" — only the hidden tensors (28 H of the 6.5 C + 28 H elements a block moves per column); the invertible chain
    #       itself stays fp32, as it does under the reference's autocast (its elementwise flows multiply by fp32 parameters).
    # Parameters, masks, (m, logs), log-determinants, parameter gradients and all arithmetic stay fp32 in both.  Measured at
    # B=64 / T=1000 / 12 blocks (tools/bf16_logdet_probe.py): log-det within 1-3e-3 of the fp32 path in either mode — the error
    # is the 2^-9 round-off of the stored hidden tensors, ~5e-3 rms on (m, logs) per block, not the flow tensor's.
    # Set `decoder.io_bf16` (train.train_step(fp16_run=True) sets "all") or GLOWTTS_IO=bf16|bf16-hidden.  Applies to training
    # forwards whose blocks all run as native flow blocks at instantiated shapes; anything else runs the fp32 path.
    io_bf16 = {"bf16": "all", "bf16-hidden": "hidden"}.get(__import__("os").environ.get("GLOWTTS_IO", "fp32"), False)

    def _blocks(self):
        """[(ActNorm, InvConvNear, CouplingBlock), ...] when the flow list has the standard layout, else None."""
        fl = list(self.flows)
        if len(fl) % 3:
            return None
        out = []
        for i in range(0, len(fl), 3):
            if not (isinstance(fl[i], ActNorm) and isinstance(fl[i + 1], InvConvNear) and isinstance(fl[i + 2], CouplingBlock)):
                return None
            out.append((fl[i], fl[i + 1], fl[i + 2]))
        return out

    def _use_bf16(self, x, g, reverse) -> int:
        """0 = fp32 tensors, 1 = hidden tensors bf16, 3 = hidden and flow tensors bf16 (csrc `io` flags)."""
        if not self.io_bf16 or reverse or not x.is_cuda:
            return 0
        blocks = self._blocks()
        if not blocks:
            return 0
        b, c, t = x.shape
        shape = (b, c * self.n_sqz, t // self.n_sqz)
        if not all(convops.flow_block_eligible(a, i, c_, x, g) and convops.flow_block_bf16_ok(c_, shape) for a, i, c_ in blocks):
            return 0
        return 1 if self.io_bf16 == "hidden" else 3

    def _stack_args(self, x, g, io, blocks, m2):
        """Arguments of convops.FlowStackFn when EVERY block of the decoder can share one autograd node: the standard
        [ActNorm, InvConvNear, CouplingBlock] layout, no conditioning input, every block eligible for the native
        executors with one common shape.  GLOWTTS_FLOW_STACK=0 keeps one node per block."""
        if g is not None or not _FLOW_STACK or not blocks:
            return None
        trip = self._blocks()
        if not trip:
            return None
        wn0 = blocks[0].wn
        cfg = None
        bplans, counts, params = [], [], []
        for a, inv, c in trip:
            if not convops.flow_block_eligible(a, inv, c, x, g):
                return None
            wn = c.wn
            k = (inv.n_split, bool(c.sigmoid_scale), float(wn.p_dropout) if wn.training else 0.0, wn.dilation_rate, wn.n_layers,
                 wn.hidden_channels, io)
            if cfg is None:
                cfg = k
            elif k != cfg or wn.kernel_size != wn0.kernel_size:
                return None
            if not hasattr(c, "_block_plan"):
                c._block_plan = convops.FlowBlockPlan()
            pk = convops.flow_block_params(a, inv, c)
            bplans.append(c._block_plan)
            counts.append(len(pk))
            params.extend(pk)
        return cfg, bplans, counts, params

    def forward(self, x, x_mask, g=None, reverse=False):
        io = self._use_bf16(x, g, reverse)
        flow16 = bool(io & 2)
        if self.n_sqz > 1:
            x, x_mask = squeeze(x, x_mask, self.n_sqz, io_bf16=flow16)
        elif flow16:
            x = x.to(torch.bfloat16)
        if reverse:
            for f in reversed(self.flows):
                x, _ = f(x, x_mask, g=g, reverse=True)
            logdet_tot = None
        else:
            x_len = ops.mask_len(ops.mask2d(x_mask))
            logdets = []                               # summed once at the end: one stack + one sum instead of 24 adds
            m2 = ops.mask2d(x_mask)
            # dropout keep-masks of EVERY coupling block's WN stack from one generator launch (12 launches of 20 MB each
            # sat on the forward's dependency chain: 0.26 ms per step); block k takes slice k
            blocks = [f for f in self.flows if isinstance(f, CouplingBlock)]
            masks = None
            if (x.is_cuda and self.training and blocks and self.p_dropout > 0
                    and all(f.wn.n_layers == blocks[0].wn.n_layers and f.wn.hidden_channels == blocks[0].wn.hidden_channels
                            and f.wn.p_dropout == blocks[0].wn.p_dropout and f.wn.training for f in blocks)):
                wn0 = blocks[0].wn
                masks = ops.keep_mask((len(blocks), wn0.n_layers, x.size(0), 2 * wn0.hidden_channels, x.size(2)),
                                      float(wn0.p_dropout), x.device, "decoder.wn")
                for k, f in enumerate(blocks):
                    f.wn._drop_pre = masks[k]
            stack = self._stack_args(x, g, io, blocks, m2)
            if stack is not None:                      # every block in ONE autograd node (convops.FlowStackFn)
                cfg, bplans, counts, params = stack
                for f in blocks:
                    f.wn._drop_pre = None
                # (one parameter as the node's differentiable input, the rest as one opaque argument: convops.ParamPack)
                x, logdet_tot = _flow_stack_apply(x, m2, x_len, masks, cfg, bplans, counts, params[0], convops.ParamPack(params))
                if self.n_sqz > 1:
                    x, x_mask = unsqueeze(x, x_mask, self.n_sqz, io_bf16=flow16)
                elif flow16:
                    x = x.float()
                return x, logdet_tot
            i = 0
            while i < len(self.flows):
                f = self.flows[i]
                nxt = self.flows[i + 1] if i + 1 < len(self.flows) else None
                cpl = self.flows[i + 2] if i + 2 < len(self.flows) else None
                if (isinstance(f, ActNorm) and isinstance(nxt, InvConvNear) and isinstance(cpl, CouplingBlock)
                        and (io or convops.flow_block_eligible(f, nxt, cpl, x, g))):
                    # the whole block [ActNorm, InvConvNear, CouplingBlock] as one autograd node, one native call each way
                    wn = cpl.wn
                    if not hasattr(cpl, "_block_plan"):
                        cpl._block_plan = convops.FlowBlockPlan()
                    drop, wn._drop_pre = getattr(wn, "_drop_pre", None), None
                    cfg = (nxt.n_split, bool(cpl.sigmoid_scale), float(wn.p_dropout) if wn.training else 0.0,
                           wn.dilation_rate, wn.n_layers, wn.hidden_channels, io, i // 3)
                    cond = convops.conv1d(wn.cond_layer, g.contiguous()) if g is not None else None     # (B, 2H * n_layers, 1)
                    x, logdet = _flow_block_apply(x, m2, x_len, drop, cfg, cpl._block_plan, cond,
                                                  *convops.flow_block_params(f, nxt, cpl))
                    i += 3
                elif (isinstance(f, ActNorm) and f.initialized and isinstance(nxt, InvConvNear) and not nxt.no_jacobian
                        and nxt.n_split in (2, 4)):
                    # the two elementwise flows of a block in one pass over the tensor (ops.ActNormInvConvFn)
                    x, logdet = _actnorm_invconv_apply(x, m2, f.logs, f.bias, nxt.weight, x_len, nxt.n_split)
                    i += 2
                else:
                    x, logdet = f(x, x_mask, g=g, reverse=False, x_len=x_len)
                    i += 1
                logdets.append(logdet)
            logdet_tot = torch.stack(logdets, 0).sum(0) if len(logdets) > 1 else (logdets[0] if logdets else 0)
        if self.n_sqz > 1:
            x, x_mask = unsqueeze(x, x_mask, self.n_sqz, io_bf16=flow16)
        elif flow16:
            x = x.float()
        return x, logdet_tot

    def store_inverse(self):
        for f in self.flows:
            f.store_inverse()


class FlowGenerator(nn.Module):
    """Top-level model (reference models.py:218-410)."""

    def __init__(self, n_vocab: int, hidden_channels: int, filter_channels: int, filter_channels_dp: int,
                 out_channels: int, kernel_size: int = 3, n_heads: int = 2, n_layers_enc: int = 6,
                 p_dropout: float = 0.0, n_blocks_dec: int = 12, kernel_size_dec: int = 5, dilation_rate: int = 5,
                 n_block_layers: int = 4, p_dropout_dec: float = 0.0, n_speakers: int = 0, gin_channels: int = 0,
                 n_split: int = 4, n_sqz: int = 1, sigmoid_scale: bool = False,
                 window_size: typing.Optional[int] = None, block_length: typing.Optional[int] = None,
                 mean_only: bool = False, hidden_channels_enc: typing.Optional[int] = None,
                 hidden_channels_dec: typing.Optional[int] = None, prenet: bool = False):
        super().__init__()
        for name, value in list(locals().items()):
            if name not in ("self", "__class__"):
                setattr(self, name, value)
        self.encoder = TextEncoder(n_vocab, out_channels, hidden_channels_enc or hidden_channels, filter_channels,
                                   filter_channels_dp, n_heads, n_layers_enc, kernel_size, p_dropout,
                                   window_size=window_size, block_length=block_length, mean_only=mean_only,
                                   prenet=prenet, gin_channels=gin_channels)
        self.decoder = FlowSpecDecoder(out_channels, hidden_channels_dec or hidden_channels, kernel_size_dec,
                                       dilation_rate, n_blocks_dec, n_block_layers, p_dropout=p_dropout_dec,
                                       n_split=n_split, n_sqz=n_sqz, sigmoid_scale=sigmoid_scale,
                                       gin_channels=gin_channels)
        if n_speakers > 1:
            self.emb_g = nn.Embedding(n_speakers, gin_channels)
            nn.init.uniform_(self.emb_g.weight, -0.1, 0.1)

    # -- helpers ------------------------------------------------------------------------------------------------
    @staticmethod
    def _expand_by_alignment(attn, stats):
        """attn [b,1,t,t'] (0/1), stats [b,d,t] -> [b,d,t']: every frame takes its aligned token's statistics."""
        return torch.matmul(attn.squeeze(1).transpose(1, 2), stats.transpose(1, 2)).transpose(1, 2)

    @staticmethod
    def _pairwise_log_likelihood(x_m, x_logs, z):
        """log N(z_t'; x_m_t, exp(x_logs_t)) for every (token, frame) pair -> [b, t, t'] (reference models.py:362-376)."""
        inv_var = torch.exp(-2 * x_logs)
        const = torch.sum(-0.5 * math.log(2 * math.pi) - x_logs, [1]).unsqueeze(-1)
        quad = torch.matmul(inv_var.transpose(1, 2), -0.5 * (z ** 2))
        cross = torch.matmul((x_m * inv_var).transpose(1, 2), z)
        bias = torch.sum(-0.5 * (x_m ** 2) * inv_var, [1]).unsqueeze(-1)
        return const + quad + cross + bias

    def preprocess(self, y, y_lengths, y_max_length):
        """Floor frame counts to a multiple of n_sqz (reference models.py:401-406)."""
        q = self.n_sqz
        if y_max_length is not None:
            y_max_length = (y_max_length // q) * q
            y = y[:, :, :y_max_length]
        return y, (y_lengths // q) * q, y_max_length

    # -- forward ------------------------------------------------------------------------------------------------
    def forward(self, x, x_lengths, y=None, y_lengths=None, g=None, gen=False, noise_scale=1.0, length_scale=1.0):
        if g is not None:
            g = F.normalize(self.emb_g(g)).unsqueeze(-1)                        # [b, gin, 1]
        # Training: the text encoder and the flow decoder do not depend on each other until the alignment search, and the
        # encoder's kernels (T_text-sized attention / LayerNorm / small convs) leave most of the GPU idle.  It runs on a
        # second stream next to the decoder; autograd replays each branch's backward on the stream of its forward.
        overlap = (not gen) and y is not None and x.is_cuda and _hip.side_stream_enabled()
        if overlap:
            main = torch.cuda.current_stream(x.device)
            side = _hip.side_stream(x.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                x_m, x_logs, logw, x_mask = self.encoder(x, x_lengths, g=g)
        else:
            x_m, x_logs, logw, x_mask = self.encoder(x, x_lengths, g=g)

        if gen:
            w_ceil = torch.ceil(torch.exp(logw) * x_mask * length_scale)
            y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
            y_max_length = None
        else:
            y_max_length = y.size(2)
        y, y_lengths, y_max_length = self.preprocess(y, y_lengths, y_max_length)
        z_mask = sequence_mask(y_lengths, y_max_length).unsqueeze(1).to(x_mask.dtype)

        if gen:
            attn_mask = x_mask.unsqueeze(-1) * z_mask.unsqueeze(2)
            attn = generate_path(w_ceil.squeeze(1), attn_mask.squeeze(1)).unsqueeze(1)
            z_m = self._expand_by_alignment(attn, x_m)
            z_logs = self._expand_by_alignment(attn, x_logs)
            logw_ = torch.log(1e-8 + torch.sum(attn, -1)) * x_mask
            z = (z_m + torch.exp(z_logs) * torch.randn_like(z_m) * noise_scale) * z_mask
            y, logdet = self.decoder(z, z_mask, g=g, reverse=True)
            return (y, z_m, z_logs, logdet, z_mask), (x_m, x_logs, x_mask), (attn, logw, logw_)

        if overlap:
            # y-side preparation above only needs x_mask's dtype; everything that reads encoder outputs comes after the join
            z, logdet = self.decoder(y, z_mask, g=g, reverse=False)
            main.wait_stream(side)
            for t in (x_m, x_logs, logw, x_mask):
                t.record_stream(main)
        else:
            z, logdet = self.decoder(y, z_mask, g=g, reverse=False)
        if x.is_cuda:
            # one contraction for the lattice, the search, then z_m / z_logs by GATHER through the frame -> token map the search
            # kernel hands out (the reference's four bmm's with a one-hot matrix, ~25 launches, become 4)
            mean_only = self.mean_only and not x_logs.requires_grad
            with torch.no_grad():
                logp = ops.align_logp(x_m, None if mean_only else x_logs, z)
                path, first, tok = ops.mas_path_spans(logp, x_lengths, y_lengths)
            attn = path.unsqueeze(1)
            z_m = _align_expand_apply(x_m, tok, first)
            z_logs = torch.zeros_like(z_m) if mean_only else _align_expand_apply(x_logs, tok, first)
            logw_ = ops.span_logw(first, x_lengths).to(x_mask.dtype)    # = log(1e-8 + sum(attn, -1)) * x_mask
            return (z, z_m, z_logs, logdet, z_mask), (x_m, x_logs, x_mask), (attn, logw, logw_)
        attn_mask = x_mask.unsqueeze(-1) * z_mask.unsqueeze(2)
        with torch.no_grad():
            logp = self._pairwise_log_likelihood(x_m, x_logs, z)
            # device-resident search; lengths are what the reference would read back off attn_mask
            attn = monotonic_align.maximum_path_lengths(logp, x_lengths, y_lengths).unsqueeze(1).detach()
        z_m = self._expand_by_alignment(attn, x_m)
        z_logs = self._expand_by_alignment(attn, x_logs)
        logw_ = torch.log(1e-8 + torch.sum(attn, -1)) * x_mask
        return (z, z_m, z_logs, logdet, z_mask), (x_m, x_logs, x_mask), (attn, logw, logw_)

    def store_inverse(self):
        self.decoder.store_inverse()


ModelType = FlowGenerator


def setup_model(config, model: typing.Optional[ModelType] = None, optimizer: typing.Optional[OptimizerType] = None,
                model_factory=ModelType, optimizer_factory=OptimizerType, create_optimizer: bool = True,
                use_cuda: bool = True) -> typing.Tuple[ModelType, typing.Optional[OptimizerType]]:
    """Build (or adopt) the model and its optimizer from a TrainingConfig-shaped object (reference models.py:417-470)."""
    if model is None:
        mc = config.model
        names = ("hidden_channels filter_channels filter_channels_dp kernel_size n_heads n_layers_enc p_dropout "
                 "n_blocks_dec kernel_size_dec dilation_rate n_block_layers p_dropout_dec n_speakers gin_channels "
                 "n_split n_sqz sigmoid_scale window_size block_length mean_only hidden_channels_enc "
                 "hidden_channels_dec prenet").split()
        model = model_factory(n_vocab=mc.num_symbols, out_channels=config.audio.mel_channels,
                              **{n: getattr(mc, n) for n in names})
    if use_cuda:
        model.cuda()
    if create_optimizer and optimizer is None:
        optimizer = optimizer_factory(model.parameters(), scheduler=config.scheduler,
                                      dim_model=config.model.hidden_channels, warmup_steps=config.warmup_steps,
                                      lr=config.learning_rate, betas=config.betas, eps=config.eps)
    return model, optimizer
