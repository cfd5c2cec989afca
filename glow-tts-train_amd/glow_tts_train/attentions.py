"""Text-encoder transformer stack and the affine coupling flow, with the reference's names, constructor arguments
and state-dict keys (reference: glow_tts_train/attentions.py).
"""
from __future__ import annotations

import math
import typing

import torch
from torch import nn
from torch.nn import functional as F

from . import convops, ops
from ._hip import direct_apply
from .layers import WN, LayerNorm

_coupling_apply = direct_apply(ops.CouplingFn)
_rel_attn_apply = direct_apply(ops.RelAttnFn)
_enc_layer_apply = direct_apply(convops.EncoderLayerFn)
_enc_stack_apply = direct_apply(convops.EncoderStackFn)
_ENC_STACK = __import__("os").environ.get("GLOWTTS_ENC_STACK", "1") != "0"


class Encoder(nn.Module):
    """Post-LN transformer layers over (B, H, T_text) (reference attentions.py:12-74)."""

    def __init__(self, hidden_channels: int, filter_channels: int, n_heads: int, n_layers: int, kernel_size: int = 1,
                 p_dropout: float = 0.0, window_size: typing.Optional[int] = None,
                 block_length: typing.Optional[int] = None):
        super().__init__()
        self.hidden_channels, self.filter_channels = hidden_channels, filter_channels
        self.n_heads, self.n_layers, self.kernel_size = n_heads, n_layers, kernel_size
        self.p_dropout, self.window_size, self.block_length = p_dropout, window_size, block_length
        self.drop = nn.Dropout(p_dropout)
        self.attn_layers = nn.ModuleList(
            MultiHeadAttention(hidden_channels, hidden_channels, n_heads, window_size=window_size, p_dropout=p_dropout,
                               block_length=block_length) for _ in range(n_layers))
        self.norm_layers_1 = nn.ModuleList(LayerNorm(hidden_channels) for _ in range(n_layers))
        self.ffn_layers = nn.ModuleList(
            FFN(hidden_channels, hidden_channels, filter_channels, kernel_size, p_dropout=p_dropout)
            for _ in range(n_layers))
        self.norm_layers_2 = nn.ModuleList(LayerNorm(hidden_channels) for _ in range(n_layers))
        for i, (attn, ffn) in enumerate(zip(self.attn_layers, self.ffn_layers)):       # names of their dropout keep-masks
            attn._site, ffn._site = f"encoder.encoder.{i}.attn", f"encoder.encoder.{i}.ffn"

    def _native_layers(self, x):
        """[(ConvGroup, attn, ffn, norm1, norm2), ...] when EVERY layer can run as one native call each way, else None."""
        layers = []
        for attn, norm1, ffn, norm2 in zip(self.attn_layers, self.norm_layers_1, self.ffn_layers, self.norm_layers_2):
            group = getattr(attn.conv_q, "_glowtts_group", (None, 0))[0]
            if group is None or not convops.encoder_layer_eligible(group, attn, ffn, norm1, norm2, x):
                return None
            layers.append((group, attn, ffn, norm1, norm2))
        return layers

    def forward(self, x, x_mask):
        layers = self._native_layers(x) if x.is_cuda else None
        if layers:
            # one autograd node per layer, one native call each way (csrc/wn_stack.hip); the keep-masks of all four dropouts of
            # all layers come from ONE generator launch
            m2 = ops.mask2d(x_mask)
            b, hch, t = x.shape
            nh, p = self.n_heads, float(self.p_dropout) if self.training else 0.0
            sizes = [b * nh * t * t, b * hch * t, b * self.filter_channels * t, b * hch * t]
            keep = None
            if p > 0.0:
                keep = ops.keep_mask((len(layers) * sum(sizes),), p, x.device, "encoder.layers")
            if _ENC_STACK and len({(f.kernel_size, f.filter_channels, a.window_size, a.heads_share, a.block_length, n.eps)
                                   for _, a, f, n, _ in layers}) == 1:
                # every layer in ONE autograd node (convops.EncoderStackFn)
                _, attn, ffn, norm1, _ = layers[0]
                cfg = (nh, ffn.kernel_size, -1 if attn.window_size is None else attn.window_size, int(attn.heads_share),
                       -1 if attn.block_length is None else attn.block_length, norm1.eps, p)
                counts, params = [], []
                for mods in layers:
                    _, live = convops._enc_layer_table(*mods)
                    counts.append(len(live))
                    params.extend(live)
                return _enc_stack_apply(x, m2, keep, cfg, layers, counts, params[0], convops.ParamPack(params)) * x_mask
            pos = 0
            for group, attn, ffn, norm1, norm2 in layers:
                drops = None
                if keep is not None:
                    drops = []
                    for n in sizes:
                        drops.append(keep[pos: pos + n])
                        pos += n
                    drops = tuple(drops)
                cfg = (nh, ffn.kernel_size, -1 if attn.window_size is None else attn.window_size, int(attn.heads_share),
                       -1 if attn.block_length is None else attn.block_length, norm1.eps, p)
                _, live = convops._enc_layer_table(group, attn, ffn, norm1, norm2)
                x = _enc_layer_apply(x, m2, drops, cfg, (group, attn, ffn, norm1, norm2), *live)
            return x * x_mask
        pair_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
        p = float(self.p_dropout) if self.training else 0.0
        for i, (attn, norm1, ffn, norm2) in enumerate(zip(self.attn_layers, self.norm_layers_1, self.ffn_layers, self.norm_layers_2)):
            x = x * x_mask
            y = attn(x, x, pair_mask)
            y = ops.dropout(y, p, f"encoder.encoder.{i}.y1") if x.is_cuda else self.drop(y)
            x = norm1(x, res=y)                                      # LN(x + y): residual add fused into the norm kernel
            y = ffn(x, x_mask)
            y = ops.dropout(y, p, f"encoder.encoder.{i}.y2") if x.is_cuda else self.drop(y)
            x = norm2(x, res=y)
        return x * x_mask


class CouplingBlock(nn.Module):
    """Affine coupling flow (reference attentions.py:77-145): (m, logs) = end(WN(start(x_0))); z_1 = (m + e^logs x_1) mask.

    The affine apply + log-det reduction is one HIP kernel writing z directly (no slice / exp / mul / cat chain).
    """

    def __init__(self, in_channels, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0,
                 sigmoid_scale=False):
        super().__init__()
        self.in_channels, self.hidden_channels, self.kernel_size = in_channels, hidden_channels, kernel_size
        self.dilation_rate, self.n_layers, self.gin_channels = dilation_rate, n_layers, gin_channels
        self.p_dropout, self.sigmoid_scale = p_dropout, sigmoid_scale
        self.start = torch.nn.utils.weight_norm(nn.Conv1d(in_channels // 2, hidden_channels, 1))
        # zero-initialised last layer: the coupling starts as the identity, which stabilises early training
        self.end = nn.Conv1d(hidden_channels, in_channels, 1)
        nn.init.zeros_(self.end.weight)
        nn.init.zeros_(self.end.bias)
        self.wn = WN(in_channels, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels, p_dropout)
        self._conv_group = convops.ConvGroup([self.start, self.end])   # one pack / un-pack launch for the two 1x1 convs

    def forward(self, x, x_mask=None, reverse: bool = False, g=None, **kwargs):
        if x_mask is None:
            x_mask = torch.ones(x.size(0), 1, x.size(2), device=x.device, dtype=x.dtype)
        m2 = ops.mask2d(x_mask)
        # the 1x1 convs consume the channel slice in place (batch stride C*T) and fold bias and mask into the epilogue
        self._conv_group.begin()
        link = None
        if not reverse and torch.is_grad_enabled() and x.requires_grad and x.is_contiguous():
            link = ops.GradLink()                       # x feeds the start conv AND the affine apply: one gradient buffer
            link.armed = True
            h = convops.conv1d(self.start, x, m2, mask_out=True, link=link)
        else:
            h = convops.conv1d(self.start, x[:, : self.in_channels // 2], m2, mask_out=True)
        h = self.wn(h, x_mask, g, m2=m2)
        out = convops.conv1d(self.end, h, m2)
        if reverse:
            return ops.coupling_reverse(x, out, m2, self.sigmoid_scale), None
        return _coupling_apply(x, out, m2, self.sigmoid_scale, link)

    def store_inverse(self):
        self.wn.remove_weight_norm()


class MultiHeadAttention(nn.Module):
    """Self-attention with windowed relative-position keys/values shared across heads (reference attentions.py:148-344).

    The relative terms are stated by index: scores[i,j] += q_i . E_k[j-i+w] and out_i += p_ij E_v[j-i+w] for |j-i| <= w,
    instead of the reference's pad / reshape skewing of (T, 2T-1) matrices.

    `bf16_mma` (module attribute; GLOWTTS_IO=bf16 or train.train_step(fp16_run=True) set it): the kernel's contractions run
    on the bf16 matrix pipe with fp32 accumulation — the reference's autocast branch (train.py:116-121) computes these
    matmuls in half precision; here q, k, v, the softmax and p_attn stay fp32.
    """
    bf16_mma = __import__("os").environ.get("GLOWTTS_IO", "fp32") == "bf16"

    def __init__(self, channels: int, out_channels: int, n_heads: int, window_size: typing.Optional[int] = None,
                 heads_share: bool = True, p_dropout: float = 0.0, block_length: typing.Optional[int] = None,
                 proximal_bias: bool = False, proximal_init: bool = False):
        super().__init__()
        assert channels % n_heads == 0
        self.channels, self.out_channels, self.n_heads = channels, out_channels, n_heads
        self.window_size, self.heads_share, self.block_length = window_size, heads_share, block_length
        self.proximal_bias, self.p_dropout = proximal_bias, p_dropout
        self.attn = None
        self.k_channels = channels // n_heads
        self.conv_q = nn.Conv1d(channels, channels, 1)
        self.conv_k = nn.Conv1d(channels, channels, 1)
        self.conv_v = nn.Conv1d(channels, channels, 1)
        if window_size is not None:
            n_rel = 1 if heads_share else n_heads
            std = self.k_channels ** -0.5
            self.emb_rel_k = nn.Parameter(torch.randn(n_rel, 2 * window_size + 1, self.k_channels) * std)
            self.emb_rel_v = nn.Parameter(torch.randn(n_rel, 2 * window_size + 1, self.k_channels) * std)
        self.conv_o = nn.Conv1d(channels, out_channels, 1)
        self.drop = nn.Dropout(p_dropout)
        for conv in (self.conv_q, self.conv_k, self.conv_v):
            nn.init.xavier_uniform_(conv.weight)
        if proximal_init:
            self.conv_k.weight.data.copy_(self.conv_q.weight.data)
            self.conv_k.bias.data.copy_(self.conv_q.bias.data)

    def forward(self, x, c, attn_mask=None):
        q, k, v = convops.conv1d(self.conv_q, x), convops.conv1d(self.conv_k, c), convops.conv1d(self.conv_v, c)
        y, self.attn = self.attention(q, k, v, mask=attn_mask)
        return convops.conv1d(self.conv_o, y)

    def _kernel_applicable(self, query, key, mask):
        """The HIP kernels cover self-attention with a (B,1,T,T) mask that is an outer product of a sequence mask
        (what Encoder builds, attentions.py:63), any length (MFMA strip kernels up to 512 tokens, tiled kernels beyond:
        csrc/attention_long.hip), d_k a multiple of 16 (<= 128), window <= 7, no proximal bias."""
        t = key.size(2)
        return (query.is_cuda and query.size(2) == t and self.k_channels % 16 == 0 and self.k_channels <= 128
                and (self.window_size is None or self.window_size <= 7) and not self.proximal_bias and mask is not None
                and mask.dim() == 4 and mask.size(1) == 1)

    def attention(self, query, key, value, mask=None):
        """[b, d, t] tensors -> ([b, d, t], p_attn [b, n_h, t, t])."""
        if self._kernel_applicable(query, key, mask):
            # recover the sequence mask from the pair mask the reference API hands over: m[i] = mask[i, i]
            m2 = torch.diagonal(mask[:, 0], dim1=1, dim2=2).contiguous().float()
            p_drop = float(self.p_dropout) if self.training else 0.0
            ek = self.emb_rel_k if self.window_size is not None else None
            ev = self.emb_rel_v if self.window_size is not None else None
            return _rel_attn_apply(query, key, value, ek, ev, m2, self.n_heads, self.window_size or 0,
                                   self.block_length, p_drop, bool(self.bf16_mma), getattr(self, "_site", "attn"))
        if query.is_cuda and not getattr(MultiHeadAttention, "_warned_general", False):
            MultiHeadAttention._warned_general = True
            import logging
            logging.getLogger("glow_tts_train.attentions").warning(
                "MultiHeadAttention: shape outside the HIP kernels' envelope (T=%d, d_k=%d not a multiple of 16 or > 128, "
                "window=%s > 7, cross-attention or proximal bias): using the composition of PyTorch ops (slower; still on the "
                "GPU)", key.size(2), self.k_channels, self.window_size)
        return self._attention_general(query, key, value, mask)

    def _attention_general(self, query, key, value, mask=None):
        """Shapes outside the kernels' envelope (cross-attention, proximal bias, odd head widths): torch ops, index-based."""
        b, d, t_s = key.size()
        t_t = query.size(2)
        nh, dk, w = self.n_heads, self.k_channels, self.window_size
        q = query.view(b, nh, dk, t_t).transpose(2, 3)
        k = key.view(b, nh, dk, t_s).transpose(2, 3)
        v = value.view(b, nh, dk, t_s).transpose(2, 3)
        scale = 1.0 / math.sqrt(dk)
        scores = torch.matmul(q, k.transpose(-2, -1)) * scale
        if w is not None:
            assert t_s == t_t, "Relative attention is only available for self-attention."
            pos = torch.arange(t_s, device=q.device)
            rel = pos[None, :] - pos[:, None]                       # j - i
            in_win = (rel.abs() <= w)
            rel_idx = (rel + w).clamp_(0, 2 * w)
            qe = torch.matmul(q, self.emb_rel_k.unsqueeze(0).transpose(-2, -1))          # (b, h, t, 2w+1)
            scores = scores + torch.gather(qe, 3, rel_idx.expand(b, nh, t_t, t_s)) * in_win * scale
        if self.proximal_bias:
            assert t_s == t_t, "Proximal bias is only available for self-attention."
            pos = torch.arange(t_s, device=q.device, dtype=scores.dtype)
            scores = scores - torch.log1p((pos[None, :] - pos[:, None]).abs())
        if mask is not None:
            scores = scores.masked_fill(mask == 0, -1e4)
            if self.block_length is not None:
                pos = torch.arange(t_s, device=q.device)
                band = ((pos[None, :] - pos[:, None]).abs() <= self.block_length).to(scores.dtype)
                scores = scores * band + -1e4 * (1 - band)
        p_attn = F.softmax(scores, dim=-1)
        if self.training and self.p_dropout > 0 and p_attn.is_cuda:
            p_attn = ops.dropout(p_attn, float(self.p_dropout), getattr(self, "_site", "attn"))
        else:
            p_attn = self.drop(p_attn)
        out = torch.matmul(p_attn, v)
        if w is not None:
            # weights on the 2w+1 diagonals around the main one: pw[..., i, r] = p[i, i + r - w]
            pos_j = torch.arange(t_s, device=q.device)[:, None] + torch.arange(-w, w + 1, device=q.device)[None, :]
            ok = (pos_j >= 0) & (pos_j < t_s)
            pw = torch.gather(p_attn, 3, pos_j.clamp(0, t_s - 1).expand(b, nh, t_t, 2 * w + 1)) * ok
            out = out + torch.matmul(pw, self.emb_rel_v.unsqueeze(0))
        out = out.transpose(2, 3).contiguous().view(b, d, t_t)
        return out, p_attn


class FFN(nn.Module):
    """Position-wise conv feed-forward (reference attentions.py:347-381)."""

    def __init__(self, in_channels, out_channels, filter_channels, kernel_size, p_dropout=0.0, activation=None):
        super().__init__()
        self.in_channels, self.out_channels, self.filter_channels = in_channels, out_channels, filter_channels
        self.kernel_size, self.p_dropout, self.activation = kernel_size, p_dropout, activation
        self.conv_1 = nn.Conv1d(in_channels, filter_channels, kernel_size, padding=kernel_size // 2)
        self.conv_2 = nn.Conv1d(filter_channels, out_channels, kernel_size, padding=kernel_size // 2)
        self.drop = nn.Dropout(p_dropout)

    def forward(self, x, x_mask):
        m2 = ops.mask2d(x_mask)
        h = convops.conv1d(self.conv_1, x, m2, mask_in=True)                       # conv_1(x * mask)
        h = h * torch.sigmoid(1.702 * h) if self.activation == "gelu" else torch.relu(h)
        if self.training and self.p_dropout > 0 and h.is_cuda:
            h = ops.dropout(h, float(self.p_dropout), getattr(self, "_site", "ffn"))
        return convops.conv1d(self.conv_2, h, m2, mask_in=True, mask_out=True)   # conv_2(h * mask) * mask
