"""glow_oracle.py — CPU ORACLE for the Glow-TTS training hot path (TEST INFRASTRUCTURE).

This file is the checker, not the product.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import it; the shipped package (glow-tts-train_amd/) never does, and fails loudly when
its HIP library is missing instead of falling back to anything here.

It restates, as stateless functions over a flat ``{state_dict_key: tensor}`` mapping (fp32 torch CPU ops, autograd
for the backward, a plain-C MAS in mas_oracle.c), the algorithm of the reference's path
``FlowGenerator.forward() -> mle_loss() -> backward -> clip -> Adam/Noam``.  Each function cites the reference
lines it follows.  Parity pinning: every function below is checked against golden vectors produced by the real
reference in the build container (oracle/make_golden.py -> tests/golden/*.npz; tests/test_oracle_*.py).

Layout conventions are the reference's: activations ``(B, C, T)`` contiguous in T, masks ``(B, 1, T)`` float 0/1,
log-determinants ``(B,)``.
"""
from __future__ import annotations

import ctypes
import math
import os
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]
_HERE = os.path.dirname(os.path.abspath(__file__))


@dataclass
class HParams:
    """Model hyper-parameters (defaults = reference ModelConfig, config.py:37-61; out_channels = AudioConfig.mel_channels)."""

    n_vocab: int = 148
    hidden_channels: int = 192
    filter_channels: int = 768
    filter_channels_dp: int = 256
    out_channels: int = 80
    kernel_size: int = 3
    n_heads: int = 2
    n_layers_enc: int = 6
    n_blocks_dec: int = 12
    kernel_size_dec: int = 5
    dilation_rate: int = 1
    n_block_layers: int = 4
    n_speakers: int = 0
    gin_channels: int = 0
    n_split: int = 4
    n_sqz: int = 2
    sigmoid_scale: bool = False
    window_size: Optional[int] = 4
    block_length: Optional[int] = None
    mean_only: bool = True
    prenet: bool = True


# ---------------------------------------------------------------------------------------------------------
# small helpers
# ---------------------------------------------------------------------------------------------------------
def sequence_mask(lengths: Tensor, max_len: int) -> Tensor:
    """utils.py:52-56 — bool (B, T): position < length."""
    return torch.arange(max_len, device=lengths.device)[None, :] < lengths[:, None]


def squeeze(x: Tensor, mask: Optional[Tensor], n: int) -> Tuple[Tensor, Tensor]:
    """utils.py:135-147 — x[b, c, n*t'+s] -> x_sqz[b, s*C + c, t'];  mask is sub-sampled at phase n-1."""
    b, c, t = x.shape
    t2 = (t // n) * n
    xs = x[:, :, :t2].reshape(b, c, t2 // n, n)           # [b, c, t', s]
    xs = xs.movedim(3, 1).reshape(b, n * c, t2 // n)      # [b, s, c, t'] -> [b, s*C+c, t']
    if mask is None:
        m = torch.ones(b, 1, t2 // n, dtype=x.dtype)
    else:
        m = mask[:, :, n - 1::n]
    return xs * m, m


def unsqueeze(x: Tensor, mask: Optional[Tensor], n: int) -> Tuple[Tensor, Tensor]:
    """utils.py:150-160 — inverse layout shuffle; mask repeated n times per squeezed column."""
    b, c, t = x.shape
    xu = x.reshape(b, n, c // n, t).movedim(1, 3).reshape(b, c // n, t * n)
    if mask is None:
        m = torch.ones(b, 1, t * n, dtype=x.dtype)
    else:
        m = mask.repeat_interleave(n, dim=2)
    return xu * m, m


def weight_norm(v: Tensor, g: Tensor) -> Tensor:
    """torch.nn.utils.weight_norm(dim=0) as used at layers.py:113,125,135 / attentions.py:100:
    w[o] = g[o] * v[o] / ||v[o]||_2 (norm over all dims but 0)."""
    n = v.flatten(1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return v * (g / n)


def _conv_w(sd: SD, prefix: str) -> Tensor:
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"]
    return weight_norm(sd[prefix + ".weight_v"], sd[prefix + ".weight_g"])


def conv1d(sd: SD, prefix: str, x: Tensor, dilation: int = 1) -> Tensor:
    w = _conv_w(sd, prefix)
    k = w.shape[-1]
    return F.conv1d(x, w, sd[prefix + ".bias"], padding=(k * dilation - dilation) // 2, dilation=dilation)


def channel_layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-4) -> Tensor:
    """layers.py:19-28 — normalise over the CHANNEL dim (dim 1), biased variance, eps 1e-4."""
    mean = x.mean(1, keepdim=True)
    var = ((x - mean) ** 2).mean(1, keepdim=True)
    y = (x - mean) * torch.rsqrt(var + eps)
    return y * gamma.view(1, -1, 1) + beta.view(1, -1, 1)


# ---------------------------------------------------------------------------------------------------------
# flows
# ---------------------------------------------------------------------------------------------------------
def actnorm(x: Tensor, mask: Tensor, logs: Tensor, bias: Tensor, reverse: bool = False):
    """layers.py:182-199 — z = (bias + exp(logs)*x)*mask, logdet = sum(logs)*x_len; reverse (x-bias)*exp(-logs)*mask."""
    x_len = mask.sum(dim=(1, 2))
    if reverse:
        return (x - bias) * torch.exp(-logs) * mask, None
    z = (bias + torch.exp(logs) * x) * mask
    return z, logs.sum() * x_len


def actnorm_init_stats(x: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor]:
    """layers.py:207-221 — data-dependent init: masked per-channel mean/var -> (logs, bias) shaped (1, C, 1)."""
    denom = mask.sum(dim=(0, 2))
    m = (x * mask).sum(dim=(0, 2)) / denom
    m_sq = (x * x * mask).sum(dim=(0, 2)) / denom
    v = m_sq - m * m
    half_logv = 0.5 * torch.log(torch.clamp_min(v, 1e-6))
    bias = (-m * torch.exp(-half_logv)).view(1, -1, 1)
    logs = (-half_logv).view(1, -1, 1)
    return logs, bias


def invconv_matrix_apply(x: Tensor, w: Tensor, n_split: int) -> Tensor:
    """layers.py:247-252,267-271 — channel ch = h*(C/2) + g*(n_split/2) + s  (h in {0,1}, g group, s in-half slot)
    is row k = h*(n_split/2)+s of group g;  z[:, k_out, g] = sum_k W[k_out, k] x[:, k, g]."""
    b, c, t = x.shape
    s2 = n_split // 2
    g = c // n_split
    x5 = x.reshape(b, 2, g, s2, t)
    w4 = w.reshape(2, s2, 2, s2)
    z5 = torch.einsum("opqr,bqgrt->bogpt", w4, x5)
    return z5.reshape(b, c, t)


def invconv(x: Tensor, mask: Tensor, w: Tensor, n_split: int, reverse: bool = False):
    """layers.py:238-272 — forward uses W, logdet = logdet(W) * (C/n_split) * x_len (:265); reverse uses W^-1 (:254-258,275)."""
    b, c, t = x.shape
    x_len = mask.sum(dim=(1, 2))
    if reverse:
        w_inv = torch.inverse(w.float()).to(w.dtype)
        return invconv_matrix_apply(x, w_inv, n_split) * mask, None
    logdet = torch.logdet(w) * (c / n_split) * x_len
    return invconv_matrix_apply(x, w, n_split) * mask, logdet


def gate(a: Tensor, b: Tensor, h: int) -> Tensor:
    """utils.py:31-38 — tanh((a+b)[:, :H]) * sigmoid((a+b)[:, H:])."""
    s = a + b
    return torch.tanh(s[:, :h]) * torch.sigmoid(s[:, h:])


class KeepMasks:
    """Injected dropout decisions.  The reference draws its Bernoulli(1 - p) keep decisions from torch's generator inside
    `F.dropout` (layers.py:58,147; attentions.py:67,71,251,379; models.py:45,49); which generator draws them is not part of
    the algorithm, so the oracle takes them as DATA: `site -> (keep {0,1} tensor, p)`, and applies exactly what
    `F.dropout(x, p, training=True)` computes with those decisions, x * keep / (1 - p).  A site that has no entry is not
    dropped (p = 0).  Site names (the order is the reference's call order within one forward):
        encoder.pre.{i}                    ConvReluNorm: after LayerNorm -> ReLU of layer i            (B, H, T_text)
        encoder.encoder.{i}.attn           MultiHeadAttention: p_attn after the softmax               (B, heads, T, T)
        encoder.encoder.{i}.y1 / .y2       Encoder: the attention / FFN branch before the residual add  (B, H, T)
        encoder.encoder.{i}.ffn            FFN: after the activation, before conv_2                    (B, filter, T)
        encoder.proj_w.{i}                 DurationPredictor: after norm_{i+1}                         (B, filter_dp, T)
        decoder.flows.{f}.wn.{l}           WN: the in-layer's output before the gate                   (B, 2H, T')
    `pinned by`: tests/golden/e2e_dropout_train.npz — the real reference run with F.dropout patched to record its masks."""

    def __init__(self, masks=None):
        self.masks = dict(masks or {})
        self.used = set()

    def __call__(self, x: Tensor, site: str) -> Tensor:
        ent = self.masks.get(site)
        if ent is None:
            return x
        keep, p = ent
        assert tuple(keep.shape) == tuple(x.shape), (site, tuple(keep.shape), tuple(x.shape))
        self.used.add(site)
        return x * (keep.to(x.dtype) * (1.0 / (1.0 - p)))


def _no_drop(x: Tensor, site: str) -> Tensor:
    return x


def wn(sd: SD, prefix: str, x: Tensor, mask: Tensor, g: Optional[Tensor], hidden: int, n_layers: int,
       dilation_rate: int, drop=None) -> Tensor:
    """layers.py:138-162 — gated dilated conv stack with residual/skip 1x1s; `drop`: KeepMasks (dropout on x_in, :147)."""
    drop = drop or _no_drop
    out = torch.zeros_like(x)
    g_all = conv1d(sd, prefix + ".cond_layer", g) if g is not None else None
    for i in range(n_layers):
        x_in = conv1d(sd, f"{prefix}.in_layers.{i}", x, dilation=dilation_rate ** i)
        x_in = drop(x_in, f"{prefix}.{i}")
        if g_all is not None:
            g_l = g_all[:, 2 * hidden * i: 2 * hidden * (i + 1)]
        else:
            g_l = torch.zeros_like(x_in)
        acts = gate(x_in, g_l, hidden)
        rs = conv1d(sd, f"{prefix}.res_skip_layers.{i}", acts)
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * mask
            out = out + rs[:, hidden:]
        else:
            out = out + rs
    return out * mask


def coupling(sd: SD, prefix: str, x: Tensor, mask: Tensor, g: Optional[Tensor], hp: HParams, hidden: int,
             reverse: bool = False, drop=None):
    """attentions.py:119-142 — affine coupling; z_0 = x_0 passes through, (m, logs) = end(WN(start(x_0)))."""
    c = x.shape[1]
    x0, x1 = x[:, : c // 2], x[:, c // 2:]
    h = conv1d(sd, prefix + ".start", x0) * mask
    h = wn(sd, prefix + ".wn", h, mask, g, hidden, hp.n_block_layers, hp.dilation_rate, drop)
    out = conv1d(sd, prefix + ".end", h)
    m, logs = out[:, : c // 2], out[:, c // 2:]
    if hp.sigmoid_scale:
        logs = torch.log(1e-6 + torch.sigmoid(logs + 2))
    if reverse:
        z1 = (x1 - m) * torch.exp(-logs) * mask
        return torch.cat([x0, z1], 1), None
    z1 = (m + torch.exp(logs) * x1) * mask
    return torch.cat([x0, z1], 1), (logs * mask).sum(dim=(1, 2))


def flow_decoder(sd: SD, x: Tensor, mask: Tensor, g: Optional[Tensor], hp: HParams, reverse: bool = False,
                 prefix: str = "decoder", drop=None):
    """models.py:193-211 — squeeze -> n_blocks x [ActNorm, InvConvNear, CouplingBlock] -> unsqueeze."""
    hidden = hp.hidden_channels
    if hp.n_sqz > 1:
        x, mask = squeeze(x, mask, hp.n_sqz)
    order = range(3 * hp.n_blocks_dec)
    logdet_tot = None if reverse else 0
    for i in (reversed(order) if reverse else order):
        p = f"{prefix}.flows.{i}"
        kind = i % 3
        if kind == 0:
            x, ld = actnorm(x, mask, sd[p + ".logs"], sd[p + ".bias"], reverse)
        elif kind == 1:
            x, ld = invconv(x, mask, sd[p + ".weight"], hp.n_split, reverse)
        else:
            x, ld = coupling(sd, p, x, mask, g, hp, hidden, reverse, drop)
        if not reverse:
            logdet_tot = logdet_tot + ld
    if hp.n_sqz > 1:
        x, mask = unsqueeze(x, mask, hp.n_sqz)
    return x, logdet_tot


# ---------------------------------------------------------------------------------------------------------
# text encoder
# ---------------------------------------------------------------------------------------------------------
def rel_attention(q: Tensor, k: Tensor, v: Tensor, attn_mask: Optional[Tensor], n_heads: int,
                  emb_rel_k: Optional[Tensor], emb_rel_v: Optional[Tensor], window: Optional[int],
                  block_length: Optional[int], drop=None, site: str = ""):
    """attentions.py:214-264 (+ helpers :266-333) stated by index instead of the pad/reshape skew:
        scores[b,h,i,j] = q_i.k_j/sqrt(d) + [|j-i|<=w] q_i.Ek[j-i+w]/sqrt(d)
        masked_fill(mask==0, -1e4); optional band |j-i|<=block_length kept, rest set to -1e4
        p = softmax_j;  out_i = sum_j p_ij v_j + sum_{|j-i|<=w} p_ij Ev[j-i+w]
    q,k,v: (B, C, T).  Returns ((B, C, T), p (B, h, T, T))."""
    b, c, t = q.shape
    d = c // n_heads
    qh = q.view(b, n_heads, d, t).transpose(2, 3)
    kh = k.view(b, n_heads, d, t).transpose(2, 3)
    vh = v.view(b, n_heads, d, t).transpose(2, 3)
    scores = torch.matmul(qh, kh.transpose(-2, -1)) / math.sqrt(d)
    idx = torch.arange(t)
    rel = idx[None, :] - idx[:, None]                      # j - i
    if window is not None:
        in_win = rel.abs() <= window
        rel_idx = (rel + window).clamp(0, 2 * window)
        qe = torch.matmul(qh, emb_rel_k[0].t())             # (B, h, T, 2w+1)
        rel_logits = torch.gather(qe, 3, rel_idx.expand(b, n_heads, t, t)) * in_win
        scores = scores + rel_logits / math.sqrt(d)
    if attn_mask is not None:
        scores = scores.masked_fill(attn_mask == 0, -1e4)
        if block_length is not None:
            band = (rel.abs() <= block_length).to(scores.dtype)
            scores = scores * band + -1e4 * (1 - band)
    p = F.softmax(scores, dim=-1)
    if drop is not None:
        p = drop(p, site)                                   # attentions.py:251 — dropout on p_attn; the dropped p is returned
    out = torch.matmul(p, vh)
    if window is not None:
        # rel_w[b,h,i,r] = p[b,h,i,i+r-w] (0 outside the sequence)
        pw = torch.zeros(b, n_heads, t, 2 * window + 1, dtype=p.dtype)
        for r in range(2 * window + 1):
            off = r - window
            diag = torch.diagonal(p, offset=off, dim1=2, dim2=3)   # length t-|off|
            if diag.shape[-1] == 0:
                continue
            start = max(0, -off)
            pw[:, :, start:start + diag.shape[-1], r] = diag
        out = out + torch.matmul(pw, emb_rel_v[0])
    out = out.transpose(2, 3).reshape(b, c, t)
    return out, p


def multi_head_attention(sd: SD, prefix: str, x: Tensor, attn_mask: Tensor, hp: HParams, drop=None, site: str = ""):
    """attentions.py:204-212 — 1x1 q/k/v projections, relative attention, 1x1 output projection."""
    q = conv1d(sd, prefix + ".conv_q", x)
    k = conv1d(sd, prefix + ".conv_k", x)
    v = conv1d(sd, prefix + ".conv_v", x)
    ek = sd.get(prefix + ".emb_rel_k")
    ev = sd.get(prefix + ".emb_rel_v")
    o, p = rel_attention(q, k, v, attn_mask, hp.n_heads, ek, ev, hp.window_size if ek is not None else None,
                         hp.block_length, drop, site)
    return conv1d(sd, prefix + ".conv_o", o), p


def ffn(sd: SD, prefix: str, x: Tensor, mask: Tensor, drop=None, site: str = "") -> Tensor:
    """attentions.py:373-381 — conv(k) -> relu -> dropout -> conv(k), masked before each conv and at the end."""
    h = torch.relu(conv1d(sd, prefix + ".conv_1", x * mask))
    if drop is not None:
        h = drop(h, site)
    return conv1d(sd, prefix + ".conv_2", h * mask) * mask


def encoder_stack(sd: SD, prefix: str, x: Tensor, mask: Tensor, hp: HParams, drop=None) -> Tensor:
    """attentions.py:62-74 — post-LN transformer layers (dropout on each branch before its residual add, :67,71)."""
    drop = drop or _no_drop
    attn_mask = mask.unsqueeze(2) * mask.unsqueeze(-1)
    for i in range(hp.n_layers_enc):
        x = x * mask
        y, _ = multi_head_attention(sd, f"{prefix}.attn_layers.{i}", x, attn_mask, hp, drop, f"{prefix}.{i}.attn")
        y = drop(y, f"{prefix}.{i}.y1")
        x = channel_layer_norm(x + y, sd[f"{prefix}.norm_layers_1.{i}.gamma"], sd[f"{prefix}.norm_layers_1.{i}.beta"])
        y = ffn(sd, f"{prefix}.ffn_layers.{i}", x, mask, drop, f"{prefix}.{i}.ffn")
        y = drop(y, f"{prefix}.{i}.y2")
        x = channel_layer_norm(x + y, sd[f"{prefix}.norm_layers_2.{i}.gamma"], sd[f"{prefix}.norm_layers_2.{i}.beta"])
    return x * mask


def prenet(sd: SD, prefix: str, x: Tensor, mask: Tensor, n_layers: int = 3, drop=None) -> Tensor:
    """layers.py:73-80 — 3 x [conv5(x*mask) -> LN -> relu -> dropout] then residual 1x1 projection, masked."""
    drop = drop or _no_drop
    h = x
    for i in range(n_layers):
        h = conv1d(sd, f"{prefix}.conv_layers.{i}", h * mask)
        h = channel_layer_norm(h, sd[f"{prefix}.norm_layers.{i}.gamma"], sd[f"{prefix}.norm_layers.{i}.beta"])
        h = drop(torch.relu(h), f"{prefix}.{i}")
    return (x + conv1d(sd, prefix + ".proj", h)) * mask


def duration_predictor(sd: SD, prefix: str, x: Tensor, mask: Tensor, drop=None) -> Tensor:
    """models.py:41-51 — 2 x [conv(x*mask) -> relu -> LN -> dropout], 1x1 projection."""
    drop = drop or _no_drop
    h = torch.relu(conv1d(sd, prefix + ".conv_1", x * mask))
    h = drop(channel_layer_norm(h, sd[prefix + ".norm_1.gamma"], sd[prefix + ".norm_1.beta"]), prefix + ".0")
    h = torch.relu(conv1d(sd, prefix + ".conv_2", h * mask))
    h = drop(channel_layer_norm(h, sd[prefix + ".norm_2.gamma"], sd[prefix + ".norm_2.beta"]), prefix + ".1")
    return conv1d(sd, prefix + ".proj", h * mask) * mask


def text_encoder(sd: SD, x_ids: Tensor, x_lengths: Tensor, g: Optional[Tensor], hp: HParams, drop=None):
    """models.py:120-142."""
    h = F.embedding(x_ids, sd["encoder.emb.weight"]) * math.sqrt(hp.hidden_channels)
    h = h.transpose(1, 2)
    mask = sequence_mask(x_lengths, h.shape[2]).unsqueeze(1).to(h.dtype)
    if hp.prenet:
        h = prenet(sd, "encoder.pre", h, mask, drop=drop)
    h = encoder_stack(sd, "encoder.encoder", h, mask, hp, drop)
    x_dp = h.detach()
    if g is not None:
        x_dp = torch.cat([x_dp, g.expand(-1, -1, h.shape[-1])], 1)
    x_m = conv1d(sd, "encoder.proj_m", h) * mask
    x_logs = torch.zeros_like(x_m) if hp.mean_only else conv1d(sd, "encoder.proj_s", h) * mask
    logw = duration_predictor(sd, "encoder.proj_w", x_dp, mask, drop)
    return x_m, x_logs, logw, mask


# ---------------------------------------------------------------------------------------------------------
# monotonic alignment search (plain C, mas_oracle.c)
# ---------------------------------------------------------------------------------------------------------
_mas_lib = None


def _load_mas():
    global _mas_lib
    if _mas_lib is None:
        path = os.path.join(_HERE, "libmas_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libmas_oracle.so missing: run `make -C oracle oracle` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        lib.mas_oracle_batch.restype = None
        lib.mas_oracle_batch.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3
        _mas_lib = lib
    return _mas_lib


def mas_numpy(value: np.ndarray, t_xs: np.ndarray, t_ys: np.ndarray) -> np.ndarray:
    """core.pyx:40-45 over C-contiguous fp32 (B, Tx, Ty); returns int32 path.  `value` is NOT modified (copied)."""
    lib = _load_mas()
    v = np.ascontiguousarray(value, dtype=np.float32).copy()
    b, tx, ty = v.shape
    p = np.zeros((b, tx, ty), np.int32)
    t_xs = np.ascontiguousarray(t_xs, dtype=np.int32)
    t_ys = np.ascontiguousarray(t_ys, dtype=np.int32)
    lib.mas_oracle_batch(p.ctypes.data, v.ctypes.data, t_xs.ctypes.data, t_ys.ctypes.data, b, tx, ty)
    return p


def maximum_path(value: Tensor, mask: Tensor) -> Tensor:
    """monotonic_align/__init__.py:6-21 — value*mask, lengths from the mask's first column/row, C kernel, path in value's dtype."""
    v = (value * mask).detach().cpu().numpy().astype(np.float32)
    m = mask.detach().cpu().numpy()
    t_xs = m.sum(1)[:, 0].astype(np.int32)
    t_ys = m.sum(2)[:, 0].astype(np.int32)
    p = mas_numpy(v, t_xs, t_ys)
    return torch.from_numpy(p).to(dtype=value.dtype)


def mas_python_loops(value: np.ndarray, t_x: int, t_y: int) -> np.ndarray:
    """Pure-Python statement of core.pyx:9-35 for tiny cases (cross-check of the C file)."""
    v = value.astype(np.float32).copy()
    path = np.zeros(v.shape, np.int32)
    neg = np.float32(-1e9)
    for y in range(t_y):
        for x in range(max(0, t_x + y - t_y), min(t_x, y + 1)):
            v_cur = neg if x == y else v[x, y - 1]
            if x == 0:
                v_prev = np.float32(0.0) if y == 0 else neg
            else:
                v_prev = v[x - 1, y - 1]
            v[x, y] = np.float32((v_prev if v_prev > v_cur else v_cur) + v[x, y])
    index = t_x - 1
    for y in range(t_y - 1, -1, -1):
        if index < 0:
            break
        path[index, y] = 1
        if index != 0 and y > 0 and (index == y or v[index, y - 1] < v[index - 1, y - 1]):
            index -= 1
    return path


# ---------------------------------------------------------------------------------------------------------
# generator forward (training) / generation, losses, optimiser step
# ---------------------------------------------------------------------------------------------------------
def speaker_embedding(sd: SD, speaker_ids: Optional[Tensor]) -> Optional[Tensor]:
    """models.py:321-322."""
    if speaker_ids is None:
        return None
    return F.normalize(F.embedding(speaker_ids, sd["emb_g.weight"])).unsqueeze(-1)


def align_logp(x_m: Tensor, x_logs: Tensor, z: Tensor) -> Tensor:
    """models.py:362-376 — log N(z_t'; x_m_t, exp(x_logs_t)) for every (text t, frame t') pair -> (B, T_text, T_mel)."""
    s = torch.exp(-2 * x_logs)
    logp1 = torch.sum(-0.5 * math.log(2 * math.pi) - x_logs, [1]).unsqueeze(-1)
    logp2 = torch.matmul(s.transpose(1, 2), -0.5 * (z ** 2))
    logp3 = torch.matmul((x_m * s).transpose(1, 2), z)
    logp4 = torch.sum(-0.5 * (x_m ** 2) * s, [1]).unsqueeze(-1)
    return logp1 + logp2 + logp3 + logp4


def generator_forward(sd: SD, hp: HParams, x_ids: Tensor, x_lengths: Tensor, y: Tensor, y_lengths: Tensor,
                      speaker_ids: Optional[Tensor] = None, attn_override: Optional[Tensor] = None, drop=None):
    """models.py:310-337,361-399 (gen=False).  `attn_override` lets a test inject a fixed alignment, `drop` (KeepMasks)
    the dropout decisions of a training-mode forward."""
    g = speaker_embedding(sd, speaker_ids)
    x_m, x_logs, logw, x_mask = text_encoder(sd, x_ids, x_lengths, g, hp, drop)
    t_max = (y.shape[2] // hp.n_sqz) * hp.n_sqz                      # preprocess, models.py:401-406
    y = y[:, :, :t_max]
    y_lengths = (y_lengths // hp.n_sqz) * hp.n_sqz
    z_mask = sequence_mask(y_lengths, t_max).unsqueeze(1).to(x_mask.dtype)
    attn_mask = x_mask.unsqueeze(-1) * z_mask.unsqueeze(2)
    z, logdet = flow_decoder(sd, y, z_mask, g, hp, reverse=False, drop=drop)
    with torch.no_grad():
        if attn_override is None:
            logp = align_logp(x_m, x_logs, z)
            attn = maximum_path(logp, attn_mask.squeeze(1)).unsqueeze(1)
        else:
            attn = attn_override
    a = attn.squeeze(1).transpose(1, 2)
    z_m = torch.matmul(a, x_m.transpose(1, 2)).transpose(1, 2)
    z_logs = torch.matmul(a, x_logs.transpose(1, 2)).transpose(1, 2)
    logw_ = torch.log(1e-8 + attn.sum(-1)) * x_mask
    return (z, z_m, z_logs, logdet, z_mask), (x_m, x_logs, x_mask), (attn, logw, logw_)


def generate_path(duration: Tensor, mask: Tensor) -> Tensor:
    """utils.py:99-115 — path[b, t, t'] = 1 for cum_dur[t-1] <= t' < cum_dur[t]."""
    b, t_x, t_y = mask.shape
    cum = torch.cumsum(duration, 1)
    ar = torch.arange(t_y, dtype=cum.dtype)
    ind = (ar[None, None, :] < cum[:, :, None]).to(mask.dtype)
    path = ind - F.pad(ind, [0, 0, 1, 0])[:, :-1]
    return path * mask


def generator_generate(sd: SD, hp: HParams, x_ids: Tensor, x_lengths: Tensor, speaker_ids: Optional[Tensor],
                       noise: Tensor, noise_scale: float = 1.0, length_scale: float = 1.0):
    """models.py:326-359 (gen=True) with the Gaussian sample injected by the caller."""
    g = speaker_embedding(sd, speaker_ids)
    x_m, x_logs, logw, x_mask = text_encoder(sd, x_ids, x_lengths, g, hp)
    w = torch.exp(logw) * x_mask * length_scale
    w_ceil = torch.ceil(w)
    y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
    y_lengths = (y_lengths // hp.n_sqz) * hp.n_sqz
    t_max = int(noise.shape[2])
    z_mask = sequence_mask(y_lengths, t_max).unsqueeze(1).to(x_mask.dtype)
    attn_mask = x_mask.unsqueeze(-1) * z_mask.unsqueeze(2)
    attn = generate_path(w_ceil.squeeze(1), attn_mask.squeeze(1)).unsqueeze(1)
    a = attn.squeeze(1).transpose(1, 2)
    z_m = torch.matmul(a, x_m.transpose(1, 2)).transpose(1, 2)
    z_logs = torch.matmul(a, x_logs.transpose(1, 2)).transpose(1, 2)
    logw_ = torch.log(1e-8 + attn.sum(-1)) * x_mask
    z = (z_m + torch.exp(z_logs) * noise * noise_scale) * z_mask
    y, _ = flow_decoder(sd, z, z_mask, g, hp, reverse=True)
    return (y, z_m, z_logs, None, z_mask), (x_m, x_logs, x_mask), (attn, logw, logw_)


def mle_loss(z: Tensor, m: Tensor, logs: Tensor, logdet: Tensor, mask: Tensor) -> Tensor:
    """utils.py:14-23."""
    nll = torch.sum(logs) + 0.5 * torch.sum(torch.exp(-2 * logs) * (z - m) ** 2)
    nll = nll - torch.sum(logdet)
    nll = nll / torch.sum(torch.ones_like(z) * mask)
    return nll + 0.5 * math.log(2 * math.pi)


def duration_loss(logw: Tensor, logw_: Tensor, lengths: Tensor) -> Tensor:
    """utils.py:26-28."""
    return torch.sum((logw - logw_) ** 2) / torch.sum(lengths)


def clip_grad_value(grads, clip_value: float) -> float:
    """utils.py:118-132 — per-tensor L2 norm accumulated into a global norm, then elementwise clamp in place."""
    total = 0.0
    for g in grads:
        total += float(g.norm(2)) ** 2
        g.clamp_(-clip_value, clip_value)
    return total ** 0.5


def noam_lr(step_num: int, dim_model: int, warmup_steps: int, lr: float = 1.0) -> float:
    """optimize.py:32-41."""
    return lr * dim_model ** -0.5 * min(step_num ** -0.5, step_num * warmup_steps ** -1.5)


class AdamNoam:
    """optimize.py:8-64 over a dict of tensors: torch Adam arithmetic (no amsgrad / weight decay) with the Noam
    schedule; the learning rate used by update k (1-based) is noam_lr(k) — the LR is advanced AFTER each update (:53-55)."""

    def __init__(self, params: SD, dim_model: int, warmup_steps: int = 4000, lr: float = 1.0,
                 betas=(0.9, 0.98), eps: float = 1e-9):
        self.params, self.dim_model, self.warmup, self.lr = params, dim_model, warmup_steps, lr
        self.b1, self.b2 = betas
        self.eps = eps
        self.step_num = 1
        self.cur_lr = noam_lr(1, dim_model, warmup_steps, lr)
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    def step(self, grads: SD):
        t = self.step_num
        bc1 = 1 - self.b1 ** t
        bc2 = 1 - self.b2 ** t
        with torch.no_grad():
            for k, p in self.params.items():
                g = grads.get(k)
                if g is None:
                    continue
                self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
                self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
                p.addcdiv_(self.m[k], denom, value=-self.cur_lr / bc1)
        self.step_num += 1
        self.cur_lr = noam_lr(self.step_num, self.dim_model, self.warmup, self.lr)


# ---------------------------------------------------------------------------------------------------------
# random-init parameters (shapes of the reference state dict) and one full CPU training step
# ---------------------------------------------------------------------------------------------------------
def init_state_dict(hp: HParams, seed: int = 1234) -> SD:
    """Random parameters with the reference's state-dict key schema / shapes (values are NOT the reference's
    initialiser draw-for-draw; used for benchmarks and HIP-vs-oracle parity where both sides load the same dict)."""
    gen = torch.Generator().manual_seed(seed)
    sd: SD = {}
    hch, fch, k = hp.hidden_channels, hp.filter_channels, hp.kernel_size

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=gen) * std

    def conv(prefix, cout, cin, ks, wn_=False, std=None):
        std = std if std is not None else (cin * ks) ** -0.5
        w = rn(cout, cin, ks, std=std)
        if wn_:
            sd[prefix + ".weight_v"] = w
            sd[prefix + ".weight_g"] = w.flatten(1).norm(dim=1).view(-1, 1, 1).clone()
        else:
            sd[prefix + ".weight"] = w
        sd[prefix + ".bias"] = rn(cout, std=0.01)

    def ln(prefix, c):
        sd[prefix + ".gamma"] = torch.ones(c)
        sd[prefix + ".beta"] = torch.zeros(c)

    sd["encoder.emb.weight"] = rn(hp.n_vocab, hch, std=hch ** -0.5)
    if hp.prenet:
        for i in range(3):
            conv(f"encoder.pre.conv_layers.{i}", hch, hch, 5)
            ln(f"encoder.pre.norm_layers.{i}", hch)
        conv("encoder.pre.proj", hch, hch, 1, std=0.01)
    d_k = hch // hp.n_heads
    for i in range(hp.n_layers_enc):
        p = f"encoder.encoder.attn_layers.{i}"
        if hp.window_size is not None:
            sd[p + ".emb_rel_k"] = rn(1, 2 * hp.window_size + 1, d_k, std=d_k ** -0.5)
            sd[p + ".emb_rel_v"] = rn(1, 2 * hp.window_size + 1, d_k, std=d_k ** -0.5)
        for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
            conv(f"{p}.{n}", hch, hch, 1)
        ln(f"encoder.encoder.norm_layers_1.{i}", hch)
        conv(f"encoder.encoder.ffn_layers.{i}.conv_1", fch, hch, k)
        conv(f"encoder.encoder.ffn_layers.{i}.conv_2", hch, fch, k)
        ln(f"encoder.encoder.norm_layers_2.{i}", hch)
    conv("encoder.proj_m", hp.out_channels, hch, 1)
    if not hp.mean_only:
        conv("encoder.proj_s", hp.out_channels, hch, 1, std=0.01)
    conv("encoder.proj_w.conv_1", hp.filter_channels_dp, hch + hp.gin_channels, k)
    ln("encoder.proj_w.norm_1", hp.filter_channels_dp)
    conv("encoder.proj_w.conv_2", hp.filter_channels_dp, hp.filter_channels_dp, k)
    ln("encoder.proj_w.norm_2", hp.filter_channels_dp)
    conv("encoder.proj_w.proj", 1, hp.filter_channels_dp, 1)
    c = hp.out_channels * hp.n_sqz
    for blk in range(hp.n_blocks_dec):
        p = f"decoder.flows.{3 * blk}"
        sd[p + ".logs"] = rn(1, c, 1, std=0.05)
        sd[p + ".bias"] = rn(1, c, 1, std=0.05)
        q, _ = torch.linalg.qr(rn(hp.n_split, hp.n_split))
        if torch.det(q) < 0:
            q[:, 0] = -q[:, 0]
        sd[f"decoder.flows.{3 * blk + 1}.weight"] = q.contiguous()
        p = f"decoder.flows.{3 * blk + 2}"
        conv(p + ".start", hch, c // 2, 1, wn_=True)
        conv(p + ".end", c, hch, 1, std=0.01)
        for l in range(hp.n_block_layers):
            conv(f"{p}.wn.in_layers.{l}", 2 * hch, hch, hp.kernel_size_dec, wn_=True)
            rs = 2 * hch if l < hp.n_block_layers - 1 else hch
            conv(f"{p}.wn.res_skip_layers.{l}", rs, hch, 1, wn_=True)
        if hp.gin_channels:
            conv(p + ".wn.cond_layer", 2 * hch * hp.n_block_layers, hp.gin_channels, 1, wn_=True)
    if hp.n_speakers > 1:
        sd["emb_g.weight"] = (torch.rand(hp.n_speakers, hp.gin_channels, generator=gen) - 0.5) * 0.2
    return sd


def train_step(sd: SD, hp: HParams, opt: AdamNoam, batch, grad_clip: float = 5.0, drop=None, attn_out=None):
    """train.py:106-151 on CPU: forward, mle + duration loss, backward, clamp, Adam/Noam.  `sd` leaves must have
    requires_grad=True.  `drop`: KeepMasks of a training-mode step; `attn_out`: a list that receives the alignment.
    Returns (loss, mel_frames)."""
    x_ids, x_lengths, y, y_lengths, spk = batch
    for p in sd.values():
        p.grad = None
    (z, z_m, z_logs, logdet, z_mask), _, (attn, logw, logw_) = generator_forward(sd, hp, x_ids, x_lengths, y,
                                                                                 y_lengths, spk, drop=drop)
    if attn_out is not None:
        attn_out.append(attn.detach())
    loss = mle_loss(z, z_m, z_logs, logdet, z_mask) + duration_loss(logw, logw_, x_lengths)
    loss.backward()
    grads = {k: p.grad for k, p in sd.items() if p.grad is not None}
    clip_grad_value(grads.values(), grad_clip)
    opt.step(grads)
    return float(loss), int(y_lengths.sum())
