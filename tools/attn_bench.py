#!/usr/bin/env python3
"""Per-kernel timing of the relative-position attention forward / backward at the encoder's config-2 shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
import torch  # noqa: E402

from glow_tts_train import _hip, attentions  # noqa: E402

torch.manual_seed(0)
mha = attentions.MultiHeadAttention(192, 192, 2, window_size=4, p_dropout=0.1).cuda().train()
B, T = 32, 160
x = torch.randn(B, 192, T, device="cuda", requires_grad=True)
mask = torch.ones(B, 1, T, T, device="cuda")
go = torch.randn(B, 192, T, device="cuda")


def step():
    y = mha(x, x, attn_mask=mask)
    y.backward(go)


for _ in range(3):
    step()
torch.cuda.synchronize()
_hip.enable_timing()
for _ in range(10):
    step()
t = _hip.disable_timing()
for k, v in t.items():
    if "attn" in k:
        print(k, round(1e3 * sum(v) / len(v), 1), "us")
