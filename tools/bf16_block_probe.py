#!/usr/bin/env python3
"""Per-tensor error of ONE flow block in the bf16-tensor modes against the fp32 block (tuning tool, GPU only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

from glow_tts_train import convops, models  # noqa: E402

torch.manual_seed(5)
dec = models.FlowSpecDecoder(80, 192, 5, 1, 1, 4, p_dropout=0.0, n_split=4, n_sqz=2).cuda().train()
with torch.no_grad():
    for f in dec.flows:
        if hasattr(f, "end"):
            f.end.weight.normal_(0, 0.02)
        if hasattr(f, "logs"):
            f.logs.normal_(0, 0.1)
            f.bias.normal_(0, 0.1)
for p in dec.parameters():
    p.grad = torch.zeros_like(p)
b, t = 8, 400
y0 = torch.randn(b, 80, t, device="cuda")
mask = torch.ones(b, 1, t, device="cuda")
saved = {}
orig = convops.FlowBlockFn.forward


def spy(ctx, *a):
    out = orig(ctx, *a)
    names = ["x", "m2", "x_len", "y", "h0", "acts", "ts", "skip", "out", "winv", "xs"]
    saved[a[4][-1]] = {n: v.float().clone() for n, v in zip(names, ctx.to_save)}
    saved[a[4][-1]]["z"] = out[0].float().clone()
    saved[a[4][-1]]["logdet"] = out[1].clone()
    return out


convops.FlowBlockFn.forward = staticmethod(spy)
for mode in (False, "hidden", "all"):
    dec.io_bf16 = mode
    dec(y0.clone().requires_grad_(True), mask)
ref = saved[0]
for io in (1, 3):
    print("io", io)
    for n in ("y", "h0", "xs", "acts", "skip", "out", "z", "logdet"):
        a, e = saved[io][n], ref[n]
        d = (a - e).abs()
        print(f"  {n:7s} max|d|/max|e| {float(d.max() / e.abs().max()):.2e}   rms d / rms e {float(d.pow(2).mean().sqrt() / e.pow(2).mean().sqrt()):.2e}   max|e| {float(e.abs().max()):.3g}")
