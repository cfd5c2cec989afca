#!/usr/bin/env python3
"""Where does the log-det error of the bf16-tensor flow stack come from?  (tuning tool, GPU only)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

from glow_tts_train import convops, models  # noqa: E402

std = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
torch.manual_seed(5)
dec = models.FlowSpecDecoder(80, 192, 5, 1, blocks, 4, p_dropout=0.0, n_split=4, n_sqz=2).cuda().train()
with torch.no_grad():
    for f in dec.flows:
        if hasattr(f, "end"):
            f.end.weight.normal_(0, std)
        if hasattr(f, "logs"):
            f.logs.normal_(0, 0.1)
            f.bias.normal_(0, 0.1)
for p in dec.parameters():
    p.grad = torch.zeros_like(p)
b, t = 64, 1000
torch.manual_seed(8)
y0 = torch.randn(b, 80, t, device="cuda")
lens = torch.linspace(t, t // 2, b, device="cuda").long() // 2 * 2
mask = (torch.arange(t, device="cuda")[None] < lens[:, None]).float().unsqueeze(1)
y0 = y0 * mask


def run(io, math="fp32"):
    dec.io_bf16 = io
    convops.set_conv_math(math)
    z, ld = dec(y0.clone().requires_grad_(True), mask)
    convops.set_conv_math("fp32")
    dec.io_bf16 = False
    return z.detach(), ld.detach()


z0, l0 = run(False)
for name, (z, l) in {"bf16 hidden + flow tensors": run("all"), "bf16 hidden tensors": run("hidden"),
                     "fp32 tensors, bf16 WN arithmetic": run(False, "bf16")}.items():
    d = (l - l0)
    print(f"{name:34s} max|dl|/max|l| {float(d.abs().max() / l0.abs().max()):.2e}   mean dl {float(d.mean()):+.3f}  "
          f"std dl {float(d.std()):.3f}   max|l0| {float(l0.abs().max()):.1f}  per-element {float((d.abs() / (80 * lens)).max()):.2e}  "
          f"z rel {float((z - z0).abs().max() / z0.abs().max()):.2e}")
print("l0[:6]", [round(float(v), 2) for v in l0[:6]], " d[:6]", [round(float(v), 3) for v in (run("hidden")[1] - l0)[:6]])
