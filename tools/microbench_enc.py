#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels at the text encoder's shapes (B=32, T=160)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
import torch  # noqa: E402

from glow_tts_train import convops  # noqa: E402

B, T = 32, 160
dev = "cuda"
torch.manual_seed(0)
for cin, cout, taps in [(192, 768, 3), (768, 192, 3), (192, 192, 1), (192, 192, 5), (192, 80, 1)]:
    x = torch.randn(B, cin, T, device=dev)
    v = torch.randn(cout, cin, taps, device=dev) * 0.05
    wf, wb, _ = convops.pack_weight(v, None)
    y = torch.empty(B, cout, T, device=dev)
    m2 = torch.ones(B, T, device=dev)
    fn = lambda: convops.conv_fwd(x, wf, None, m2, y, cin, cout, taps, 1, (taps - 1) // 2, mask_in=True)   # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 20
    gf = 2.0 * cout * cin * taps * B * T
    print(f"fwd M{cout} K{cin}x{taps}: {us:7.1f} us  {gf / us / 1e6:6.1f} TFLOP/s")
