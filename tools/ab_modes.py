#!/usr/bin/env python3
"""A/B of two arithmetic modes of the WN convolutions in ONE process (boxes differ by several percent): alternating blocks
of training steps.  Usage: python tools/ab_modes.py MODE_A MODE_B [steps_per_block] [blocks]
(modes: glowtts_conv_math names, optionally "@both" / "@off" for the native WN executor setting)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from glow_tts_train import convops  # noqa: E402
from glow_tts_train.train import train_batch  # noqa: E402

modes = sys.argv[1:3]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 4
sys.argv = [sys.argv[0]]
args = bench.parse()
model, opt, batch, cfg = bench.build_workload(args, torch.device("cuda:0"), 0)
for _ in range(8):
    train_batch(model, opt, batch, cfg.grad_clip, None)
res = {m: [] for m in modes}
for blk in range(2 * blocks):
    mode = modes[blk % 2]
    math, _, executor = mode.partition("@")        # "bf16x6+wrw@both": arithmetic @ WN executor (fwd = default, both, off)
    convops.set_conv_math(math)
    convops._WN_NATIVE = executor or "fwd"
    for _ in range(3):
        train_batch(model, opt, batch, cfg.grad_clip, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        train_batch(model, opt, batch, cfg.grad_clip, None)
    torch.cuda.synchronize()
    res[mode].append(1e3 * (time.perf_counter() - t0) / n)
for mode in modes:
    print(f"{mode:18s}: " + "  ".join(f"{v:.2f}" for v in res[mode]) + f"   mean {sum(res[mode]) / len(res[mode]):.2f} ms/step")
