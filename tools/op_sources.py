#!/usr/bin/env python3
"""Which Python lines launch the small ATen kernels of a training step (torch.profiler with stacks; tuning tool).
Usage: python tools/op_sources.py [op-substring ...]   default ops: add fill_ zero_ copy_ mul sum cat"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd")]
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from glow_tts_train.train import train_batch  # noqa: E402

ops = sys.argv[1:] or ["aten::add", "aten::fill_", "aten::zero_", "aten::copy_", "aten::mul", "aten::sum", "aten::cat", "aten::div",
                       "aten::sub", "aten::exp", "aten::where", "aten::masked_fill"]
sys.argv = [sys.argv[0]]
args = bench.parse()
model, opt, batch, cfg = bench.build_workload(args, torch.device("cuda:0"), 0)
for _ in range(3):
    train_batch(model, opt, batch, cfg.grad_clip, None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_batch(model, opt, batch, cfg.grad_clip, None)
    torch.cuda.synchronize()
count = collections.Counter()
for ev in prof.events():
    if any(ev.name == o or ev.name.startswith(o + "_") and False for o in ops) or ev.name in ops:
        if not ev.stack:
            count[(ev.name, "<backward engine / no python frame>")] += 1
            continue
        site = next((f for f in ev.stack if "glow" in f or "oracle" in f or "bench" in f), ev.stack[0])
        count[(ev.name, site[-110:])] += 1
for (name, site), n in count.most_common(60):
    print(f"{n:4d} {name:16s} {site}")
