#!/usr/bin/env python3
"""tools/host_time.py with bf16 tensors in HBM (decoder.io_bf16 = "all", attention contractions on the bf16 MFMA)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from glow_tts_train.attentions import MultiHeadAttention  # noqa: E402
from glow_tts_train.train import train_batch  # noqa: E402

sys.argv = [sys.argv[0]]
args = bench.parse()
model, opt, batch, cfg = bench.build_workload(args, torch.device("cuda:0"), 0)
model.decoder.io_bf16 = "all"
for m in model.modules():
    if isinstance(m, MultiHeadAttention):
        m.bf16_mma = True
for _ in range(5):
    train_batch(model, opt, batch, cfg.grad_clip, None)
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_batch(model, opt, batch, cfg.grad_clip, None)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
print("host enqueue ms/step:", " ".join(f"{h:.1f}" for h in host))
print("step (sync to sync) ms:", " ".join(f"{h:.1f}" for h in total))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    train_batch(model, opt, batch, cfg.grad_clip, None)
torch.cuda.synchronize()
print("pipelined ms/step: %.2f" % ((time.perf_counter() - t0) * 1e3 / 20))
