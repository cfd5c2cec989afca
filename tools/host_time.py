#!/usr/bin/env python3
"""Host (Python + launch) time of one training step vs its GPU time (tuning tool)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from glow_tts_train.train import train_batch  # noqa: E402

sys.argv = [sys.argv[0]]
args = bench.parse()
model, opt, batch, cfg = bench.build_workload(args, torch.device("cuda:0"), 0)
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

# split: forward / backward / optimizer enqueue time
from glow_tts_train.utils import duration_loss, mle_loss  # noqa: E402
from glow_tts_train._hip import zero_scope, join_side_streams  # noqa: E402
from glow_tts_train.convops import flush_groups  # noqa: E402

x, xl, y, yl, sp = batch
f, b, o = [], [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad()
    with zero_scope(y.device):
        (z, z_m, z_logs, logdet, z_mask), _, (_a, logw, logw_) = model(x, xl, y, yl, g=sp)
        loss = mle_loss(z, z_m, z_logs, logdet, z_mask) + duration_loss(logw, logw_, xl)
        t1 = time.perf_counter()
        loss.backward()
        join_side_streams()
        flush_groups()
    t2 = time.perf_counter()
    flat = opt._optim
    flat.clip_grad_value_(cfg.grad_clip)
    opt.step()
    t3 = time.perf_counter()
    f.append((t1 - t0) * 1e3); b.append((t2 - t1) * 1e3); o.append((t3 - t2) * 1e3)
print("forward enqueue ms :", " ".join(f"{v:.1f}" for v in f))
print("backward enqueue ms:", " ".join(f"{v:.1f}" for v in b))
print("optimizer enqueue ms:", " ".join(f"{v:.2f}" for v in o))
