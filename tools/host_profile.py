#!/usr/bin/env python3
"""cProfile of the host side of one training step (tuning tool)."""
import cProfile
import os
import pstats
import sys

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
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    train_batch(model, opt, batch, cfg.grad_clip, None)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
