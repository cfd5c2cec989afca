#!/usr/bin/env python3
"""Which stream ends the backward?  HIP events at the end of each stream's work just before train_batch joins the side streams:
how long after the main stream (decoder backward chain) do the encoder stream and the weight-gradient stream finish.
Usage: python tools/tail_probe.py [steps=20]   (bench.py's workload, eager, no synchronisation between steps)"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
import torch  # noqa: E402

import bench  # noqa: E402
from glow_tts_train import _hip, train  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
args = bench.parse() if False else types.SimpleNamespace(batch=32, t_mel=800, t_text=0, blocks=12, speakers=0, gin=0)
dev = torch.device("cuda:0")
_hip.load()
model, opt, batch, cfg = bench.build_workload(args, dev, 0)
rec = []
orig = train.join_side_streams


def probed():
    main = torch.cuda.current_stream(dev)
    ev = {"main": torch.cuda.Event(enable_timing=True)}
    ev["main"].record(main)
    for (d, role), s in _hip._side_streams.items():
        e = torch.cuda.Event(enable_timing=True)
        e.record(s)
        ev[role] = e
    orig()
    e = torch.cuda.Event(enable_timing=True)
    e.record(main)
    ev["joined"] = e
    rec.append(ev)


train.join_side_streams = probed
for _ in range(5):
    train.train_batch(model, opt, batch, cfg.grad_clip, None)
torch.cuda.synchronize()
rec.clear()
t0 = torch.cuda.Event(enable_timing=True)
t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(steps):
    train.train_batch(model, opt, batch, cfg.grad_clip, None)
t1.record()
torch.cuda.synchronize()
print(f"{t0.elapsed_time(t1) / steps:.3f} ms per step")
keys = [k for k in rec[0] if k != "main"]
for k in keys:
    v = sorted(r["main"].elapsed_time(r[k]) for r in rec)
    print(f"  {k:8s} ends {v[len(v) // 2] * 1e3:8.1f} us after the main stream's last backward kernel (median; min {v[0] * 1e3:.1f}, max {v[-1] * 1e3:.1f})")
