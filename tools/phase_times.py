#!/usr/bin/env python3
"""GPU time of the phases of a training step under the real multi-stream schedule (events on the main stream)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from glow_tts_train import models as M  # noqa: E402
from glow_tts_train._hip import join_side_streams, zero_scope  # noqa: E402
from glow_tts_train.convops import flush_groups  # noqa: E402
from glow_tts_train.train import train_batch  # noqa: E402
from glow_tts_train.utils import duration_loss, mle_loss  # noqa: E402

sys.argv = [sys.argv[0]]
args = bench.parse()
model, opt, batch, cfg = bench.build_workload(args, torch.device("cuda:0"), 0)
for _ in range(8):
    train_batch(model, opt, batch, cfg.grad_clip, None)
x, xl, y, yl, sp = batch
marks = {}
orig_dec = model.decoder.forward


def dec_forward(*a, **k):
    marks["dec0"] = torch.cuda.Event(enable_timing=True); marks["dec0"].record()
    out = orig_dec(*a, **k)
    marks["dec1"] = torch.cuda.Event(enable_timing=True); marks["dec1"].record()
    return out


model.decoder.forward = dec_forward
rows = []
side_rows = []
for _ in range(12):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    ev[0].record()
    opt.zero_grad()
    with zero_scope(y.device):
        (z, z_m, z_logs, logdet, z_mask), _, (_a, logw, logw_) = model(x, xl, y, yl, g=sp)
        loss = mle_loss(z, z_m, z_logs, logdet, z_mask) + duration_loss(logw, logw_, xl)
        ev[1].record()
        loss.backward()
        ev[2].record()
        from glow_tts_train import _hip as H
        side_ev = {}
        for key, st in H._side_streams.items():
            e = torch.cuda.Event(enable_timing=True)
            e.record(st)
            side_ev[key[1]] = e
        join_side_streams()
        flush_groups()
        ev[3].record()
    opt._optim.clip_grad_value_(cfg.grad_clip)
    opt.step()
    ev[4].record()
    torch.cuda.synchronize()
    side_rows.append({k: ev[1].elapsed_time(e) for k, e in side_ev.items()})
    rows.append([ev[0].elapsed_time(marks["dec0"]), marks["dec0"].elapsed_time(marks["dec1"]), marks["dec1"].elapsed_time(ev[1]),
                 ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3]), ev[3].elapsed_time(ev[4]), ev[0].elapsed_time(ev[4])])
import numpy as np  # noqa: E402
r = np.median(np.array(rows[2:]), axis=0)
for n, v in zip(["zero_grad+enc launch .. decoder start", "decoder forward (main)", "join enc + MAS + losses", "backward (main chain)",
                 "wait side streams", "clip + adam", "TOTAL (sync per step)"], r):
    print(f"{n:40s} {v:7.2f} ms")
for k in side_rows[-1]:
    print(f"backward start -> end of '{k}' stream queue   {np.median([r[k] for r in side_rows[2:]]):7.2f} ms")
