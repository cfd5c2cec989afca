#!/usr/bin/env python3
"""What a data-parallel RANK costs before any byte crosses xGMI: the step with the DP reducer's real launch
path in a ONE-rank RCCL group, variant by variant in one process (boxes differ by several percent), with the host time spent
inside the reducer on the backward thread.

  python tools/dp_probe.py [steps_per_block] [blocks]

Variants (alternating blocks of steps):
  none/1   no reducer, gradients in place, one forward chain          (what a rank would cost without any reducer)
  none/2   ... two half-batch forward chains
  inl/1    reducer, collectives issued by the announcing thread        (round 4's path)
  inl/2
  thr/1    reducer, collectives issued by the reducer's launcher thread (round 5 default)
  thr/2
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd")]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from glow_tts_train import convops, parallel  # noqa: E402
from glow_tts_train.train import train_batch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
model, opt, batch, cfg = bench.build_workload(args, dev, 0)

T = {}


def timed(cls, name):
    fn = getattr(cls, name)

    def wrap(self, *a, **k):
        t0 = time.perf_counter()
        try:
            return fn(self, *a, **k)
        finally:
            T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    setattr(cls, name, wrap)


for nm in ("_launch", "_on_announce", "_on_hook", "finish"):
    timed(parallel.FlowBlockReducer, nm)

variants = [("none", 1), ("none", 2), ("inl", 2), ("thr", 1), ("thr", 2)]
state = {"reducer": None, "kind": None}


def select(kind, chains):
    convops._HALF_BATCH_ENV = "1" if chains == 2 else "0"
    convops._HALF_BATCH_FWD = chains == 2
    if state["kind"] == kind:
        return
    if state["reducer"] is not None:
        state["reducer"].remove_hooks()
    convops.set_direct_grads(True if kind == "none" else None)
    state["reducer"] = None if kind == "none" else parallel.FlowBlockReducer(model, opt, force=True, measure=False,
                                                                            comm_thread=(kind == "thr"))
    state["kind"] = kind


res = {v: [] for v in variants}
host = {v: [] for v in variants}
inside = {v: {} for v in variants}
for blk in range(blocks + 1):                      # block 0 = warm-up of every variant
    for v in variants:
        select(*v)
        for _ in range(5):
            train_batch(model, opt, batch, cfg.grad_clip, state["reducer"])
        torch.cuda.synchronize()
        T.clear()
        h = 0.0
        t0 = time.perf_counter()
        for _ in range(n):
            h0 = time.perf_counter()
            train_batch(model, opt, batch, cfg.grad_clip, state["reducer"])
            h += time.perf_counter() - h0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if blk:
            res[v].append(1e3 * dt / n)
            host[v].append(1e3 * h / n)
            for k, s in T.items():
                inside[v].setdefault(k, []).append(1e3 * s / n)
print(f"{'variant':10s} {'ms/step (blocks)':32s} {'host enqueue ms/step':28s} time inside the reducer, ms/step")
for v in variants:
    ins = ", ".join(f"{k} {sum(x) / len(x):.2f}" for k, x in sorted(inside[v].items()))
    print(f"{v[0]}/{v[1]:<6d} {' '.join(f'{x:6.2f}' for x in res[v]):32s} {' '.join(f'{x:6.2f}' for x in host[v]):28s} {ins}", flush=True)

# where the host time goes with the reducer attached (both threads): torch's CPU-side profile of five steps
if os.environ.get("DP_PROBE_PROFILE", "1") == "1":
    from torch.profiler import ProfilerActivity, profile

    for v in (("none", 1), ("thr", 1)):
        select(*v)
        for _ in range(3):
            train_batch(model, opt, batch, cfg.grad_clip, state["reducer"])
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            for _ in range(5):
                train_batch(model, opt, batch, cfg.grad_clip, state["reducer"])
            torch.cuda.synchronize()
        print(f"---- host profile, variant {v[0]}/{v[1]} (5 steps; CPU time of autograd nodes and operators)")
        print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=28, max_name_column_width=60))
if state["reducer"] is not None:
    state["reducer"].remove_hooks()
dist.destroy_process_group()
