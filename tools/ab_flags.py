#!/usr/bin/env python3
"""A/B of host-side switches in ONE process (boxes differ by several percent): alternating blocks of training steps.
Usage: python tools/ab_flags.py NAME=VALUE_A,VALUE_B [steps_per_block] [blocks]
  NAME in: boundary (0|1: one launch between the WN stacks of consecutive blocks), stackpack (0|1: one weight-pack and one W^-1 launch for the flow stack), fuseflows (0|1: coupling(k) fused with ActNorm + InvConv(k + 1) in FlowStackFn), enc_wgrad (0|1: encoder weight gradients on the decoder's weight-gradient stream), native (both|fwd: whole-block /
  whole-layer executors vs the per-operator path), io (fp32|all|hidden), fused (0|1: layer-resident WN forward kernel),
  envs (the library's tuning switches, flipped with glowtts_set_knob: envs=GLOWTTS_A:0+GLOWTTS_B:4,GLOWTTS_A:1+GLOWTTS_B:2)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from glow_tts_train import convops  # noqa: E402
from glow_tts_train.train import train_batch  # noqa: E402

name, _, vals = sys.argv[1].partition("=")
vals = vals.split(",")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 4
sys.argv = [sys.argv[0]]
args = bench.parse()
model, opt, batch, cfg = bench.build_workload(args, torch.device("cuda:0"), 0)


def apply(v):
    if name == "enc_wgrad":
        convops._ENC_WGRAD = v == "1"
    elif name == "native":
        convops._WN_NATIVE = v
    elif name == "io":
        model.decoder.io_bf16 = False if v == "fp32" else v
    elif name == "fused":                                   # layer-resident WN forward kernel (csrc/wn_fused.hip) on / off
        from glow_tts_train import _hip
        _hip.wn_fused(v == "1")
    elif name == "fuseflows":                               # FlowStackFn: coupling(k) fused with ActNorm + InvConv(k + 1) on / off
        convops._FUSE_FLOWS = v == "1"
    elif name == "fwdchains":                               # number of part-batch forward chains of FlowStackFn (1, 2, 4)
        convops._HALF_BATCH_FWD, convops._FWD_CHAINS = int(v) > 1, max(int(v), 1)
    elif name == "halfbatch":                               # FlowStackFn forward as two half-batch chains on two streams on / off
        convops._HALF_BATCH_FWD = v == "1"
    elif name == "boundarybwd":                             # the same boundary launch in the backward on / off
        convops._FLOW_BOUNDARY_BWD = v == "1"
    elif name == "boundary":                                # FlowStackFn: end conv(k) + flows + start conv(k + 1) in one launch on / off
        convops._FLOW_BOUNDARY = v == "1"
    elif name == "stackpack":                               # FlowStackFn: one weight-pack launch per stack (1) or one per block (0)
        convops._STACK_PACK = v == "1"
    elif name == "nopack":                                  # TIMING ONLY (wrong numerics): the decoder's weight packs left out
        if not hasattr(convops.WNPackPlan, "_pack_real"):
            convops.WNPackPlan._pack_real = convops.WNPackPlan.pack
        convops.WNPackPlan.pack = (lambda self: None) if v == "1" else convops.WNPackPlan._pack_real
    elif name == "wrw1pipe":                                # software-pipelined 1x1 weight gradient (csrc/convgemm_split.hip) on / off
        from glow_tts_train import _hip
        _hip.set_knob("GLOWTTS_WRW1_PIPE", int(v))
    elif name == "envs":                                    # library tuning switches: envs=K1:a+K2:b,K1:c+K2:d
        for kv in v.split("+"):
            k, _, val = kv.partition(":")
            from glow_tts_train import _hip
            _hip.set_knob(k, int(val))                     # (the library latches its environment once: glowtts_set_knob flips a switch)
    elif name == "chain":                                   # whole step on a high-priority stream (1) or the default stream (0)
        global CHAIN
        CHAIN = v == "1"
    else:
        raise SystemExit("unknown switch " + name)


CHAIN = False
_chain_stream = torch.cuda.Stream(priority=-1)
_tb = train_batch


def train_batch(*a):                                         # noqa: F811
    if not CHAIN:
        return _tb(*a)
    _chain_stream.wait_stream(torch.cuda.default_stream())
    with torch.cuda.stream(_chain_stream):
        out = _tb(*a)
    torch.cuda.default_stream().wait_stream(_chain_stream)
    return out


for _ in range(8):
    train_batch(model, opt, batch, cfg.grad_clip, None)
res = {v: [] for v in vals}
for blk in range(len(vals) * blocks):
    v = vals[blk % len(vals)]
    apply(v)
    for _ in range(3):
        train_batch(model, opt, batch, cfg.grad_clip, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        train_batch(model, opt, batch, cfg.grad_clip, None)
    torch.cuda.synchronize()
    res[v].append(1e3 * (time.perf_counter() - t0) / n)
for v in vals:
    print(f"{name}={v:8s}: " + "  ".join(f"{t:.2f}" for t in res[v]) + f"   mean {sum(res[v]) / len(res[v]):.2f} ms/step")
