#!/usr/bin/env python3
"""Micro-benchmark of the MFMA conv kernels at config-2 WN shapes (B=32, H=192, T'=400): HIP-event timing per launch.
Usage: python tools/microbench_conv.py [reps]      (GPU only; used for kernel tuning and PMC runs)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
import torch  # noqa: E402

from glow_tts_train import convops  # noqa: E402
from glow_tts_train._hip import call, ptr  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2] if len(sys.argv) > 2 else ""
B, H, T, C = int(os.environ.get("MB_B", "32")), 192, 400, 160
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(B, H, T, device=dev)
m2 = torch.ones(B, T, device=dev)
v_in = torch.randn(2 * H, H, 5, device=dev) * 0.03
g_in = torch.ones(2 * H, 1, 1, device=dev)
b_in = torch.zeros(2 * H, device=dev)
v_rs = torch.randn(2 * H, H, 1, device=dev) * 0.07
wf_in, wb_in, _ = convops.pack_weight(v_in, g_in)
wf_rs, wb_rs, _ = convops.pack_weight(v_rs, None)
acts = torch.empty(B, H, T, device=dev)
ts = torch.empty(B, 2 * H, T, device=dev)
xo = torch.empty(B, H, T, device=dev)
sk = torch.empty(B, H, T, device=dev)
d2 = torch.randn(B, 2 * H, T, device=dev)
dx = torch.empty(B, H, T, device=dev)
dwp5 = torch.zeros(5, H, 2 * H, device=dev)
dwp1 = torch.zeros(1, H, 2 * H, device=dev)
x0 = torch.randn(B, C, T, device=dev)
v_st = torch.randn(H, C // 2, 1, device=dev) * 0.1
wf_st, wb_st, _ = convops.pack_weight(v_st, None)
h = torch.empty(B, H, T, device=dev)

# MB_MATH=bf16x6+wrw (the package default) runs the bf16-plane kernels: planes of every packed weight, all bound at once
from glow_tts_train import _hip  # noqa: E402
_math = os.environ.get("MB_MATH", "bf16x6+wrw")
if _math != "fp32":
    _hip.conv_math(_math)
    _all = [wf_in, wb_in, wf_rs, wb_rs, wf_st, wb_st]
    _flat = torch.cat([w.reshape(-1) for w in _all])
    _views, _o = [], 0
    for w in _all:
        _views.append(_flat[_o: _o + w.numel()])
        _o += w.numel()
    wf_in, wb_in, wf_rs, wb_rs, wf_st, wb_st = (v.view_as(w) for v, w in zip(_views, _all))
    _planes = torch.empty(3 * _flat.numel(), device=dev, dtype=torch.int16)
    call("glowtts_conv_split_weights", ptr(_flat), _flat.numel(), ptr(_planes))
    _hip.conv_bind_planes(_flat, _planes)


def gate():
    call("glowtts_conv_gate_fwd", ptr(x), ptr(wf_in), ptr(b_in), None, None, 1.0, ptr(acts), ptr(ts), B, H, T, 5, 1, 2)


def resskip():
    call("glowtts_conv_res_skip_fwd", ptr(acts), ptr(wf_rs), ptr(b_in), ptr(m2), ptr(x), ptr(sk), ptr(xo), ptr(sk), B, H, T, 0)


def bwd_data5():
    convops.conv_fwd(d2, wb_in, None, None, dx, 2 * H, H, 5, 1, 2, addend=d2[:, :H])


def bwd_data1():
    convops.conv_fwd(d2, wb_rs, None, None, dx, 2 * H, H, 1, 1, 0)


def wrw5():
    call("glowtts_conv_wrw", ptr(x), x.stride(0), ptr(d2), d2.stride(0), None, None, ptr(dwp5), None, B, H, 2 * H, T, 5, 1, 2)


def wrw1():
    call("glowtts_conv_wrw", ptr(x), x.stride(0), ptr(d2), d2.stride(0), None, None, ptr(dwp1), None, B, H, 2 * H, T, 1, 1, 0)


def start():
    convops.conv_fwd(x0[:, : C // 2], wf_st, None, m2, h, C // 2, H, 1, 1, 0, mask_out=True)


GF = {"gate": 2 * 384 * 960 * B * T, "resskip": 2 * 384 * 192 * B * T, "bwd_data5": 2 * 192 * 1920 * B * T,
      "bwd_data1": 2 * 192 * 384 * B * T, "wrw5": 2 * 384 * 960 * B * T, "wrw1": 2 * 384 * 192 * B * T,
      "start": 2 * 192 * 80 * B * T}
for name, fn in [("gate", gate), ("resskip", resskip), ("bwd_data5", bwd_data5), ("bwd_data1", bwd_data1), ("wrw5", wrw5),
                 ("wrw1", wrw1), ("start", start)]:
    if only and only != name:
        continue
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    print(f"{name:10s} {us:8.1f} us  {GF[name] / us / 1e6:7.1f} TFLOP/s  ({100 * GF[name] / us / 1e6 / 157.3:4.1f}% of fp32 MFMA peak)")
