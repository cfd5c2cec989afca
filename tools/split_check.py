#!/usr/bin/env python3
"""Accuracy (against an fp64 reference) and launch time of the WN convolutions in every arithmetic mode of
glowtts_conv_math: native fp32 MFMA, bf16, bf16x3, bf16x6 — at config-2 shapes (B=32, H=192, T'=400).
Usage: python tools/split_check.py [reps]      (GPU only)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from glow_tts_train import _hip, convops  # noqa: E402
from glow_tts_train._hip import call, ptr  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, H, T = int(os.environ.get("MB_B", "32")), 192, 400
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(B, H, T, device=dev)
m2 = torch.ones(B, T, device=dev)
v_in = torch.randn(2 * H, H, 5, device=dev) * 0.03
b_in = torch.randn(2 * H, device=dev) * 0.1
v_rs = torch.randn(2 * H, H, 1, device=dev) * 0.07
wf_in, wb_in, _ = convops.pack_weight(v_in, None)
wf_rs, wb_rs, _ = convops.pack_weight(v_rs, None)
acts_in = torch.randn(B, H, T, device=dev) * 0.5
skip_in = torch.randn(B, H, T, device=dev)
d2 = torch.randn(B, 2 * H, T, device=dev)


def gate():
    acts = torch.empty(B, H, T, device=dev)
    ts = torch.empty(B, 2 * H, T, device=dev)
    call("glowtts_conv_gate_fwd", ptr(x), ptr(wf_in), ptr(b_in), None, None, 1.0, ptr(acts), ptr(ts), B, H, T, 5, 1, 2)
    return acts


def gate_ref():
    pre = F.conv1d(x.double(), v_in.double(), b_in.double(), padding=2)
    return torch.tanh(pre[:, :H]) * torch.sigmoid(pre[:, H:])


def resskip():
    xo = torch.empty(B, H, T, device=dev)
    sk = torch.empty(B, H, T, device=dev)
    call("glowtts_conv_res_skip_fwd", ptr(acts_in), ptr(wf_rs), ptr(b_in), ptr(m2), ptr(x), ptr(skip_in), ptr(xo), ptr(sk),
         B, H, T, 0)
    return torch.cat([xo, sk], 1)


def resskip_ref():
    rs = F.conv1d(acts_in.double(), v_rs.double(), b_in.double())
    return torch.cat([x.double() + rs[:, :H], skip_in.double() + rs[:, H:]], 1)


def bwd_data5():
    dx = torch.empty(B, H, T, device=dev)
    convops.conv_fwd(d2, wb_in, None, None, dx, 2 * H, H, 5, 1, 2, addend=skip_in)
    return dx


def bwd_data5_ref():
    return F.conv_transpose1d(d2.double(), v_in.double(), padding=2) + skip_in.double()


def wrw5():
    dwp = torch.zeros(5, H, 2 * H, device=dev)
    db = torch.zeros(2 * H, device=dev)
    call("glowtts_conv_wrw", ptr(x), x.stride(0), ptr(d2), d2.stride(0), None, None, ptr(dwp), ptr(db), B, H, 2 * H, T, 5, 1, 2)
    return torch.cat([dwp.reshape(-1), db])


def wrw5_ref():
    dw = torch.nn.grad.conv1d_weight(x.double(), (2 * H, H, 5), d2.double(), padding=2)
    return torch.cat([dw.permute(2, 1, 0).reshape(-1), d2.double().sum((0, 2))])


def wrw1():
    dwp = torch.zeros(1, H, 2 * H, device=dev)
    call("glowtts_conv_wrw", ptr(acts_in), acts_in.stride(0), ptr(d2), d2.stride(0), ptr(m2), None, ptr(dwp), None, B, H, 2 * H, T,
         1, 1, 0)
    return dwp.reshape(-1)


def wrw1_ref():
    return torch.einsum("bot,bct->co", d2.double(), acts_in.double()).reshape(-1)


CASES = [("gated in-conv k=5", gate, gate_ref, (wf_in,)), ("res/skip 1x1", resskip, resskip_ref, (wf_rs,)),
         ("backward-data k=5", bwd_data5, bwd_data5_ref, (wb_in,)), ("weight grad k=5", wrw5, wrw5_ref, ()),
         ("weight grad 1x1", wrw1, wrw1_ref, ())]


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, fn, ref_fn, weights in CASES:
    ref = ref_fn()
    scale = ref.abs().max()
    for mode in ("fp32", "bf16x6", "bf16x3", "bf16"):
        _hip.conv_math(mode + "+wrw")
        for w in weights:            # one buffer per case: split it and bind its planes to this thread
            planes = torch.empty(3 * w.numel(), device=dev, dtype=torch.int16)
            call("glowtts_conv_split_weights", ptr(w), w.numel(), ptr(planes))
            _hip.conv_bind_planes(w, planes)
        out = fn().double()
        err = (out - ref).abs()
        us = timed(fn)
        print(f"{name:20s} {mode:7s} {us:8.1f} us   max|err|/max|ref| {float(err.max() / scale):.3e}   "
              f"rms err/rms ref {float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.3e}", flush=True)
    _hip.conv_bind_planes(None)
    _hip.conv_math("fp32")

# ---- weight gradient from operands split ONCE (glowtts_split_planes + glowtts_conv_wrw_planes), 3 planes
xp = torch.empty(3 * x.numel(), device=dev, dtype=torch.int16)
dp = torch.empty(3 * d2.numel(), device=dev, dtype=torch.int16)
ap = torch.empty(3 * acts_in.numel(), device=dev, dtype=torch.int16)
dwp5 = torch.zeros(5, H, 2 * H, device=dev)
db5 = torch.zeros(2 * H, device=dev)
dwp1 = torch.zeros(1, H, 2 * H, device=dev)


def split_xd():
    call("glowtts_split_planes", ptr(x), x.numel(), ptr(xp), 3)
    call("glowtts_split_planes", ptr(d2), d2.numel(), ptr(dp), 3)


def wrw5_planes():
    call("glowtts_conv_wrw_planes", ptr(xp), x.numel(), H * T, ptr(dp), d2.numel(), 2 * H * T, ptr(dwp5), ptr(db5), B, H, 2 * H, T, 5, 3)


def wrw1_planes():
    call("glowtts_conv_wrw_planes", ptr(ap), acts_in.numel(), H * T, ptr(dp), d2.numel(), 2 * H * T, ptr(dwp1), None, B, H, 2 * H, T, 1, 3)


split_xd()
call("glowtts_split_planes", ptr(acts_in), acts_in.numel(), ptr(ap), 3)
dwp5.zero_(); db5.zero_(); dwp1.zero_()
wrw5_planes(); wrw1_planes()
for name, got, ref in (("weight grad k=5 from planes", torch.cat([dwp5.reshape(-1), db5]).double(), wrw5_ref()),
                       ("weight grad 1x1 from planes", dwp1.reshape(-1).double(), wrw1_ref())):
    err = (got - ref).abs()
    print(f"{name:28s} max|err|/max|ref| {float(err.max() / ref.abs().max()):.3e}   rms err/rms ref "
          f"{float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.3e}")
print(f"split x (9.8 MB) + d (19.7 MB) into planes: {timed(split_xd):.1f} us;  k=5 from planes: {timed(wrw5_planes):.1f} us;  "
      f"1x1 from planes: {timed(wrw1_planes):.1f} us")
