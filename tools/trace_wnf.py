#!/usr/bin/env python3
"""Phase timeline of the layer-resident WN forward kernel (csrc/wn_fused.hip; needs the trace build: `make -C glow-tts-train_amd/csrc
trace`).  Usage: GLOWTTS_HIP_LIB=tools/libglowtts_trace.bin python tools/trace_wnf.py [B T n_layers]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
os.environ.setdefault("GLOWTTS_HIP_LIB", os.path.join(ROOT, "tools", "libglowtts_trace.bin"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from glow_tts_train import _hip, convops, layers, ops  # noqa: E402


class Ctx:
    def save_for_backward(self, *a):
        self.saved = a


b, t, nl = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 400, 4)
h, p = 192, 0.05
convops.set_conv_math("bf16x6+wrw")
torch.manual_seed(5)
wn = layers.WN(2 * h, h, kernel_size=5, dilation_rate=1, n_layers=nl, p_dropout=p).cuda().train()
m2 = torch.ones(b, t, device="cuda")
x = torch.randn(b, h, t, device="cuda")
keep = ops.keep_mask((nl, b, 2 * h, t), p, "cuda", "trace")
flat = []
for a, r in zip(wn.in_layers, wn.res_skip_layers):
    flat.extend(wn._conv_params(a))
    flat.extend(wn._conv_params(r))
plan = convops.WNPackPlan(want_planes=True)
lib = _hip.load()
_hip.wn_fused(True)
nwg = b * ((t + 51) // 52)
buf = (ctypes.c_ulonglong * (8192 * 16))()
with torch.no_grad():
    for _ in range(5):
        convops.WNFn.forward(Ctx(), x, m2, None, p, 1, nl, plan, keep, *flat)
    torch.cuda.synchronize()
    lib.glowtts_debug_trace_read_wnf(buf, 8192 * 16, 1)
    convops.WNFn.forward(Ctx(), x, m2, None, p, 1, nl, plan, keep, *flat)
    torch.cuda.synchronize()
lib.glowtts_debug_trace_read_wnf(buf, 8192 * 16, 0)
tr = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 32)[:nwg].astype(np.int64)
t0 = tr[:, 0].min()
us = lambda c: (tr[:, c] - t0) / 100.0               # wall clock: 100 MHz        # noqa: E731
print(f"{nwg} workgroups; start skew {us(0).max():.1f} us; end first/median/last {np.percentile(us(26), [0, 50, 100])}")
print(f"prologue (start -> before first barrier): median {np.median(us(1) - us(0)):.2f} us")
names = ["in-conv loop", "B1 wait + gate", "B2 wait", "res/skip loop", "B3 wait + update"]
for l in range(nl):
    base = 2 + 6 * l
    segs = [us(base + 1) - us(base), us(base + 2) - us(base + 1), us(base + 3) - us(base + 2), us(base + 4) - us(base + 3),
            us(base + 5) - us(base + 4)]
    print(f"layer {l}: " + "  ".join(f"{n} {np.median(s):.2f}" for n, s in zip(names, segs)) + f"   (layer total {np.median(us(base + 5) - us(base)):.2f})")
print(f"layer 1: wait at B1 {np.median(us(30) - us(3 + 6)):.2f} us, gate phase (B1 release -> before B2) {np.median(us(4 + 6) - us(30)):.2f} us")
cyc = (tr[:, 28] - tr[:, 27]).astype(np.float64)
wall = (tr[:, 3] - tr[:, 2]).astype(np.float64) / 100.0
print(f"shader clock inside layer 0's in-conv loop: median {np.median(cyc / wall):.0f} cycles/us; loop = {np.median(cyc):.0f} cycles "
      f"for {30 * 48} MFMAs of wave 0 (the oldest of its SIMD's three waves; the SIMD's 4 320 MFMAs end at the B1 release)")
xcc = tr[:, 29] & 7
print("workgroups per XCC:", np.bincount(xcc, minlength=8).tolist())
