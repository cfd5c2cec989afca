#!/usr/bin/env python3
"""Launch times of the HBM-bound flow kernels at config-2 shapes (B=32, C=160, T'=400): HIP events over `reps` launches.
Usage: python tools/microbench_flows.py [reps]      (GPU only; tuning)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
import torch  # noqa: E402

from glow_tts_train._hip import call, ptr  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B, C, T = int(os.environ.get("MB_B", "32")), 160, 400
dev = "cuda"
torch.manual_seed(0)
x, dz = torch.randn(B, C, T, device=dev), torch.randn(B, C, T, device=dev)
out = torch.randn(B, C, T, device=dev) * 0.1
m = torch.ones(B, T, device=dev)
logs, bias = torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
w = torch.linalg.qr(torch.randn(4, 4))[0].to(dev).contiguous()
winv, ldw = torch.linalg.inv(w).contiguous(), torch.zeros(1, device=dev)
xlen = m.sum(1)
z, dx, dout = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
logdet, dld = torch.zeros(B, device=dev), torch.randn(B, device=dev)
dlogs, dbias, dw = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(16, device=dev)
ws = torch.zeros(32768, device=dev)
X = x.numel() * 4

cases = {
    "actnorm_invconv_fwd (2X)": (2 * X, lambda: call("glowtts_actnorm_invconv_fwd", ptr(x), ptr(m), ptr(logs), ptr(bias), ptr(w), ptr(ldw), ptr(xlen), ptr(z), ptr(logdet), B, C, T, 4)),
    "actnorm_invconv_bwd (3X)": (3 * X, lambda: call("glowtts_actnorm_invconv_bwd", ptr(x), ptr(m), ptr(logs), ptr(bias), ptr(w), ptr(winv), ptr(dz), ptr(dld), ptr(xlen), ptr(dx), ptr(dlogs), ptr(dbias), ptr(dw), B, C, T, 4)),
    "coupling_fwd (3X)": (3 * X, lambda: call("glowtts_coupling_fwd", ptr(x), ptr(out), ptr(m), ptr(z), ptr(logdet), B, C, T, 0, 0)),
    "coupling_bwd (4X)": (4 * X, lambda: call("glowtts_coupling_bwd", ptr(x), ptr(out), ptr(m), ptr(dz), ptr(dld), ptr(dx), ptr(dout), B, C, T, 0)),
}
for name, (nbytes, fn) in cases.items():
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    print(f"{name:36s} {us:7.2f} us  {nbytes / us / 1e6:7.2f} TB/s")
