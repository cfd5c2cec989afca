#!/usr/bin/env python3
"""Attention kernels alone, fp32 MFMA vs bf16 MFMA contractions (BASELINE config 2's text encoder: B=32, 2 heads, d_k=96, T=160;
config 5: T=240).  Usage: python tools/microbench_attn.py [B] [T] [channels]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd")]
import torch  # noqa: E402

from glow_tts_train import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 160
C = int(sys.argv[3]) if len(sys.argv) > 3 else 192
H, w = 2, 4
dk = C // H
torch.manual_seed(0)
q, k, v = (torch.randn(B, C, T, device="cuda", requires_grad=True) for _ in range(3))
ek, ev = (torch.randn(1, 2 * w + 1, dk, device="cuda", requires_grad=True) * dk ** -0.5 for _ in range(2))
ek, ev = ek.detach().requires_grad_(True), ev.detach().requires_grad_(True)
lens = torch.randint(T // 2, T + 1, (B,), device="cuda")
m2 = (torch.arange(T, device="cuda")[None] < lens[:, None]).float()
r = torch.randn(B, C, T, device="cuda")


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


flops_f = 2 * B * H * T * T * dk * 2            # q k^T and P v (relative terms: + 2 * (2w+1) / T of that)
for flag in (False, True):
    out = [None]

    def fwd():
        with torch.no_grad():
            out[0] = ops.RelAttnFn.apply(q, k, v, ek, ev, m2, H, w, None, 0.0, flag)

    o, _ = ops.RelAttnFn.apply(q, k, v, ek, ev, m2, H, w, None, 0.0, flag)

    def bwd():
        torch.autograd.grad(o, (q, k, v, ek, ev), r, retain_graph=True)

    tf, tb = timed(fwd), timed(bwd)
    print(f"B={B} T={T} d_k={dk} {'bf16' if flag else 'fp32'} MFMA: fwd {tf:.1f} us ({flops_f / tf / 1e6:.1f} TFLOP/s)   "
          f"bwd (3 kernels + torch glue) {tb:.1f} us")
