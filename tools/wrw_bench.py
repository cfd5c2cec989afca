#!/usr/bin/env python3
"""Weight-gradient kernel alone at the benchmark's shape: time per launch (HIP events around back-to-back launches on one
stream) and error against fp64.   GLOWTTS_WRW_TR=0 python tools/wrw_bench.py  selects the frame-packed kernel for A/B."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "glow-tts-train_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from glow_tts_train import _hip, convops  # noqa: E402
from glow_tts_train._hip import call, ptr  # noqa: E402


def main():
    _hip.load()
    shapes = [(32, 192, 384, 400, 5), (48, 192, 384, 600, 5), (64, 192, 384, 500, 5), (8, 192, 384, 200, 5), (32, 192, 384, 400, 1)]
    if os.environ.get("WRW_BENCH_TAPS"):
        shapes = [sh for sh in shapes + [(32, 192, 192, 400, 1), (32, 192, 192, 160, 1)] if sh[4] == int(os.environ["WRW_BENCH_TAPS"])]
    for mode in ("bf16x6+wrw", "fp32")[:1 if os.environ.get("WRW_BENCH_TAPS") else 2]:
        convops.set_conv_math(mode)
        for (b, k, m, t, taps) in shapes:
            torch.manual_seed(1)
            x = torch.randn(b, k, t, device="cuda")
            d = torch.randn(b, m, t, device="cuda")
            lens = torch.linspace(t, t // 2, b).long()
            mask = (torch.arange(t)[None] < lens[:, None]).float().cuda()
            dwp = torch.zeros(taps, k, m, device="cuda")
            db = torch.zeros(m, device="cuda")

            def run(with_mask):
                call("glowtts_conv_wrw", ptr(x), x.stride(0), ptr(d), d.stride(0), ptr(mask) if with_mask else None, None, ptr(dwp),
                     ptr(db), b, k, m, t, taps, 1, (taps - 1) // 2)

            run(True)
            torch.cuda.synchronize()
            dm = d.double() * mask[:, None].double()
            ref = torch.nn.grad.conv1d_weight(x.double(), (m, k, taps), dm, padding=(taps - 1) // 2).permute(2, 1, 0)
            err = float((dwp.double() - ref).abs().max() / ref.abs().max())
            eb = float((db.double() - dm.sum((0, 2))).abs().max() / dm.sum((0, 2)).abs().max())
            for _ in range(5):
                run(False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50
            e0.record()
            for _ in range(n):
                run(False)
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / n
            flops = 2.0 * m * k * taps * b * t
            print(f"{mode:11s} B={b:3d} K={k} M={m} T={t:4d} taps={taps}: {us:7.1f} us/launch  {flops / us / 1e6:7.1f} TFLOP/s fp32-equivalent  "
                  f"err dW {err:.2e} dbias {eb:.2e}", flush=True)


if __name__ == "__main__":
    main()
