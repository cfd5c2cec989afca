#!/usr/bin/env python3
"""Weight packing / un-packing of one WN stack (config 2: 4 layers, 192 hidden channels, 5 taps) and of one encoder layer,
back-to-back launches on one stream: microseconds per launch and the HBM-side rate of the bytes a launch has to move."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "glow-tts-train_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from glow_tts_train import _hip, convops  # noqa: E402


def timed(fn, n=100):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def main():
    _hip.load()
    dev = "cuda"
    H = 192
    cases = {
        "WN stack (4 x [384<-192 x5, 384<-192 x1])": [(2 * H, H, 5, True), (2 * H, H, 1, True)] * 3 + [(2 * H, H, 5, True), (H, H, 1, True)],
        "encoder layer (4 x 192<-192 x1, 768<-192 x3, 192<-768 x3)": [(H, H, 1, False)] * 4 + [(768, H, 3, False), (H, 768, 3, False)],
        "one 1x1 conv 192<-80": [(H, 80, 1, False)],
    }
    for mode in ("bf16x6+wrw", "fp32"):
        convops.set_conv_math(mode)
        for name, shapes in cases.items():
            params = []
            for co, ci, k, wn in shapes:
                v = torch.randn(co, ci, k, device=dev) * 0.1
                g = torch.rand(co, 1, 1, device=dev) + 0.5 if wn else None
                b = torch.zeros(co, device=dev)
                for t in (v, g, b):
                    if t is not None:
                        t.grad = torch.zeros_like(t)
                params += [v, g, b]
            plan = convops.WNPackPlan(want_planes=True)
            plan.ensure(params, n_convs=len(shapes))
            n_w = sum(co * ci * k for co, ci, k, _ in shapes)
            t_pack = timed(plan.pack)
            planes = mode != "fp32"
            pack_bytes = 4 * n_w + plan.wp_arena.numel() * (4 + (6 if planes else 0))
            plan.dwp.normal_()
            t_un = timed(lambda: plan.unpack_into_grads(params))
            un_bytes = 4 * n_w * 4
            print(f"{mode:11s} {name}: pack {t_pack:6.1f} us ({pack_bytes / 1e6:5.1f} MB, {pack_bytes / t_pack / 1e6:5.2f} TB/s)   "
                  f"un-pack {t_un:6.1f} us ({un_bytes / 1e6:5.1f} MB, {un_bytes / t_un / 1e6:5.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()
