#!/usr/bin/env python3
"""DESIGN.md lesson 12 probe: glowtts_actnorm_invconv_bwd run beside different co-resident kernels.

  python tools/lesson12_probe.py [iterations=150]
  GLOWTTS_HIP_LIB=glow-tts-train_amd/lib/exp/lib_slp.so python tools/lesson12_probe.py     # flows.hip built WITH the SLP
                                                        # vectorizer (tools/exp_build.sh flows.hip "slp:-fslp-vectorize")

Aggressors on a second stream: our bf16-plane weight gradient (the original trigger), our native-fp32 weight gradient, a torch
bf16 GEMM and a torch fp32 GEMM (hipBLASLt / rocBLAS kernels: code that is not ours), nothing.  Reported: launches whose dW /
dlogs / dbias differ from the first launch by more than 5e-6 of the largest entry (atomics' ordering noise is ~1e-7).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "glow-tts-train_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from glow_tts_train import _hip  # noqa: E402
from glow_tts_train._hip import call, ptr  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    _hip.load()
    print("library:", _hip.library_path())
    H = 192
    side = torch.cuda.Stream()
    a16 = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
    a32 = torch.randn(1024, 1024, device="cuda")
    for (B, C, T) in ((8, 160, 64), (8, 160, 124)):
        torch.manual_seed(0)
        x, dz = torch.randn(B, C, T, device="cuda"), torch.randn(B, C, T, device="cuda")
        m = torch.ones(B, T, device="cuda")
        logs, bias = torch.randn(C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
        w = torch.linalg.qr(torch.randn(4, 4))[0].cuda().contiguous()
        winv = torch.linalg.inv(w).contiguous()
        xlen, dld = m.sum(1), torch.randn(B, device="cuda")
        dx = torch.empty_like(x)
        xw, d2 = torch.randn(B, H, 120, device="cuda"), torch.randn(B, 2 * H, 120, device="cuda")
        dwp5 = torch.zeros(5, H, 2 * H, device="cuda")

        def wrw():
            for _ in range(4):
                call("glowtts_conv_wrw", ptr(xw), xw.stride(0), ptr(d2), d2.stride(0), None, None, ptr(dwp5), None, B, H, 2 * H,
                     120, 5, 1, 2)

        aggressors = [
            ("bf16-plane weight gradient (ours)", "bf16x6+wrw", wrw),
            ("native fp32 weight gradient (ours)", "fp32", wrw),
            ("torch bf16 GEMM 2048^3 (library kernel)", "fp32", lambda: torch.matmul(a16, a16)),
            ("torch fp32 GEMM 1024^3 (library kernel)", "fp32", lambda: torch.matmul(a32, a32)),
            ("nothing", "fp32", lambda: None),
        ]
        for name, math, fn in aggressors:
            before = _hip.conv_math(math)
            try:
                ref, worst, bad = None, 0.0, 0
                for _ in range(iters):
                    dlogs, dbias, dw = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(16, device="cuda")
                    torch.cuda.synchronize()
                    with torch.cuda.stream(side):
                        fn()
                    call("glowtts_actnorm_invconv_bwd", ptr(x), ptr(m), ptr(logs), ptr(bias), ptr(w), ptr(winv), ptr(dz), ptr(dld),
                         ptr(xlen), ptr(dx), ptr(dlogs), ptr(dbias), ptr(dw), B, C, T, 4)
                    torch.cuda.synchronize()
                    got = torch.cat([dw, dlogs, dbias])
                    if ref is None:
                        ref = got.clone()
                    err = float((got - ref).abs().max() / ref.abs().max())
                    worst = max(worst, err)
                    bad += err > 5e-6
                print(f"T'={T:4d}  beside {name:42s}: {bad:4d} / {iters} launches off, worst {worst:.2e}", flush=True)
            finally:
                _hip.conv_math(before)


if __name__ == "__main__":
    main()
