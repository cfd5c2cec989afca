#!/usr/bin/env python3
"""Four weight gradients of one shape: one launch per problem against ONE batched launch (glowtts_conv_wrw_batch), results compared.
B=32, T'=400, 192 -> 384 channels, 5 taps and 1 tap, single- and two-source forms."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "glow-tts-train_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from glow_tts_train import _hip, convops  # noqa: E402
from glow_tts_train._hip import call, ptr  # noqa: E402


_keep = []


def parr(ts):
    a = (ctypes.c_void_p * len(ts))(*[0 if t is None else t.data_ptr() for t in ts])
    _keep.append(a)
    return ctypes.addressof(a)


def main():
    _hip.load()
    convops.set_conv_math("bf16x6+wrw")
    b, k, m, t, n = 32, 192, 384, 400, 4
    for taps in (5, 1):
        for two in (False, True):
            torch.manual_seed(taps)
            warm = os.environ.get("WRW_BATCH_WARM") == "1"          # the same operands for every problem: they stay in L2 / MALL
            xs = [torch.randn(b, k, t, device="cuda") for _ in range(n)]
            ds = [torch.randn(b, m // 2 if two else m, t, device="cuda") for _ in range(n)]
            d2 = [torch.randn(b, m // 2, t, device="cuda") for _ in range(n)] if two else None
            if warm:
                xs, ds = [xs[0]] * n, [ds[0]] * n
                d2 = [d2[0]] * n if two else None
            ref = [torch.zeros(taps, k, m, device="cuda") for _ in range(n)]
            out = [torch.zeros(taps, k, m, device="cuda") for _ in range(n)]
            rb = [torch.zeros(m, device="cuda") for _ in range(n)]
            ob = [torch.zeros(m, device="cuda") for _ in range(n)]
            pad = (taps - 1) // 2

            def singles(dst, dbs):
                for q in range(n):
                    if two:
                        call("glowtts_conv_wrw2", ptr(xs[q]), xs[q].stride(0), ptr(ds[q]), ds[q].stride(0), ptr(d2[q]), d2[q].stride(0),
                             m // 2, ptr(dst[q]), ptr(dbs[q]), b, k, m, t, taps, 1, pad)
                    else:
                        call("glowtts_conv_wrw", ptr(xs[q]), xs[q].stride(0), ptr(ds[q]), ds[q].stride(0), None, None, ptr(dst[q]),
                             ptr(dbs[q]), b, k, m, t, taps, 1, pad)

            ax, ad, ad2 = parr(xs), parr(ds), (parr(d2) if two else None)

            def batch(dst, dbs):
                call("glowtts_conv_wrw_batch", n, ax, xs[0].stride(0), ad, ds[0].stride(0), ad2, d2[0].stride(0) if two else 0,
                     m // 2 if two else 0, None, None, parr(dst), parr(dbs), b, k, m, t, taps, 1, pad)

            singles(ref, rb)
            batch(out, ob)
            torch.cuda.synchronize()
            err = max(float((o - r).abs().max() / r.abs().max()) for o, r in zip(out, ref))
            eb = max(float((o - r).abs().max() / r.abs().max()) for o, r in zip(ob, rb))

            def timed(fn):
                for _ in range(3):
                    fn(out, ob)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn(out, ob)
                e1.record()
                torch.cuda.synchronize()
                return 1e3 * e0.elapsed_time(e1) / 20

            print(f"taps={taps} two_source={two}: 4 launches {timed(singles):7.1f} us, one batched launch {timed(batch):7.1f} us, "
                  f"max rel difference dW {err:.1e} dbias {eb:.1e}", flush=True)


if __name__ == "__main__":
    main()
