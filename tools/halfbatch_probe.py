"""Does a dependent chain of conv launches run faster as TWO half-batch chains on two streams (one workgroup of each per CU, out of
phase: one chain's prologue / epilogue under the other's MFMA loop) than as one full-batch chain whose two co-resident workgroups
per CU move in lock-step?  WN stack forward (gated 5-tap conv + 1x1 res/skip conv per layer) at the benchmark's shape.

  python tools/halfbatch_probe.py [B T n_layers reps]

Prints microseconds per stack forward of the FULL batch for: one chain of B; two chains of B/2 on two streams; two chains of B/2
one after the other on one stream (what the smaller grids cost by themselves); host enqueue time of each form."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
import torch  # noqa: E402

from glow_tts_train import _hip, convops, layers, ops  # noqa: E402


class Ctx:
    def save_for_backward(self, *a):
        self.saved = a


def main():
    b, t, nl, reps = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (32, 400, 4, 30)
    h, p = 192, 0.05
    convops.set_conv_math("bf16x6+wrw")
    _hip.wn_fused(False)
    torch.manual_seed(5)
    wn = layers.WN(2 * h, h, kernel_size=5, dilation_rate=1, n_layers=nl, p_dropout=p).cuda().train()
    flat = []
    for a, r in zip(wn.in_layers, wn.res_skip_layers):
        flat.extend(wn._conv_params(a))
        flat.extend(wn._conv_params(r))
    plan = convops.WNPackPlan(want_planes=True)
    plan.ensure(flat, nl)
    plan.pack()
    bound = plan.bind()
    hb = b // 2

    def inputs(n):
        return (torch.randn(n, h, t, device="cuda"), torch.ones(n, t, device="cuda"), ops.keep_mask((nl, n, 2 * h, t), p, "cuda", "probe"))

    full, ha, hbb = inputs(b), inputs(hb), inputs(hb)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()

    def fwd(inp):
        x, m2, keep = inp
        return convops.WNFn._forward(Ctx(), x, m2, None, p, 1, nl, plan, keep, *flat)

    def one_chain():
        with torch.cuda.stream(s0):
            fwd(full)

    def two_chains():
        with torch.cuda.stream(s0):
            fwd(ha)
        with torch.cuda.stream(s1):
            fwd(hbb)

    def halves_serial():
        with torch.cuda.stream(s0):
            fwd(ha)
            fwd(hbb)

    def timed(fn):
        with torch.no_grad():
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        return 1e6 * (t2 - t0) / reps, 1e6 * (t1 - t0) / reps

    for rnd in range(3):
        for name, fn in (("one chain, B=%d" % b, one_chain), ("two chains of B=%d on two streams" % hb, two_chains),
                         ("two chains of B=%d, one stream" % hb, halves_serial)):
            gpu, host = timed(fn)
            print(f"round {rnd}: {name:42s} {gpu:8.1f} us per full-batch stack forward   (host enqueue {host:6.1f} us)", flush=True)
    plan.unbind(bound)


if __name__ == "__main__":
    main()
