"""Time of the WN stack's forward at the benchmark's shape (B=32, H=192, T'=400, 4 layers, dropout on), layer-resident kernel vs
the per-layer launch sequence: HIP events around back-to-back calls of glowtts_wn_fwd.  python tools/wnf_bench.py [B T n_layers]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
import torch  # noqa: E402

from glow_tts_train import _hip, convops, layers, ops  # noqa: E402


class Ctx:
    def save_for_backward(self, *a):
        self.saved = a


def main():
    b, t, nl = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 400, 4)
    h, p = 192, 0.05
    convops.set_conv_math("bf16x6+wrw")
    torch.manual_seed(5)
    wn = layers.WN(2 * h, h, kernel_size=5, dilation_rate=1, n_layers=nl, p_dropout=p).cuda().train()
    m2 = torch.ones(b, t, device="cuda")
    x = torch.randn(b, h, t, device="cuda")
    keep = ops.keep_mask((nl, b, 2 * h, t), p, "cuda", "bench")
    flat = []
    for a, r in zip(wn.in_layers, wn.res_skip_layers):
        flat.extend(wn._conv_params(a))
        flat.extend(wn._conv_params(r))
    plan = convops.WNPackPlan(want_planes=True)
    for fused in (True, False, True, False):
        _hip.wn_fused(fused)
        with torch.no_grad():
            for _ in range(3):
                convops.WNFn.forward(Ctx(), x, m2, None, p, 1, nl, plan, keep, *flat)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n):
                convops.WNFn.forward(Ctx(), x, m2, None, p, 1, nl, plan, keep, *flat)
            e1.record()
            torch.cuda.synchronize()
        print(f"fused={int(fused)}: {1e3 * e0.elapsed_time(e1) / n:8.1f} us per WN-stack forward (incl. one weight pack + plane split launch)")


if __name__ == "__main__":
    main()
