"""Debug aid for csrc/wn_fused.hip: runs a WN stack's forward with the layer-resident kernel on and off and compares every tensor
it leaves behind (skip, x_{l+1}, acts, tanh / sigmoid), printing where they differ.  python tools/wnf_debug.py B T n_layers p_drop"""
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
    b, t, nl, p = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    h = 192
    convops.set_conv_math("bf16x6+wrw")
    torch.manual_seed(5)
    wn = layers.WN(2 * h, h, kernel_size=5, dilation_rate=1, n_layers=nl, p_dropout=p).cuda().train()
    with torch.no_grad():
        for q in wn.parameters():
            if q.dim() == 1:
                q.normal_(0, 0.05)
    lens = torch.randint(max(1, t // 2), t + 1, (b,))
    lens[0] = t
    m2 = (torch.arange(t)[None] < lens[:, None]).float().cuda()
    x = (torch.randn(b, h, t).cuda() * m2[:, None]).contiguous()
    keep = ops.keep_mask((nl, b, 2 * h, t), p, "cuda", "dbg") if p > 0 else None
    flat = []
    for a, r in zip(wn.in_layers, wn.res_skip_layers):
        flat.extend(wn._conv_params(a))
        flat.extend(wn._conv_params(r))
    plan = convops.WNPackPlan(want_planes=True)

    def run(fused):
        _hip.wn_fused(fused)
        ctx = Ctx()
        with torch.no_grad():
            skip = convops.WNFn.forward(ctx, x, m2, None, p, 1, nl, plan, keep, *flat)
        torch.cuda.synchronize()
        sv = ctx.saved
        out = {"skip": skip, "acts": sv[2], "ts": sv[3]}
        if nl > 1:
            out["xs"] = sv[4]
        return out

    n0 = _hip.wn_fused_launches()
    a = run(True)
    print("fused launches:", _hip.wn_fused_launches() - n0)
    r = run(False)
    for k in r:
        d = (a[k] - r[k]).abs()
        scale = float(r[k].abs().max())
        print(f"{k:5s} shape {tuple(r[k].shape)} max |diff| {float(d.max()):.3e} (ref max {scale:.3e}) nan {int(torch.isnan(a[k]).sum())}")
        if float(d.max()) > 1e-4 * max(1.0, scale):
            bad = (d > 1e-4 * max(1.0, scale)).nonzero()
            print("   first bad:", bad[:6].tolist(), " count", bad.shape[0])
            for dim in range(r[k].dim()):
                print(f"   bad index set along dim {dim}:", sorted(set(bad[:, dim].tolist()))[:40])


if __name__ == "__main__":
    main()
