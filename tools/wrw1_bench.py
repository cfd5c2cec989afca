"""Time of glowtts_conv_wrw1_multi on a flow block's six 1x1 weight gradients (B=32, H=192, C=160, T'=400) and on a transformer
layer's four (B=32, 192 channels, T=160): HIP events around back-to-back launches.  python tools/wrw1_bench.py [reps]
GLOWTTS_WRW1_CUS=<n> sizes the split-K for n compute units; GLOWTTS_WRW1_EXP (tuning build only:
GLOWTTS_HIP_LIB=tools/libglowtts_trace.bin) bit 0 drops the atomics, bit 2 the MFMAs, bit 3 the loads after the first step (timing
experiments: the results are then wrong)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
import torch  # noqa: E402

from glow_tts_train import _hip, convops  # noqa: E402


def problems(specs, b, t, dev="cuda"):
    probs = (_hip.Wrw1Problem * len(specs))()
    keep = []
    mask = torch.ones(b, t, device=dev)
    for j, (cin, m, split, md, xw) in enumerate(specs):
        x = torch.randn(b, xw or cin, t, device=dev)
        d = torch.randn(b, split if split else m, t, device=dev)
        d2 = torch.randn(b, m - split, t, device=dev) if split else None
        dwp, dbias = torch.zeros(cin, m, device=dev), torch.zeros(m, device=dev)
        q = probs[j]
        q.x, q.d, q.d2 = x.data_ptr(), d.data_ptr(), (d2.data_ptr() if split else None)
        q.mask_d, q.mask_x = (mask.data_ptr() if md else None), None
        q.dwp, q.dbias = dwp.data_ptr(), dbias.data_ptr()
        q.x_bs, q.d_bs, q.d2_bs = x.shape[1] * t, d.shape[1] * t, ((m - split) * t if split else 0)
        q.Cin, q.M, q.d_split = cin, m, split
        keep.append((x, d, d2, dwp, dbias, mask))
    return probs, keep


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    convops.set_conv_math("bf16x6+wrw")
    for name, specs, b, t in (("flow block (6 problems, 12 800 frames)",
                               [(192, 384, 192, False, 0)] * 3 + [(192, 192, 0, False, 0), (192, 160, 0, False, 0), (80, 192, 0, True, 160)], 32, 400),
                              ("transformer layer (4 problems, 5 120 frames)", [(192, 192, 0, False, 0)] * 4, 32, 160)):
        probs, keep = problems(specs, b, t)
        for _ in range(5):
            _hip.call("glowtts_conv_wrw1_multi", len(specs), ctypes.addressof(probs), b, t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            _hip.call("glowtts_conv_wrw1_multi", len(specs), ctypes.addressof(probs), b, t)
        e1.record()
        torch.cuda.synchronize()
        gf = sum(2.0 * c * m * b * t for c, m, *_ in specs) / 1e9
        us = 1e3 * e0.elapsed_time(e1) / reps
        print(f"{name}: {us:7.1f} us per launch  ({gf:.2f} GFLOP -> {6 * gf / us / 1e3:.0f} TFLOP/s on the bf16 pipe)", flush=True)


if __name__ == "__main__":
    main()
