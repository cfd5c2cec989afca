"""Is the attention backward the same from launch to launch?  (DESIGN.md lesson 33: it was not.)

  python tools/attn_repeat_probe.py [T] [d_k] [launches]

dS, dQ, dK, dV have no atomics in them: `launches` backward launches on fixed inputs must agree bit for bit, in the fp32 and in the
bf16 MFMA form.  For every launch that differs from the first the probe prints which (utterance, head, 16-row tile) of dS differs and,
per differing row, the columns whose error is NOT explained by a wrong row sum (diff / p constant along a row = only `sum p dp` is
off; the odd columns are where dp itself is wrong).  Before csrc/common.hpp's mfma_settle(): ~25 % of launches at T = 128 / 160 / 192,
rows 3, 7, 11, 15 of the last two row tiles, columns of the last 16-key tile — the accumulator register the last MFMA pass writes,
read one wait state after the MFMA behind a taken branch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd")]
import torch  # noqa: E402

from glow_tts_train import _hip  # noqa: E402
from glow_tts_train._hip import call  # noqa: E402


def main():
    t = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    dk = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    _hip.load()
    ptr = lambda x: None if x is None else x.data_ptr()                    # noqa: E731
    torch.manual_seed(1)
    b, h, w = 3, 2, 4
    dev = "cuda"
    q, k, v, dout = (torch.randn(b, h * dk, t, device=dev) for _ in range(4))
    lens = torch.tensor([t, (3 * t) // 4, t // 2 - 3], device=dev)
    m2 = (torch.arange(t, device=dev)[None] < lens[:, None]).float().contiguous()
    ek, ev = (torch.randn(1, 2 * w + 1, dk, device=dev) * dk ** -0.5 for _ in range(2))

    def backward(bf, p):
        ds = torch.full((b, h, t, t), float("nan"), device=dev)
        dq, dkk, dv = (torch.full_like(q, float("nan")) for _ in range(3))
        dek, dev_ = torch.zeros_like(ek), torch.zeros_like(ev)
        call("glowtts_rel_attn_bwd_ex", ptr(dout), ptr(q), ptr(k), ptr(v), ptr(ek), ptr(ev), ptr(m2), None, 1.0, ptr(p), ptr(ds), ptr(dq),
             ptr(dkk), ptr(dv), ptr(dek), ptr(dev_), b, h, t, dk, w, 1, -1, int(bf))
        torch.cuda.synchronize()
        return {"ds": ds, "dq": dq, "dk": dkk, "dv": dv}

    for bf in (False, True):
        p = torch.empty(b, h, t, t, device=dev)
        out = torch.empty_like(q)
        call("glowtts_rel_attn_fwd_ex", ptr(q), ptr(k), ptr(v), ptr(ek), ptr(ev), ptr(m2), None, 1.0, ptr(p), ptr(out), b, h, t, dk, w, 1,
             -1, int(bf))
        first = backward(bf, p)
        bad = 0
        for it in range(n):
            r = backward(bf, p)
            d = {name: float((x - first[name]).abs().max()) for name, x in r.items()}
            if not any(x > 0 or x != x for x in d.values()):
                continue
            bad += 1
            if bad > 3:
                continue
            e = r["ds"] - first["ds"]
            tiles = sorted({(int(a), int(c), int(rr) // 16) for a, c, rr in torch.nonzero(e.abs().amax(dim=3) > 0)})
            print(f"{'bf16' if bf else 'fp32'} launch {it}: max |diff| {d}; (utterance, head, row tile) of dS: {tiles[:12]}")
            for a, c, rr in torch.nonzero(e.abs().amax(dim=3) > 0)[:4].tolist():
                ratio = e[a, c, rr] / p[a, c, rr].clamp_min(1e-30)
                med = ratio.median()
                odd = torch.nonzero((ratio - med).abs() > 1e-3 * med.abs() + 1e-6).flatten().tolist()
                print(f"    row {rr}: diff / p = {float(med):+.4f} along the row; columns where dp itself is off: "
                      f"{odd[:1]} .. {odd[-1:]} ({len(odd)})")
        print(f"{'bf16' if bf else 'fp32'} MFMA form, T = {t}, d_k = {dk}: {bad} of {n} launches differ from the first")


if __name__ == "__main__":
    main()
