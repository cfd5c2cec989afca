"""csrc/flow_boundary.hip alone: the one-launch block boundary against the three launches it replaces, forward and backward, at the
benchmark's shape (B = 32, C = 160, H = 192, T' = 400), HIP events around 200 back-to-back launches; then with one ingredient
dropped at a time (only with GLOWTTS_HIP_LIB=tools/libglowtts_trace.bin, the tuning build; GLOWTTS_BND_EXP bits: 1 = first contraction's MFMAs, 2 = second contraction's, 4 = the element-wise phase,
8 = backward: the group reduction of the parameter-gradient partials).  DESIGN.md lesson 36.

  python tools/boundary_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd")]
import torch  # noqa: E402

from glow_tts_train import _hip  # noqa: E402

_hip.load()
call = _hip.call
P = lambda x: x.data_ptr()                                                 # noqa: E731
dev = "cuda"
b, c, h, t, ns = 32, 160, 192, 400, 4
f = lambda *s: torch.randn(*s, device=dev)                                 # noqa: E731
mask, x_len = torch.ones(b, t, device=dev), torch.full((b,), float(t), device=dev)
logs, bias, w = f(c) * 0.1, f(c) * 0.1, torch.linalg.qr(f(ns, ns))[0].contiguous()
w_inv = torch.cat([torch.inverse(w).flatten(), torch.zeros(1, device=dev)]).contiguous()
logdet_w = w_inv[ns * ns:]
cur = torch.cuda.current_stream()


def timeit(name, fn, iters=200):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    for _ in range(iters):
        fn()
    e1.record(cur)
    torch.cuda.synchronize()
    print(f"{name:44s} {1e3 * e0.elapsed_time(e1) / iters:7.2f} us", flush=True)


# ---- forward -----------------------------------------------------------------------------------------------------------------
skip, y_prev = f(b, h, t), f(b, c, t)
wp_end, b_end = f(h // 16, c, 16) * 0.05, f(c) * 0.1
wp_start, b_start = f((c // 2 + 15) // 16, h, 16) * 0.05, f(h) * 0.1
out, y, h0 = torch.empty(b, c, t, device=dev), torch.empty(b, c, t, device=dev), torch.empty(b, h, t, device=dev)
out2, y2, h02 = torch.empty_like(out), torch.empty_like(y), torch.empty_like(h0)
ldp, ld, ldp2, ld2 = (torch.zeros(b, device=dev) for _ in range(4))


def fwd():
    call("glowtts_flow_boundary_fwd", P(skip), P(wp_end), P(b_end), P(y_prev), P(mask), P(logs), P(bias), P(w), P(logdet_w), P(x_len),
         P(wp_start), P(b_start), P(out), P(y), P(h0), P(ldp), P(ld), b, c, h, t, ns, 0)


def fwd3():
    call("glowtts_conv_fwd", P(skip), h * t, P(wp_end), P(b_end), None, None, 0, P(out2), c * t, b, h, c, t, 1, 1, 0, 0, 0, 0)
    call("glowtts_coupling_actnorm_invconv_fwd", P(y_prev), P(out2), P(mask), P(logs), P(bias), P(w), P(logdet_w), P(x_len), P(y2),
         P(ldp2), P(ld2), b, c, t, ns, 0)
    call("glowtts_conv_fwd", P(y2), c * t, P(wp_start), P(b_start), P(mask), None, 0, P(h02), h * t, b, c // 2, h, t, 1, 1, 0, 0, 1, 0)


fwd(); fwd3(); torch.cuda.synchronize()
print("forward: max |difference| out / y / h0 against the three launches:", float((out - out2).abs().max()), float((y - y2).abs().max()),
      float((h0 - h02).abs().max()))
timeit("forward boundary, one launch", fwd)
timeit("forward boundary, three launches", fwd3)
EXP = "libglowtts_trace" in _hip.library_path()            # the skip-work experiments exist in the tuning build only (make trace)
for e in (1, 2, 4, 7) if EXP else ():
    _hip.set_knob("GLOWTTS_BND_EXP", e)
    timeit(f"  forward, GLOWTTS_BND_EXP={e}", fwd)
if EXP:
    _hip.set_knob("GLOWTTS_BND_EXP", 0)

# ---- backward ----------------------------------------------------------------------------------------------------------------
dx_wn, dy_next, out_prev, dlogdet = f(b, h, t), f(b, c, t), f(b, c, t) * 0.3, f(b)
wb_start, wb_end = f(h // 16, c // 2, 16) * 0.05, f((c + 15) // 16, h, 16) * 0.05
got = [torch.empty(b, c, t, device=dev), torch.empty(b, c, t, device=dev), torch.empty(b, h, t, device=dev)]
want = [torch.empty_like(x) for x in got]
gg = [torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.zeros(ns, ns, device=dev)]
part = torch.empty(b * ((t + 31) // 32) * (c // ns) * (2 * ns + ns * ns), device=dev)
dyf = dy_next.clone()


def bwd():
    call("glowtts_flow_boundary_bwd", P(dx_wn), P(wb_start), P(dy_next), P(y_prev), P(out_prev), P(mask), P(logs), P(bias), P(w),
         P(dlogdet), P(wb_end), P(got[0]), P(got[1]), P(got[2]), P(part), b, c, h, t, ns, 0, 1)


def red():
    call("glowtts_flow_boundary_bwd_reduce", P(part), P(w_inv), P(dlogdet), P(x_len), P(gg[0]), P(gg[1]), P(gg[2]), b, c, t, ns)


def bwd3():
    dyf.copy_(dy_next)
    call("glowtts_conv_fwd", P(dx_wn), h * t, P(wb_start), None, P(mask), P(dyf), c * t, P(dyf), c * t, b, h, c // 2, t, 1, 1, 0, 1, 0, 0)
    call("glowtts_coupling_actnorm_invconv_bwd", P(y_prev), P(out_prev), P(mask), P(logs), P(bias), P(w), P(w_inv), P(dyf), P(dlogdet),
         P(x_len), P(want[0]), P(want[1]), P(gg[0]), P(gg[1]), P(gg[2]), b, c, t, ns, 0)
    call("glowtts_conv_fwd", P(want[1]), c * t, P(wb_end), None, P(mask), None, 0, P(want[2]), h * t, b, c, h, t, 1, 1, 0, 0, 1, 0)


bwd(); bwd3(); torch.cuda.synchronize()
print("backward: max |difference| dy / dout / dskip against the three launches:", *[float((a - e).abs().max()) for a, e in zip(got, want)])
timeit("backward boundary, one launch", bwd)
timeit("backward: reduction of the partials", red)
timeit("backward boundary, three launches (+ one copy)", bwd3)
for e in (1, 2, 4, 8, 15) if EXP else ():
    _hip.set_knob("GLOWTTS_BND_EXP", e)
    timeit(f"  backward, GLOWTTS_BND_EXP={e}", bwd)
