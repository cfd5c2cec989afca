#!/usr/bin/env python3
"""Phase timeline of one conv-kernel launch (tuning tool; needs the trace build: `make -C glow-tts-train_amd/csrc trace`).
Usage: GLOWTTS_HIP_LIB=tools/libglowtts_trace.bin python tools/trace_conv.py [gate|resskip|bwd_data5|bwd_data1|start]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
os.environ.setdefault("GLOWTTS_HIP_LIB", os.path.join(ROOT, "tools", "libglowtts_trace.bin"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from glow_tts_train import _hip, convops  # noqa: E402
from glow_tts_train._hip import call, ptr  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "gate"
B, H, T, C = 32, 192, 400, 160
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(B, H, T, device=dev)
m2 = torch.ones(B, T, device=dev)
v_in = torch.randn(2 * H, H, 5, device=dev) * 0.03
g_in = torch.ones(2 * H, 1, 1, device=dev)
b_in = torch.zeros(2 * H, device=dev)
v_rs = torch.randn(2 * H, H, 1, device=dev) * 0.07
wf_in, wb_in, _ = convops.pack_weight(v_in, g_in)
wf_rs, wb_rs, _ = convops.pack_weight(v_rs, None)
acts = torch.empty(B, H, T, device=dev)
ts = torch.empty(B, 2 * H, T, device=dev)
xo = torch.empty(B, H, T, device=dev)
sk = torch.empty(B, H, T, device=dev)
d2 = torch.randn(B, 2 * H, T, device=dev)
dx = torch.empty(B, H, T, device=dev)

fns = {
    "gate": lambda: call("glowtts_conv_gate_fwd", ptr(x), ptr(wf_in), ptr(b_in), None, None, 1.0, ptr(acts), ptr(ts), B, H, T, 5, 1, 2),
    "resskip": lambda: call("glowtts_conv_res_skip_fwd", ptr(acts), ptr(wf_rs), ptr(b_in), ptr(m2), ptr(x), ptr(sk), ptr(xo), ptr(sk), B, H, T, 0),
    "bwd_data5": lambda: convops.conv_fwd(d2, wb_in, None, None, dx, 2 * H, H, 5, 1, 2, addend=d2[:, :H]),
    "bwd_data1": lambda: convops.conv_fwd(d2, wb_rs, None, None, dx, 2 * H, H, 1, 1, 0),
    "gate_bwd": lambda: call("glowtts_conv_gate_bwd", ptr(d2), None, ptr(wb_rs), ptr(ts), None, 1.0, ptr(torch.empty_like(d2)), B, 2 * H, H, T),
    "wrw5": lambda: call("glowtts_conv_wrw", ptr(x), x.stride(0), ptr(d2), d2.stride(0), None, None, ptr(dwp5), None, B, H, 2 * H, T, 5, 1, 2),
    "wrw1": lambda: call("glowtts_conv_wrw", ptr(x), x.stride(0), ptr(d2), d2.stride(0), None, None, ptr(dwp1), None, B, H, 2 * H, T, 1, 1, 0),
}
dwp5 = torch.zeros(5, H, 2 * H, device=dev)
dwp1 = torch.zeros(1, H, 2 * H, device=dev)
if which == "wrw5p":                                   # weight gradient from operands split once (3 bf16 planes)
    xp = torch.empty(3 * x.numel(), device=dev, dtype=torch.int16)
    dp = torch.empty(3 * d2.numel(), device=dev, dtype=torch.int16)
    call("glowtts_split_planes", ptr(x), x.numel(), ptr(xp), 3)
    call("glowtts_split_planes", ptr(d2), d2.numel(), ptr(dp), 3)
    fns["wrw5p"] = lambda: call("glowtts_conv_wrw_planes", ptr(xp), x.numel(), H * T, ptr(dp), d2.numel(), 2 * H * T, ptr(dwp5),
                                None, B, H, 2 * H, T, 5, 3)
    os.environ.setdefault("TRACE_CONV_MATH", "fp32")
if which == "wino":                                    # the gated in-conv in its Winograd form (csrc/convwino.hip)
    fns["wino"] = fns["gate"]
    os.environ.setdefault("TRACE_CONV_MATH", "bf16x6+wrw")
fn = fns[which]
lib = _hip.load()
split = os.environ.get("TRACE_CONV_MATH")          # e.g. "bf16x6+wrw": trace the bf16-plane kernels (convgemm_split.hip)
if split:
    _hip.conv_math(split)
    _planes = {}
    for w in (wf_in, wb_in, wf_rs, wb_rs):          # bind the one buffer the chosen kernel uses
        _planes[w.data_ptr()] = torch.empty(3 * w.numel(), device=dev, dtype=torch.int16)
    _use = {"gate": wf_in, "wino": wf_in, "resskip": wf_rs, "bwd_data5": wb_in, "bwd_data1": wb_rs, "gate_bwd": wb_rs}.get(which)
    if _use is not None:
        call("glowtts_conv_split_weights", ptr(_use), _use.numel(), ptr(_planes[_use.data_ptr()]))
        _hip.conv_bind_planes(_use, _planes[_use.data_ptr()])
rd = lib.glowtts_debug_trace_read_split if (split or which == "wrw5p") else lib.glowtts_debug_trace_read
if which == "wino":
    n_u = _hip.wino_plane_elems(wf_in.numel())
    u_planes = torch.zeros(3 * n_u, device=dev, dtype=torch.int16)
    table = torch.tensor([[0, H // 16, 2 * H]], dtype=torch.int64, device=dev)
    call("glowtts_wino_weights", ptr(wf_in), wf_in.numel(), ptr(table), 1, ptr(u_planes), n_u)
    _hip.conv_bind_wino(wf_in, u_planes)
    _hip.set_knob("WINO", 1)
    rd = lib.glowtts_debug_trace_read_wino
if which == "wrw5" and split and "wrw" in split and os.environ.get("GLOWTTS_WRW_TR", "1") != "0":
    rd = lib.glowtts_debug_trace_read_tr            # the frame-major / transposed-read kernel (convwrw_tr.hip)
rd.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
NW = 8192 * 16
buf = np.zeros(NW, dtype=np.uint64)
for _ in range(3):
    fn()
torch.cuda.synchronize()
rd(buf.ctypes.data, NW, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
fn()
e1.record()
torch.cuda.synchronize()
rd(buf.ctypes.data, NW, 0)
tr = buf.reshape(8192, 16).astype(np.int64)
wg_index = np.nonzero(tr[:, 0] > 0)[0]
tr = tr[tr[:, 0] > 0]
t00 = tr[:, 0].min()
us = lambda a: (a - t00) / 100.0      # 100 MHz wall clock -> microseconds
print(f"{which}: {len(tr)} workgroups, event time {e0.elapsed_time(e1) * 1e3:.1f} us, "
      f"first start -> last end {us(tr[:, 10].max()):.1f} us")
cols = [c for c in ((0, 1, 2, 3, 4, 10) if rd is getattr(lib, "glowtts_debug_trace_read_tr", None) else range(11)) if (tr[:, c] > 0).all()]
print("trace point : " + "  ".join(f"{c:>7d}" for c in cols))
for name, f in (("min", np.min), ("median", np.median), ("max", np.max)):
    print(f"abs {name:7s}: " + "  ".join(f"{f(us(tr[:, c])):7.1f}" for c in cols))
prev = None
print("phase lengths (per workgroup, us):")
for c in cols:
    if prev is not None:
        d = (tr[:, c] - tr[:, prev]) / 100.0
        print(f"  {prev:2d} -> {c:2d}: min {d.min():6.1f}  median {np.median(d):6.1f}  max {d.max():6.1f}")
    prev = c
if rd is getattr(lib, "glowtts_debug_trace_read_tr", None):
    # convwrw_tr.hip: shader cycles of the main loop by role, summed over a workgroup's half periods (wave 0 of each group)
    for g, cols_ in (("group 0", (5, 6, 7, 8)), ("group 1", (9, 11, 12, 13))):
        v = [np.median(tr[:, c]) for c in cols_]
        print(f"main loop, {g}: multiply {v[0]:.0f} + barrier wait {v[1]:.0f}; store {v[2]:.0f} + barrier wait {v[3]:.0f}  (cycles, median)")
elif (tr[:, 11] > 0).all():
    cyc = (tr[:, 12] - tr[:, 11]).astype(float)
    wall = (tr[:, 4] - tr[:, 3]) / 100.0
    print(f"shader clock over the 3 -> 4 phase: median {np.median(cyc / wall):.0f} cycles/us")
hw = tr[:, 15]
# HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13(+), ... ; XCC_ID separate register on gfx94x
cu_key = (hw >> 8) & 0xFFFF
print("distinct (cu, sh, se, ...) keys:", len(np.unique(cu_key)))
order = np.argsort(cu_key, kind="stable")
print("sample of co-resident workgroups (key, start, p1, p2, p3, p4, end):")
for i in order[:24]:
    print(f"  {cu_key[i]:6x} " + " ".join(f"{us(tr[i, c]):6.1f}" for c in (0, 1, 2, 3, 4, 10)))
start = us(tr[:, 0])
print("start-time histogram (us):", np.histogram(start, bins=8)[0].tolist(), [round(v, 1) for v in np.histogram(start, bins=8)[1].tolist()])

if (tr[:, 14] & 0x100).all() and rd is not getattr(lib, "glowtts_debug_trace_read_tr", None):
    # which CU a workgroup ran on (XCC, SE, SH, CU), how many workgroups shared it, and whether the slow ones have anything in common
    xcc = tr[:, 14] & 0xF
    cu = (xcc << 16) | ((hw >> 8) & 0xFFFF)
    keys, inv, cnt = np.unique(cu, return_inverse=True, return_counts=True)
    print("compute units used:", len(keys), " workgroups per CU:", dict(zip(*np.unique(cnt, return_counts=True))))
    life = us(tr[:, 10]) - us(tr[:, 0])
    for n in np.unique(cnt):
        sel = cnt[inv] == n
        print(f"  CUs with {n} workgroup(s): lifetime median {np.median(life[sel]):.1f} max {life[sel].max():.1f}; end median "
              f"{np.median(us(tr[sel, 10])):.1f} max {us(tr[sel, 10]).max():.1f}")
    print("per XCC: end-time median / max:", [(int(x), round(float(np.median(us(tr[xcc == x, 10]))), 1), round(float(us(tr[xcc == x, 10]).max()), 1))
                                               for x in np.unique(xcc)])
    slow = np.argsort(-life)[:16]
    print("slowest workgroups (index, xcc, cu key, co-resident, start, p1, p2, p3, p4, end):")
    for i in slow:
        print(f"  {wg_index[i]:5d} {xcc[i]:2d} {(hw[i] >> 8) & 0xFFFF:6x} {cnt[inv[i]]:2d} " + " ".join(f"{us(tr[i, c]):6.1f}" for c in (0, 1, 2, 3, 4, 10)))
    print("lifetime histogram:", np.histogram(life, bins=10)[0].tolist(), [round(v, 1) for v in np.histogram(life, bins=10)[1].tolist()])
