#!/usr/bin/env python3
"""Where does a direct forward-type convolution kernel (csrc/convgemm_split.hip) spend what is not MFMA time?  The tuning build's
GLOWTTS_CONV_EXP bits make the kernel SKIP pieces (results wrong, timing valid): 1 = no activation loads, 2 = no split / LDS stores,
4 = no weight loads after the first three, 8 = no epilogue.  One kernel at the benchmark's shape, alone, back to back.

  make -C glow-tts-train_amd/csrc trace && GLOWTTS_HIP_LIB=tools/libglowtts_trace.bin python tools/conv_exp.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd")]
os.environ.setdefault("GLOWTTS_HIP_LIB", os.path.join(ROOT, "tools", "libglowtts_trace.bin"))
import torch  # noqa: E402

from glow_tts_train import _hip, convops  # noqa: E402
from glow_tts_train._hip import call, ptr  # noqa: E402

B, H, T = 32, 192, 400
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(B, H, T, device=dev)
m2 = torch.ones(B, T, device=dev)
d2 = torch.randn(B, 2 * H, T, device=dev)
v_in = torch.randn(2 * H, H, 5, device=dev) * 0.03
v_rs = torch.randn(2 * H, H, 1, device=dev) * 0.07
b_in = torch.zeros(2 * H, device=dev)
wf_in, wb_in, _ = convops.pack_weight(v_in, None)
wf_rs, wb_rs, _ = convops.pack_weight(v_rs, None)
acts, ts = torch.empty(B, H, T, device=dev), torch.randn(B, 2 * H, T, device=dev)
xo, sk, dx = torch.empty(B, H, T, device=dev), torch.zeros(B, H, T, device=dev), torch.empty(B, H, T, device=dev)
dpre = torch.empty(B, 2 * H, T, device=dev)
convops.set_conv_math("bf16x6+wrw")
_hip.set_knob("WINO", 0)
planes = {}
for w in (wf_in, wb_in, wf_rs, wb_rs):
    planes[w.data_ptr()] = torch.empty(3 * w.numel(), device=dev, dtype=torch.int16)
    call("glowtts_conv_split_weights", ptr(w), w.numel(), ptr(planes[w.data_ptr()]))
kernels = {
    "gated in-conv <3,2,5,1,5>": (wf_in, lambda: call("glowtts_conv_gate_fwd", ptr(x), ptr(wf_in), ptr(b_in), None, None, 1.0, ptr(acts), ptr(ts), B, H, T, 5, 1, 2)),
    "5-tap backward-data <3,1,5,4,5>": (wb_in, lambda: convops.conv_fwd(d2, wb_in, None, None, dx, 2 * H, H, 5, 1, 2, addend=x)),
    "res/skip 1x1 <3,2,5,2,1>": (wf_rs, lambda: call("glowtts_conv_res_skip_fwd", ptr(acts), ptr(wf_rs), ptr(b_in), ptr(m2), ptr(x), ptr(sk), ptr(xo), ptr(sk), B, H, T, 0)),
    "gate backward 1x1 <3,1,5,5,1>": (wb_rs, lambda: call("glowtts_conv_gate_bwd", ptr(d2), None, ptr(wb_rs), ptr(ts), None, 1.0, ptr(dpre), B, 2 * H, H, T)),
}
variants = [(0, "everything"), (1, "no activation loads"), (3, "no activation loads, no split / stores"), (4, "no weight loads"),
            (8, "no epilogue"), (11, "no staging, no epilogue"), (15, "MFMAs (+ LDS reads) only"),
            (16, "next chunk's loads issued mid-chunk"), (32, "... four steps before the end"), (64, "... one round every second step"),
            (128, "every activation load from the same 4 KB (same instructions, no memory traffic)")]
for name, (w, fn) in kernels.items():
    _hip.conv_bind_planes(w, planes[w.data_ptr()])
    print(name)
    for bits, what in variants:
        _hip.set_knob("CONV_EXP", bits)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"   {what:84s} {e0.elapsed_time(e1) * 10:7.2f} us per launch (back to back)", flush=True)
_hip.set_knob("CONV_EXP", 0)
