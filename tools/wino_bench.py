#!/usr/bin/env python3
"""The gated 5-tap in-conv at the benchmark's shape (B = 32, H = 192, T' = 400): the direct bf16x6 kernel against its Winograd
F(4, 5) form (csrc/convwino.hip), alone on the GPU, back to back, HIP events.   python tools/wino_bench.py [B] [T]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd")]
import torch  # noqa: E402

from glow_tts_train import _hip, convops  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
t = int(sys.argv[2]) if len(sys.argv) > 2 else 400
h = 192
dev = "cuda"
call, ptr = _hip.call, _hip.ptr
torch.manual_seed(0)
x = torch.randn(b, h, t, device=dev)
v_in = torch.randn(2 * h, h, 5, device=dev) * 0.03
b_in = torch.randn(2 * h, device=dev) * 0.1
wf, _, _ = convops.pack_weight(v_in, None)
keep = (torch.rand(b, 2 * h, t, device=dev) > 0.05).to(torch.uint8)
acts = torch.empty(b, h, t, device=dev)
ts = torch.empty(b, 2 * h, t, device=dev)
convops.set_conv_math("bf16x6+wrw")
planes = torch.empty(3 * wf.numel(), device=dev, dtype=torch.int16)
call("glowtts_conv_split_weights", ptr(wf), wf.numel(), ptr(planes))
_hip.conv_bind_planes(wf, planes)
n_u = _hip.wino_plane_elems(wf.numel())
u = torch.zeros(3 * n_u, device=dev, dtype=torch.int16)
table = torch.tensor([[0, h // 16, 2 * h]], dtype=torch.int64, device=dev)
call("glowtts_wino_weights", ptr(wf), wf.numel(), ptr(table), 1, ptr(u), n_u)
_hip.conv_bind_wino(wf, u)


# cold weights, as in the training step (48 different convolutions per step, each used once): NW weight sets in ONE packed buffer
NW = int(os.environ.get("WINO_BENCH_SETS", "1"))
if NW > 1:
    arena = torch.cat([wf.reshape(-1).clone() for _ in range(NW)])
    planes = torch.empty(3 * arena.numel(), device=dev, dtype=torch.int16)
    call("glowtts_conv_split_weights", ptr(arena), arena.numel(), ptr(planes))
    _hip.conv_bind_planes(arena, planes)
    n_u = _hip.wino_plane_elems(arena.numel())
    u = torch.zeros(3 * n_u, device=dev, dtype=torch.int16)
    table = torch.tensor([[i * wf.numel(), h // 16, 2 * h] for i in range(NW)], dtype=torch.int64, device=dev)
    call("glowtts_wino_weights", ptr(arena), arena.numel(), ptr(table), NW, ptr(u), n_u)
    _hip.conv_bind_wino(arena, u)
    sets = [arena[i * wf.numel():(i + 1) * wf.numel()] for i in range(NW)]
    xs = [torch.randn(b, h, t, device=dev) for _ in range(4)]
else:
    sets, xs = [wf], [x]


def run(n):
    for i in range(n):
        call("glowtts_conv_gate_fwd", ptr(xs[i % len(xs)]), ptr(sets[i % NW]), ptr(b_in), None, ptr(keep), 1.0 / 0.95, ptr(acts), ptr(ts), b, h, t, 5, 1, 2)


variants = (0, 1, 0, 1) + ((3, 7, 15, 17) if 'trace' in os.environ.get('GLOWTTS_HIP_LIB', '') else ())
for wino in variants:
    _hip.set_knob("WINO", wino)
    run(10)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run(200)
    e1.record()
    torch.cuda.synchronize()
    names = {0: 'direct bf16x6  ', 1: 'Winograd F(4,5)', 3: 'Winograd, no split / store (timing only)', 5: 'Winograd, no staging arithmetic (timing only)',
             7: 'Winograd, no staging at all (timing only)', 9: 'Winograd, split arithmetic with a third of the image stores (timing only)',
             11: 'Winograd, regions without the MFMA : VALU pattern', 13: 'Winograd, pattern 1 MFMA : 2 VALU', 17: 'Winograd, input loads waited for, no staging arithmetic (timing only)', 15: 'Winograd, six independent v_fma behind every MFMA pair instead of the staging (timing only)'}
    print(f"{names[wino]}: {e0.elapsed_time(e1) * 1e3 / 200:7.2f} us per launch (back to back, B={b}, T'={t}, {NW} weight set(s))", flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    call("glowtts_wino_weights", ptr(wf), wf.numel(), ptr(table), 1, ptr(u), n_u)
e1.record()
torch.cuda.synchronize()
print(f"weight transform of one convolution: {e0.elapsed_time(e1) * 1e3 / 50:.2f} us")
_hip.set_knob("WINO", 0)
