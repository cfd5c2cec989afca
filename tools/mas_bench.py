#!/usr/bin/env python3
"""The alignment search alone (csrc/mas.hip), HIP events around back-to-back launches: the single-wave kernel (GLOWTTS_MAS_WAVES=0)
against the multi-wave search + the path expansion from its spans, at the benchmark's lattice (32 x 160 x 800) and configs[4]'s
(48 x 240 x 1200).   python tools/mas_bench.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
import torch  # noqa: E402

from glow_tts_train import _hip  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
_hip.load()
P = _hip.ptr
cur = torch.cuda.current_stream()


def timeit(fn):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    for _ in range(reps):
        fn()
    e1.record(cur)
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for b, tx, ty in ((32, 160, 800), (48, 240, 1200), (32, 100, 400)):
    torch.manual_seed(tx)
    v = torch.randn(b, tx, ty, device="cuda") * 3
    txs = torch.full((b,), tx, device="cuda", dtype=torch.int32)
    tys = torch.full((b,), ty, device="cuda", dtype=torch.int32)
    path = torch.empty(b, tx, ty, device="cuda")
    first = torch.empty(b, tx + 1, device="cuda", dtype=torch.int32)
    tok = torch.empty(b, ty, device="cuda", dtype=torch.int32)
    both = lambda: _hip.call("glowtts_mas_path_spans", P(v), P(path), P(first), P(tok), P(txs), P(tys), b, tx, ty)   # noqa: E731
    search = lambda: _hip.call("glowtts_mas_path_spans", P(v), None, P(first), P(tok), P(txs), P(tys), b, tx, ty)    # noqa: E731
    expand = lambda: _hip.call("glowtts_mas_path_from_spans", P(first), P(path), b, tx, ty)                          # noqa: E731
    _hip.set_knob("GLOWTTS_MAS_WAVES", 0)
    t_old = timeit(both)
    ref = path.clone()
    _hip.set_knob("GLOWTTS_MAS_WAVES", 1)
    t_both = timeit(both)
    same = bool((path == ref).all())
    t_search, t_expand = timeit(search), timeit(expand)
    print(f"lattice {b} x {tx} x {ty}: single-wave kernel {t_old:7.1f} us | multi-wave search + path {t_both:7.1f} us "
          f"(search alone {t_search:6.1f}, path from spans {t_expand:5.1f}); paths identical: {same}", flush=True)
