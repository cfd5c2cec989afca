#!/usr/bin/env python3
"""Phase timeline of one MAS launch (tuning tool; trace build: `make -C glow-tts-train_amd/csrc trace`)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
os.environ.setdefault("GLOWTTS_HIP_LIB", os.path.join(ROOT, "tools", "libglowtts_trace.bin"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from glow_tts_train import _hip, monotonic_align as M  # noqa: E402

B, Tx, Ty = 32, 160, 800
v = torch.randn(B, Tx, Ty, device="cuda")
mask = torch.ones(B, Tx, Ty, device="cuda")
for _ in range(3):
    M.maximum_path(v, mask)
torch.cuda.synchronize()
lib = _hip.load()
rd = lib.glowtts_debug_mas_trace_read
rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(1024 * 8, dtype=np.uint64)
rd(buf.ctypes.data, 1024 * 8)
tr = buf.reshape(1024, 8).astype(np.int64)[:B]
t0 = tr[:, 0].min()
names = ["start", "dp done (wave 0)", "all waves past dp", "backtrack done", "path written"]
for i, n in enumerate(names):
    print(f"{n:22s} median {np.median((tr[:, i] - t0) / 100.0):7.1f} us   max {((tr[:, i] - t0) / 100.0).max():7.1f}")
cyc = (tr[:, 6] - tr[:, 5]).astype(float)
wall = (tr[:, 1] - tr[:, 0]) / 100.0
print(f"shader clock over the DP phase: median {np.median(cyc / wall):.0f} cycles/us; {np.median(cyc) / Ty:.0f} cycles per column")
