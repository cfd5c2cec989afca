#!/usr/bin/env python3
"""Phase timeline of the attention q-block kernel (forward) — trace build: `make -C glow-tts-train_amd/csrc trace`."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
os.environ.setdefault("GLOWTTS_HIP_LIB", os.path.join(ROOT, "tools", "libglowtts_trace.bin"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from glow_tts_train import _hip, attentions  # noqa: E402

torch.manual_seed(0)
mha = attentions.MultiHeadAttention(192, 192, 2, window_size=4, p_dropout=0.1).cuda().train()
B, T = 32, 160
x = torch.randn(B, 192, T, device="cuda")
mask = torch.ones(B, 1, T, T, device="cuda")
with torch.no_grad():
    for _ in range(3):
        mha(x, x, attn_mask=mask)
torch.cuda.synchronize()
lib = _hip.load()
rd = lib.glowtts_debug_attn_trace_read
rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(1024 * 8, dtype=np.uint64)
rd(buf.ctypes.data, 1024 * 8)
tr = buf.reshape(1024, 8).astype(np.int64)
tr = tr[tr[:, 0] > 0]
t0 = tr[:, 0].min()
names = ["start", "S = QK^T done", "rel keys added", "softmax + P written", "PV done", "stored"]
print(len(tr), "workgroups")
for i, n in enumerate(names):
    v = (tr[:, i] - t0) / 100.0
    print(f"{n:22s} median {np.median(v):7.1f} us   max {v.max():7.1f}")
