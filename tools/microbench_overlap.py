#!/usr/bin/env python3
"""Do two co-resident workgroups of the gate kernel overlap?  Launch B=16 problems (<= 1 WG per CU) on ONE stream
back-to-back vs on TWO streams concurrently (two independent launches share the CUs, not in lockstep)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
import torch  # noqa: E402

from glow_tts_train import convops  # noqa: E402
from glow_tts_train._hip import load  # noqa: E402

lib = load()
B, H, T = 16, 192, 400
dev = "cuda"
torch.manual_seed(0)
v_in = torch.randn(2 * H, H, 5, device=dev) * 0.03
wf_in, _, _ = convops.pack_weight(v_in, None)
b_in = torch.zeros(2 * H, device=dev)


def mk():
    return (torch.randn(B, H, T, device=dev), torch.empty(B, H, T, device=dev), torch.empty(B, 2 * H, T, device=dev))


sets = [mk(), mk()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def launch(i, stream):
    x, acts, ts = sets[i]
    rc = lib.glowtts_conv_gate_fwd(x.data_ptr(), wf_in.data_ptr(), b_in.data_ptr(), None, None, 1.0, acts.data_ptr(),
                                   ts.data_ptr(), B, H, T, 5, 1, 2, stream.cuda_stream)
    assert rc == 0


for mode in ("one stream, 2 launches serial", "two streams, concurrent"):
    for _ in range(3):
        launch(0, streams[0]); launch(1, streams[0] if mode.startswith("one") else streams[1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        launch(0, streams[0])
        launch(1, streams[0] if mode.startswith("one") else streams[1])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e6
    print(f"{mode:34s}: {dt:7.1f} us per pair of B=16 gate launches")
