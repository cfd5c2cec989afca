#!/usr/bin/env python3
"""Race hunt: the first training step's gradients at BASELINE configs[4] (speakers, 20 blocks, B=48, T_mel=1200) in a chosen conv
arithmetic, N times with the side streams on against once on a single stream; lists EVERY parameter whose gradient deviates by more
than 1e-3 of its own largest element.   python tools/repeat_check_c5.py [runs=6] [math=fp32]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from glow_tts_train import convops  # noqa: E402
from glow_tts_train.train import train_batch  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
math = sys.argv[2] if len(sys.argv) > 2 else "fp32"
sys.argv = [sys.argv[0], "--batch", "48", "--t-mel", "1200", "--blocks", "20", "--speakers", "4"]
args = bench.parse()
_START = {}


def run(side):
    os.environ["GLOWTTS_SIDE_STREAM"] = "1" if side else "0"
    torch.manual_seed(1234)
    model, opt, batch, cfg = bench.build_workload(args, torch.device("cuda:0"), 0)
    convops.set_conv_math(math)
    if "p" not in _START:
        _START["p"] = opt._optim.flat_p.detach().clone()
    else:
        opt._optim.flat_p.copy_(_START["p"])
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    for f in model.decoder.flows:
        if hasattr(f, "wn"):
            f.wn.p_dropout = 0.0
    train_batch(model, opt, batch, cfg.grad_clip, None)
    torch.cuda.synchronize()
    names = {id(p): n for n, p in model.named_parameters()}
    fo = opt._optim
    return fo.flat_g.detach().clone(), [(names[id(p)], o, p.numel()) for p, o in zip(fo._params, fo.offsets)]


ref, layout = run(False)
for i in range(runs):
    g, _ = run(True)
    bad = []
    for n, o, num in layout:
        r, q = ref[o:o + num], g[o:o + num]
        tm = float(r.abs().max())
        dv = float((q - r).abs().max())
        if tm > 0 and dv > 1e-3 * tm:
            bad.append(f"{n}: {dv:.2e} of {tm:.2e}")
    print(f"run {i}: {len(bad)} parameter(s) off by more than 1e-3 of their largest element" + ("" if not bad else ": " + "; ".join(bad[:12])), flush=True)
