#!/usr/bin/env python3
"""Confidence check: the first training step's gradients at BASELINE config 2 (bench.py's workload), N times with the side
streams on against once on a single stream; prints the largest deviation per run relative to the largest gradient.
Usage: python tools/repeat_check.py [runs=5] [io=fp32|all]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from glow_tts_train.train import train_batch  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
io = sys.argv[2] if len(sys.argv) > 2 else "fp32"
sys.argv = [sys.argv[0]]
args = bench.parse()


_START = {}


def run(side):
    os.environ["GLOWTTS_SIDE_STREAM"] = "1" if side else "0"
    torch.manual_seed(1234)
    model, opt, batch, cfg = bench.build_workload(args, torch.device("cuda:0"), 0)
    # identical starting parameters in every run: the data-dependent ActNorm init inside build_workload reduces with float
    # atomics, and bf16 tensors turn a 1e-7 difference of a scale into whole bf16 steps a few blocks later
    if "p" not in _START:
        _START["p"] = opt._optim.flat_p.detach().clone()
    else:
        opt._optim.flat_p.copy_(_START["p"])
    for m in model.modules():                                  # dropout off: the generator's state is not part of the check
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    for f in model.decoder.flows:
        if hasattr(f, "wn"):
            f.wn.p_dropout = 0.0
    model.decoder.io_bf16 = False if io == "fp32" else io
    train_batch(model, opt, batch, cfg.grad_clip, None)
    torch.cuda.synchronize()
    names = {id(p): n for n, p in model.named_parameters()}
    fo = opt._optim
    return fo.flat_g.detach().clone(), [(names[id(p)], o, p.numel()) for p, o in zip(fo._params, fo.offsets)]


ref, layout = run(False)
gmax = float(ref.abs().max())
for i in range(runs):
    g, _ = run(True)
    d = (g - ref).abs()
    worst = float(d.max()) / gmax
    where = ""
    if worst > 5e-6:
        j = int(d.argmax())
        where = next(n for n, o, num in layout if o <= j < o + num)
    print(f"run {i}: max |dg| / gmax = {worst:.2e} {where}")
