#!/usr/bin/env python3
"""Race hunt (round 5): the whole training step at BASELINE configs[4] with RAGGED lengths (as tests/test_full_size_parity.py builds
it), in a chosen conv arithmetic: N repetitions with the side streams on against one single-stream reference, in ONE process, after a
configs[1]-sized step (so the step's zero arena is too small, as in the full test suite).  Lists every parameter whose gradient deviates
by more than 1e-3 of its own largest element.   python tools/race_hunt_c5.py [runs=20] [math=fp32]

What it found (round 5; DESIGN.md lesson 45): the (B, T') mask was missing from the tensors `record_stream`-ed on the weight-gradient
stream.  With GLOWTTS_WGRAD_MAIN_BLOCKS=0 (every block's weight gradients on the side stream) the first repetition failed every time before
the fix — tests/test_full_size_parity.py runs exactly that as a regression test."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "glow-tts-train_amd")]
import torch  # noqa: E402

from glow_tts_train import convops, models, optimize  # noqa: E402
from glow_tts_train.train import train_batch  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
math = sys.argv[2] if len(sys.argv) > 2 else "fp32"
# HUNT_SHAPE=c2: the benchmark's configuration instead (BASELINE configs[1]: 12 blocks, no speakers, B=32, T_mel=800: the flow-stack path)
C2 = os.environ.get("HUNT_SHAPE", "c5") == "c2"


def build(blocks, speakers, seed=41):
    torch.manual_seed(seed)
    m = models.FlowGenerator(n_vocab=148, hidden_channels=192, filter_channels=768, filter_channels_dp=256, out_channels=80, kernel_size=3,
                             n_heads=2, n_layers_enc=6, p_dropout=0.0, n_blocks_dec=blocks, kernel_size_dec=5, dilation_rate=1,
                             n_block_layers=4, p_dropout_dec=0.0, n_speakers=speakers, gin_channels=64 if speakers else 0, n_split=4, n_sqz=2,
                             sigmoid_scale=False, window_size=4, mean_only=False, prenet=True)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith(".end.weight"):
                p.copy_(0.02 * torch.randn_like(p))
            if k.endswith(".logs") or (k.endswith(".bias") and "flows" in k and p.dim() == 3):
                p.copy_(0.1 * torch.randn_like(p))
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m.cuda().train()


def ragged(b, tx, ty, seed, speakers):
    gen = torch.Generator().manual_seed(seed)
    yl = torch.linspace(ty, ty // 2, b).long()
    xl = (yl // 5).clamp(min=1)
    x = torch.randint(1, 148, (b, tx), generator=gen) * (torch.arange(tx)[None] < xl[:, None])
    y = torch.randn(b, 80, ty, generator=gen) * (torch.arange(ty)[None, None] < yl[:, None, None])
    spk = (torch.arange(b) % speakers) if speakers else None
    return tuple(None if t is None else t.cuda() for t in (x, xl, y, yl, spk))


def step(model, batch, side):
    os.environ["GLOWTTS_SIDE_STREAM"] = "1" if side else "0"
    opt = optimize.Adam(model.parameters(), scheduler="noam", dim_model=192, warmup_steps=4000, lr=1.0)
    flat = opt._optim
    p0 = flat.flat_p.detach().clone()
    train_batch(model, opt, batch, 5.0)
    torch.cuda.synchronize()
    g = flat.flat_g.detach().clone()
    flat.flat_p.copy_(p0)                                 # same parameters for the next repetition
    names = {id(p): n for n, p in model.named_parameters()}
    return g, [(names[id(p)], o, p.numel()) for p, o in zip(flat._params, flat.offsets)]


convops.set_conv_math(math)
small = build(12, 0)
step(small, ragged(32, 160, 800, 12, 0), True)            # a configs[1] step first: sizes the zero arena for the smaller model
del small
model = build(12, 0) if C2 else build(20, 4)
batch = ragged(32, 160, 800, 12, 0) if C2 else ragged(48, 240, 1200, 12, 4)
ref, layout = step(model, batch, False)
nbad = 0
for i in range(runs):
    g, _ = step(model, batch, True)
    bad = []
    for n, o, num in layout:
        r, q = ref[o:o + num], g[o:o + num]
        tm = float(r.abs().max())
        dv = float((q - r).abs().max())
        if tm > 1e-9 and not (dv <= 1e-3 * tm):           # (a NaN counts as a deviation)
            bad.append(f"{n}: {dv:.2e} of {tm:.2e}")
    nbad += bool(bad)
    print(f"run {i}: {len(bad)} parameter(s) off" + ("" if not bad else ": " + "; ".join(bad[:8])), flush=True)
print(f"{nbad} of {runs} repetitions had a deviating gradient")
