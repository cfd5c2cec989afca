#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE, build-container only).

Runs the REAL reference (imported read-only from /root/reference, never copied) on small seeded inputs and
writes inputs + expected outputs as .npz fixtures under tests/golden/.  The fixtures are data only; the
reference itself never travels to the GPU box.

How the reference is made importable here:
  * sys.path gets /root/reference so `import glow_tts_train` resolves to the reference package;
  * `dataclasses_json` (absent in this image, imported by glow_tts_train/config.py:8) is stubbed with an empty
    DataClassJsonMixin — no reference code path used below touches it;
  * `glow_tts_train.monotonic_align.core` (the Cython kernel, core.pyx) is loaded from oracle/_ref/, which
    oracle/Makefile builds from the reference's own core.pyx where it lies.

Usage:  make -C oracle ref && python oracle/make_golden.py          (writes tests/golden/*.npz)
        python oracle/make_golden.py attention-long                  (only mha_c192_t240_w4 / mha_c64_t256_w4)
        python oracle/make_golden.py host                            (only the collate / table / config / checkpoint fixtures)
        python oracle/make_golden.py dropout                         (only e2e_dropout_train: a step with recorded dropout masks)

This script must be run in its own process: it imports the reference under the package name `glow_tts_train`,
which is also the name of this repo's drop-in package.
"""
import glob
import importlib.machinery
import importlib.util
import math
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = os.environ.get("GLOWTTS_REFERENCE", "/root/reference")

warnings.filterwarnings("ignore")


def import_reference():
    stub = types.ModuleType("dataclasses_json")

    class DataClassJsonMixin:  # noqa: D401 - empty stand-in for an absent third-party mixin
        pass

    stub.DataClassJsonMixin = DataClassJsonMixin
    sys.modules["dataclasses_json"] = stub
    sys.path.insert(0, REF)
    import glow_tts_train  # noqa: F401  (the reference package)

    assert os.path.realpath(glow_tts_train.__path__[0]).startswith(os.path.realpath(REF))
    so = glob.glob(os.path.join(HERE, "_ref", "core*.so"))
    assert so, "run `make -C oracle ref` first"
    name = "glow_tts_train.monotonic_align.core"
    loader = importlib.machinery.ExtensionFileLoader(name, so[0])
    spec = importlib.util.spec_from_loader(name, loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    sys.modules[name] = mod
    from glow_tts_train import attentions, layers, models, monotonic_align, optimize, utils

    return types.SimpleNamespace(
        attentions=attentions, layers=layers, models=models, monotonic_align=monotonic_align,
        optimize=optimize, utils=utils, core=mod,
    )


def npy(t):
    if t is None:
        return np.zeros((0,), np.float32)
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy().copy()
    return np.asarray(t)


def sd_dict(module, prefix="sd."):
    return {prefix + k: npy(v) for k, v in module.state_dict().items()}


def grads_dict(module, prefix="grad."):
    return {prefix + k: npy(p.grad) for k, p in module.named_parameters() if p.grad is not None}


def ragged_mask(lengths, t):
    lengths = torch.as_tensor(lengths)
    return (torch.arange(t)[None, :] < lengths[:, None]).float().unsqueeze(1)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path)} bytes, {len(arrs)} arrays")


# ----------------------------------------------------------------------------------------------------------
def flow_case(R, make_flow, name, c, b=2, t=12, lengths=(12, 7), gin=0, perturb=None):
    """forward / reverse / autograd grads of one flow operator f(x, x_mask, g, reverse) -> (z, logdet)."""
    torch.manual_seed(1234)
    f = make_flow()
    if perturb is not None:
        perturb(f)
    f.train()
    x = torch.randn(b, c, t)
    mask = ragged_mask(lengths, t)
    x = (x * mask).requires_grad_(True)
    g = torch.randn(b, gin, 1) if gin else None
    if g is not None:
        g.requires_grad_(True)
    r = torch.randn(b, c, t)
    s = torch.randn(b)
    z, logdet = f(x, mask, g=g, reverse=False)
    loss = (z * r).sum() + (logdet * s).sum()
    loss.backward()
    arrs = dict(x=npy(x), mask=npy(mask), r=npy(r), s=npy(s), z=npy(z), logdet=npy(logdet), dx=npy(x.grad))
    if g is not None:
        arrs["g"] = npy(g)
        arrs["dg"] = npy(g.grad)
    arrs.update(sd_dict(f))
    arrs.update(grads_dict(f))
    # reverse (inference) path on z: must give x back on the unmasked region
    with torch.no_grad():
        f.eval()
        # state-dict BEFORE store_inverse is what is saved above; store_inverse may strip weight-norm
        f.store_inverse()
        xr, ld_r = f(z.detach(), mask, g=None if g is None else g.detach(), reverse=True)
        assert ld_r is None
        arrs["x_rev"] = npy(xr)
    save(name, **arrs)


def gen_flows(R):
    L, A = R.layers, R.attentions

    def pert_actnorm(f):
        with torch.no_grad():
            f.logs.copy_(0.3 * torch.randn_like(f.logs))
            f.bias.copy_(0.5 * torch.randn_like(f.bias))

    flow_case(R, lambda: L.ActNorm(8), "actnorm_c8", c=8, perturb=pert_actnorm)
    flow_case(R, lambda: L.ActNorm(160), "actnorm_c160", c=160, b=3, t=10, lengths=(10, 6, 1),
              perturb=pert_actnorm)

    def pert_invconv(f):
        with torch.no_grad():
            f.weight.add_(0.2 * torch.randn_like(f.weight))
            if torch.det(f.weight) < 0:
                f.weight[:, 0] *= -1

    flow_case(R, lambda: L.InvConvNear(8, n_split=4), "invconv_c8_s4", c=8, perturb=pert_invconv)
    flow_case(R, lambda: L.InvConvNear(8, n_split=2), "invconv_c8_s2", c=8, perturb=pert_invconv)
    flow_case(R, lambda: L.InvConvNear(160, n_split=4), "invconv_c160_s4", c=160, b=3, t=10,
              lengths=(10, 6, 1), perturb=pert_invconv)

    def pert_coupling(f):
        # the end conv is zero-initialised (attentions.py:104-106); give it weights so m/logs are exercised
        with torch.no_grad():
            f.end.weight.copy_(0.1 * torch.randn_like(f.end.weight))
            f.end.bias.copy_(0.1 * torch.randn_like(f.end.bias))

    for sig in (False, True):
        for gin in (0, 8):
            flow_case(
                R,
                lambda: A.CouplingBlock(8, 16, kernel_size=5, dilation_rate=1, n_layers=3,
                                        gin_channels=gin, p_dropout=0.0, sigmoid_scale=sig),
                f"coupling_c8_h16_sig{int(sig)}_gin{gin}", c=8, gin=gin, perturb=pert_coupling,
            )
    flow_case(
        R,
        lambda: A.CouplingBlock(8, 16, kernel_size=3, dilation_rate=2, n_layers=3, gin_channels=0,
                                p_dropout=0.0, sigmoid_scale=False),
        "coupling_c8_h16_k3_d2", c=8, t=20, lengths=(20, 13), perturb=pert_coupling,
    )

    # ActNorm data-dependent initialisation (layers.py:207-221)
    torch.manual_seed(1234)
    f = L.ActNorm(8, ddi=True)
    x = torch.randn(3, 8, 12) * 2.0 + 0.7
    mask = ragged_mask((12, 9, 4), 12)
    z, logdet = f(x * mask, mask)
    save("actnorm_ddi", x=npy(x * mask), mask=npy(mask), z=npy(z), logdet=npy(logdet), **sd_dict(f))


def gen_wn_gate_squeeze(R):
    L, U = R.layers, R.utils
    for gin in (0, 8):
        torch.manual_seed(1234)
        wn = L.WN(16, 16, 5, 1, 3, gin_channels=gin, p_dropout=0.0)
        x = torch.randn(2, 16, 12)
        mask = ragged_mask((12, 7), 12)
        x = (x * mask).requires_grad_(True)
        g = torch.randn(2, gin, 1).requires_grad_(True) if gin else None
        r = torch.randn(2, 16, 12)
        out = wn(x, mask, g)
        (out * r).sum().backward()
        arrs = dict(x=npy(x), mask=npy(mask), r=npy(r), out=npy(out), dx=npy(x.grad))
        if g is not None:
            arrs.update(g=npy(g), dg=npy(g.grad))
        arrs.update(sd_dict(wn))
        arrs.update(grads_dict(wn))
        save(f"wn_h16_gin{gin}", **arrs)

    torch.manual_seed(1234)
    a = torch.randn(2, 32, 9, requires_grad=True)
    bb = torch.randn(2, 32, 1, requires_grad=True)
    r = torch.randn(2, 16, 9)
    acts = U.fused_add_tanh_sigmoid_multiply(a, bb, torch.IntTensor([16]))
    (acts * r).sum().backward()
    save("gate_h16", a=npy(a), b=npy(bb), r=npy(r), acts=npy(acts), da=npy(a.grad), db=npy(bb.grad))

    torch.manual_seed(1234)
    x = torch.randn(3, 6, 11)
    mask = ragged_mask((11, 8, 3), 11)
    xs, ms = U.squeeze(x, mask, 2)
    xu, mu = U.unsqueeze(xs, ms, 2)
    save("squeeze_c6_t11", x=npy(x), mask=npy(mask), x_sqz=npy(xs), mask_sqz=npy(ms), x_unsqz=npy(xu),
         mask_unsqz=npy(mu))


# the attention kernel's upper envelope (T = 240: config 5's text length; T = 256: the kernel's limit), added in round 2
ATTENTION_LONG = [
    ("mha_c192_t240_w4", 240, (240, 151), 4, None, 192),
    ("mha_c64_t256_w4", 256, (256, 199), 4, None, 64),
    # round 4: beyond 256 tokens (the attention kernel's LONG form: 256 < T <= 512)
    ("mha_c192_t300_w4", 300, (300, 217), 4, None, 192),
    ("mha_c64_t512_w4", 512, (512, 399), 4, None, 64),
]


def gen_attention(R, only_long=False):
    A = R.attentions
    cases = [
        ("mha_t12_w4", 12, (12, 7), 4, None),      # length > window+1 : pad branch (attentions.py:290-294)
        ("mha_t4_w4", 4, (4, 3), 4, None),         # length < window+1 : slice branch (:288,297-299)
        ("mha_t5_w4", 5, (5, 2), 4, None),         # length == window+1
        ("mha_t12_w4_blk3", 12, (12, 9), 4, 3),    # block-local mask (:241-249)
        ("mha_t12_nowin", 12, (12, 7), None, None),
        ("mha_t70_w4", 70, (70, 33), 4, None),     # spans more than one 64-wide tile
    ]
    # (name, t, lengths, window, block, channels): channels 32/192 give d_k = 16/96, the MFMA kernel's envelope
    cases = [c + (16,) for c in cases] + [
        ("mha_c32_t70_w4", 70, (70, 33), 4, None, 32),
        ("mha_c32_t12_w4_blk3", 12, (12, 9), 4, 3, 32),
        ("mha_c32_t5_w4", 5, (5, 2), 4, None, 32),
        ("mha_c32_t40_nowin", 40, (40, 17), None, None, 32),
        ("mha_c192_t160_w4", 160, (160, 101), 4, None, 192),
    ] + ATTENTION_LONG
    if only_long:
        cases = ATTENTION_LONG
    for name, t, lengths, win, blk, ch in cases:
        torch.manual_seed(1234)
        m = A.MultiHeadAttention(ch, ch, 2, window_size=win, p_dropout=0.0, block_length=blk)
        x = torch.randn(2, ch, t)
        mask = ragged_mask(lengths, t)
        x = (x * mask).requires_grad_(True)
        attn_mask = mask.unsqueeze(2) * mask.unsqueeze(-1)
        r = torch.randn(2, ch, t)
        y = m(x, x, attn_mask)
        (y * r).sum().backward()
        arrs = dict(x=npy(x), mask=npy(mask), r=npy(r), y=npy(y), p_attn=npy(m.attn), dx=npy(x.grad),
                    window=np.int64(-1 if win is None else win), block=np.int64(-1 if blk is None else blk))
        arrs.update(sd_dict(m))
        arrs.update(grads_dict(m))
        save(name, **arrs)


def gen_mas(R):
    rng = np.random.RandomState(1234)
    cases = []

    def add(b, tx, ty, txs, tys, quant=None, scale=1.0):
        v = (rng.randn(b, tx, ty) * scale).astype(np.float32)
        if quant:
            v = np.round(v * quant) / quant  # many exact ties
            v = v.astype(np.float32)
        cases.append((v, np.asarray(txs, np.int32), np.asarray(tys, np.int32)))

    add(1, 1, 1, [1], [1])
    add(1, 1, 7, [1], [7])
    add(1, 5, 5, [5], [5])                       # t_x == t_y : forced diagonal
    add(2, 4, 9, [4, 2], [9, 5])
    add(3, 9, 40, [9, 7, 3], [40, 31, 12])
    add(3, 9, 40, [9, 7, 3], [40, 31, 12], quant=2)
    add(4, 20, 64, [20, 17, 11, 1], [64, 60, 33, 2])
    add(2, 64, 64, [64, 40], [64, 41])
    add(2, 65, 130, [65, 64], [130, 65])        # crosses the 64-lane boundary
    add(2, 100, 400, [100, 77], [400, 311])
    add(2, 100, 400, [100, 77], [400, 311], quant=1)
    add(1, 130, 257, [130], [257], scale=30.0)
    add(5, 33, 97, [33, 32, 31, 2, 1], [97, 96, 64, 63, 1])
    add(2, 7, 300, [7, 1], [300, 299])
    add(1, 160, 800, [160], [800])
    add(2, 200, 256, [200, 129], [256, 255], quant=4)
    add(3, 3, 3, [3, 2, 1], [3, 3, 3])
    add(2, 16, 17, [16, 15], [17, 16])
    add(1, 250, 251, [250], [251])
    add(2, 48, 1000, [48, 3], [1000, 777])
    add(1, 2, 2, [1], [2])
    add(2, 12, 30, [12, 0], [30, 0])            # an empty utterance in the batch
    arrs = {"n": np.int64(len(cases))}
    for i, (v, txs, tys) in enumerate(cases):
        b, tx, ty = v.shape
        mask = np.zeros((b, tx, ty), np.float32)
        for j in range(b):
            mask[j, : txs[j], : tys[j]] = 1.0
        # (1) through the reference's Python wrapper (monotonic_align/__init__.py:6-21)
        if (txs > 0).all() and (tys > 0).all():
            path = R.monotonic_align.maximum_path(torch.from_numpy(v), torch.from_numpy(mask)).numpy()
            path = path.astype(np.int8)
        else:
            # wrapper derives t_x from mask[:, :, 0] which is fine, but keep the raw-kernel call for the
            # degenerate case so the fixture states exactly what the kernel does
            vv = (v * mask).astype(np.float32)
            p = np.zeros_like(vv, dtype=np.int32)
            R.core.maximum_path_c(p, vv, txs, tys)
            path = p.astype(np.int8)
        arrs[f"value{i}"] = v
        arrs[f"tx{i}"] = txs
        arrs[f"ty{i}"] = tys
        arrs[f"path{i}"] = path
    save("mas_cases", **arrs)


def small_generator(R, gin=0, n_speakers=0, mean_only=True, n_split=4, sigmoid_scale=False):
    return R.models.FlowGenerator(
        n_vocab=148, hidden_channels=32, filter_channels=64, filter_channels_dp=32, out_channels=80,
        kernel_size=3, n_heads=2, n_layers_enc=2, p_dropout=0.0, n_blocks_dec=2, kernel_size_dec=5,
        dilation_rate=1, n_block_layers=2, p_dropout_dec=0.0, n_speakers=n_speakers, gin_channels=gin,
        n_split=n_split, n_sqz=2, sigmoid_scale=sigmoid_scale, window_size=4, block_length=None,
        mean_only=mean_only, hidden_channels_enc=32, hidden_channels_dec=32, prenet=True,
    )


def e2e_inputs(seed=1234):
    g = torch.Generator().manual_seed(seed)
    b, tx, ty = 3, 9, 41
    x_lengths = torch.tensor([9, 7, 4])
    y_lengths = torch.tensor([41, 30, 17])
    x = torch.randint(1, 148, (b, tx), generator=g)
    x = x * (torch.arange(tx)[None] < x_lengths[:, None])
    y = torch.randn(b, 80, ty, generator=g)
    y = y * (torch.arange(ty)[None, None] < y_lengths[:, None, None])
    return x, x_lengths, y, y_lengths


def gen_e2e(R):
    U = R.utils
    for tag, kw in (("base", {}), ("spk", dict(gin=8, n_speakers=3, mean_only=False, sigmoid_scale=True))):
        torch.manual_seed(1234)
        model = small_generator(R, **kw)
        # un-zero the coupling end convs and make ActNorm non-trivial so every path carries signal
        with torch.no_grad():
            for f in model.decoder.flows:
                if hasattr(f, "end"):
                    f.end.weight.copy_(0.05 * torch.randn_like(f.end.weight))
                    f.end.bias.copy_(0.05 * torch.randn_like(f.end.bias))
                if hasattr(f, "logs") and hasattr(f, "bias"):
                    f.logs.copy_(0.1 * torch.randn_like(f.logs))
                    f.bias.copy_(0.1 * torch.randn_like(f.bias))
            model.encoder.pre.proj.weight.copy_(0.05 * torch.randn_like(model.encoder.pre.proj.weight))
        model.train()
        # the prenet hard-codes Dropout(0.5) (models.py:100, layers.py:58): RNG cannot match across
        # implementations, so parity fixtures run with every dropout probability at 0 (SURVEY.md §7 hard part 5)
        for mod in model.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        x, xl, y, yl = e2e_inputs()
        spk = torch.tensor([0, 2, 1]) if kw else None
        arrs = dict(x=npy(x), x_lengths=npy(xl), y=npy(y), y_lengths=npy(yl))
        if spk is not None:
            arrs["speaker_ids"] = npy(spk)
        arrs.update(sd_dict(model))

        (z, z_m, z_logs, logdet, z_mask), (x_m, x_logs, x_mask), (attn, logw, logw_) = model(x, xl, y, yl, g=spk)
        l_mle = U.mle_loss(z, z_m, z_logs, logdet, z_mask)
        l_len = U.duration_loss(logw, logw_, xl)
        (l_mle + l_len).backward()
        arrs.update(z=npy(z), z_m=npy(z_m), z_logs=npy(z_logs), logdet=npy(logdet), z_mask=npy(z_mask),
                    x_m=npy(x_m), x_logs=npy(x_logs), x_mask=npy(x_mask), attn=npy(attn).astype(np.int8),
                    logw=npy(logw), logw_=npy(logw_), l_mle=npy(l_mle), l_length=npy(l_len))
        arrs.update(grads_dict(model))

        # clip + three Adam/Noam steps on these (fixed) grads: optimize.py:8-64, utils.py:118-132
        total_norm = U.clip_grad_value_(model.parameters(), 5.0)
        arrs["total_norm"] = np.float64(total_norm)
        opt = R.optimize.Adam(model.parameters(), scheduler="noam", dim_model=32, warmup_steps=4000, lr=1.0,
                              betas=(0.9, 0.98), eps=1e-9)
        lrs = [opt.cur_lr]
        for _ in range(3):
            opt.step()
            lrs.append(opt.cur_lr)
        arrs["lrs"] = np.asarray(lrs, np.float64)
        arrs.update({"sd_after3." + k: npy(v) for k, v in model.state_dict().items()})
        save(f"e2e_{tag}_train", **arrs)

        # inference (gen=True) with injected noise after store_inverse() (infer.py:116, models.py:326-359)
        torch.manual_seed(1234)
        model2 = small_generator(R, **kw)
        model2.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in arrs.items() if k.startswith("sd.")})
        model2.eval()
        model2.decoder.store_inverse()
        noise_holder = {}
        orig = torch.randn_like

        def fixed_randn_like(t, *a, **k):
            gg = torch.Generator().manual_seed(77)
            n = torch.randn(t.shape, generator=gg, dtype=t.dtype)
            noise_holder["noise"] = n
            return n

        torch.randn_like = fixed_randn_like
        try:
            with torch.no_grad():
                (ymel, gz_m, gz_logs, gld, gz_mask), _, (gattn, glogw, glogw_) = model2(
                    x, xl, g=spk, gen=True, noise_scale=0.667, length_scale=1.0)
        finally:
            torch.randn_like = orig
        assert gld is None
        save(f"e2e_{tag}_gen", x=npy(x), x_lengths=npy(xl),
             **({"speaker_ids": npy(spk)} if spk is not None else {}),
             noise=npy(noise_holder["noise"]), y=npy(ymel), z_m=npy(gz_m), z_logs=npy(gz_logs),
             z_mask=npy(gz_mask), attn=npy(gattn).astype(np.int8), logw=npy(glogw), logw_=npy(glogw_),
             noise_scale=np.float64(0.667))


def dropout_site_names(n_pre, n_enc, n_dp, n_blocks, n_block_layers):
    """The reference's F.dropout call order within ONE training forward (models.py:120-142 -> layers.py:73-80, attentions.py:62-74,
    :251, :379, models.py:41-51; then models.py:193-211 -> layers.py:147), as the oracle's KeepMasks site names."""
    names = [f"encoder.pre.{i}" for i in range(n_pre)]
    for i in range(n_enc):
        names += [f"encoder.encoder.{i}.attn", f"encoder.encoder.{i}.y1", f"encoder.encoder.{i}.ffn", f"encoder.encoder.{i}.y2"]
    names += [f"encoder.proj_w.{i}" for i in range(n_dp)]
    for blk in range(n_blocks):
        names += [f"decoder.flows.{3 * blk + 2}.wn.{l}" for l in range(n_block_layers)]
    return names


def gen_e2e_dropout(R):
    """Round 4: the training step WITH dropout (the state bench.py times).  torch.nn.functional.dropout is patched for the run:
    it draws its keep decisions from a seeded generator, applies what F.dropout applies (x * keep / (1 - p)) and records
    (keep, p) in call order — the decisions become data, so the oracle and the HIP path can be fed the same ones."""
    import torch.nn.functional as TF

    U = R.utils
    torch.manual_seed(1234)
    model = R.models.FlowGenerator(
        n_vocab=148, hidden_channels=32, filter_channels=64, filter_channels_dp=32, out_channels=80,
        kernel_size=3, n_heads=2, n_layers_enc=2, p_dropout=0.1, n_blocks_dec=2, kernel_size_dec=5,
        dilation_rate=1, n_block_layers=2, p_dropout_dec=0.05, n_speakers=0, gin_channels=0,
        n_split=4, n_sqz=2, sigmoid_scale=False, window_size=4, block_length=None,
        mean_only=True, hidden_channels_enc=32, hidden_channels_dec=32, prenet=True)
    with torch.no_grad():
        for f in model.decoder.flows:
            if hasattr(f, "end"):
                f.end.weight.copy_(0.05 * torch.randn_like(f.end.weight))
                f.end.bias.copy_(0.05 * torch.randn_like(f.end.bias))
            if hasattr(f, "logs") and hasattr(f, "bias"):
                f.logs.copy_(0.1 * torch.randn_like(f.logs))
                f.bias.copy_(0.1 * torch.randn_like(f.bias))
        model.encoder.pre.proj.weight.copy_(0.05 * torch.randn_like(model.encoder.pre.proj.weight))
    model.train()
    x, xl, y, yl = e2e_inputs()
    arrs = dict(x=npy(x), x_lengths=npy(xl), y=npy(y), y_lengths=npy(yl))
    arrs.update(sd_dict(model))

    gen = torch.Generator().manual_seed(4321)
    record = []
    orig = TF.dropout

    def recording_dropout(inp, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return inp
        keep = (torch.rand(inp.shape, generator=gen) >= p)
        record.append((keep, float(p)))
        return inp * keep.to(inp.dtype) / (1.0 - p)

    TF.dropout = recording_dropout
    try:
        (z, z_m, z_logs, logdet, z_mask), (x_m, x_logs, x_mask), (attn, logw, logw_) = model(x, xl, y, yl)
        l_mle = U.mle_loss(z, z_m, z_logs, logdet, z_mask)
        l_len = U.duration_loss(logw, logw_, xl)
        (l_mle + l_len).backward()
    finally:
        TF.dropout = orig
    names = dropout_site_names(3, 2, 2, 2, 2)
    assert len(record) == len(names), (len(record), len(names))
    ps = {"encoder.pre": 0.5, "encoder.encoder": 0.1, "encoder.proj_w": 0.1, "decoder.flows": 0.05}
    for n, (keep, p) in zip(names, record):
        assert abs(p - ps[".".join(n.split(".")[:2])]) < 1e-12, (n, p)
        arrs["keep." + n] = npy(keep).astype(np.uint8)
        arrs["p." + n] = np.float64(p)
    assert arrs["keep.encoder.encoder.0.attn"].shape == (3, 2, 9, 9) and arrs["keep.decoder.flows.5.wn.1"].shape == (3, 64, 20)
    arrs.update(z=npy(z), z_m=npy(z_m), z_logs=npy(z_logs), logdet=npy(logdet), z_mask=npy(z_mask),
                x_m=npy(x_m), x_logs=npy(x_logs), x_mask=npy(x_mask), attn=npy(attn).astype(np.int8),
                logw=npy(logw), logw_=npy(logw_), l_mle=npy(l_mle), l_length=npy(l_len))
    arrs.update(grads_dict(model))
    save("e2e_dropout_train", **arrs)


def gen_losses(R):
    U = R.utils
    torch.manual_seed(1234)
    b, c, t = 3, 8, 14
    mask = ragged_mask((14, 10, 6), t)
    z = (torch.randn(b, c, t) * mask).requires_grad_(True)
    m = (torch.randn(b, c, t) * mask).requires_grad_(True)
    logs = (0.3 * torch.randn(b, c, t) * mask).requires_grad_(True)
    logdet = torch.randn(b, requires_grad=True)
    loss = U.mle_loss(z, m, logs, logdet, mask)
    loss.backward()
    logw = torch.randn(b, 1, 5, requires_grad=True)
    logw_ = torch.randn(b, 1, 5)
    lens = torch.tensor([5, 4, 2])
    dl = U.duration_loss(logw, logw_, lens)
    dl.backward()
    save("losses", z=npy(z), m=npy(m), logs=npy(logs), logdet=npy(logdet), mask=npy(mask), loss=npy(loss),
         dz=npy(z.grad), dm=npy(m.grad), dlogs=npy(logs.grad), dlogdet=npy(logdet.grad),
         logw=npy(logw), logw_=npy(logw_), lengths=npy(lens), dur_loss=npy(dl), dlogw=npy(logw.grad))


def numpy_scalar_globals():
    import numpy._core.multiarray as ncm
    return [ncm.scalar, np.dtype, type(np.dtype(np.float64)), type(np.dtype(np.int64)), type(np.dtype(np.float32))]


def gen_host(R):
    """Fixtures for the data formats either side of the path: the reference's collate on ragged
    utterance lists, its text / mel table readers, the field defaults of its config classes, and a checkpoint FILE
    written by its own save_checkpoint() after two optimisation steps of a tiny model."""
    import dataclasses
    import io
    import json
    from pathlib import Path

    from glow_tts_train import checkpoint as ref_ckpt
    from glow_tts_train import config as ref_cfg
    from glow_tts_train import dataset as ref_data

    # -- collate --------------------------------------------------------------------------------------------
    arrs = {}
    cases = [("a", 1, False, [5, 9, 3, 7], [21, 40, 13, 30]),
             ("b", 2, True, [6, 6, 2, 11, 6], [25, 27, 9, 45, 26]),          # ties in text length, odd mel maximum
             ("c", 4, True, [1], [7])]
    for tag, nfps, multi, tls, mls in cases:
        gen = torch.Generator().manual_seed(len(tls) * 100 + nfps)
        batch = []
        for i, (tl, ml) in enumerate(zip(tls, mls)):
            text = torch.randint(1, 148, (tl,), generator=gen, dtype=torch.int32)
            mel = torch.randn(8, ml, generator=gen)
            batch.append((text, mel, tl, (i * 7) % 4) if multi else (text, mel, tl))
        out = ref_data.PhonemeMelCollate(n_frames_per_step=nfps, multispeaker=multi)(batch)
        arrs[f"{tag}.n_frames_per_step"] = np.int64(nfps)
        arrs[f"{tag}.multispeaker"] = np.int64(multi)
        arrs[f"{tag}.text_lengths"] = np.asarray(tls, np.int64)
        arrs[f"{tag}.mel_lengths"] = np.asarray(mls, np.int64)
        arrs[f"{tag}.texts"] = np.concatenate([npy(b[0]) for b in batch])
        arrs[f"{tag}.mels"] = np.concatenate([npy(b[1]) for b in batch], axis=1)
        arrs[f"{tag}.speakers"] = np.asarray([b[3] for b in batch] if multi else [], np.int64)
        for name, t in zip(("text_padded", "input_lengths", "mel_padded", "output_lengths", "speaker_ids"), out):
            arrs[f"{tag}.out.{name}"] = npy(t)
            if t is not None:
                arrs[f"{tag}.out.{name}.dtype"] = np.asarray(str(t.dtype))
    # -- table readers --------------------------------------------------------------------------------------
    csv_text = "utt1|1 2 3 4\nutt2|5 6\nutt3|7 8 9 10 11 12 13\nutt4| 14 15 16 \n"
    jsonl_text = '{"id": "utt1", "mel": [[0.5, 1.5, 2.5], [3.0, 4.0, 5.0]]}\n\n{"id": "utt4", "mel": [[-1.0], [2.0]]}\n'
    cfg = ref_cfg.TrainingConfig(min_seq_length=3, max_seq_length=6)
    ph = ref_data.load_phonemes(io.StringIO(csv_text), cfg)
    mels = ref_data.load_mels(io.StringIO(jsonl_text))
    arrs["tables.csv"] = np.asarray(csv_text)
    arrs["tables.jsonl"] = np.asarray(jsonl_text)
    arrs["tables.phoneme_ids"] = np.asarray(sorted(ph))
    for k, v in ph.items():
        arrs[f"tables.phonemes.{k}"] = npy(v)
        arrs[f"tables.phonemes.{k}.dtype"] = np.asarray(str(v.dtype))
    arrs["tables.mel_ids"] = np.asarray(sorted(mels))
    for k, v in mels.items():
        arrs[f"tables.mels.{k}"] = npy(v)
    save("host_dataset", **arrs)

    # -- config defaults ------------------------------------------------------------------------------------
    with open(os.path.join(OUT, "host_config_defaults.json"), "w") as f:
        json.dump(dataclasses.asdict(ref_cfg.TrainingConfig()), f, indent=1, sort_keys=True)

    # -- checkpoint written by the reference ---------------------------------------------------------------
    U = R.utils
    mc = ref_cfg.ModelConfig(num_symbols=20, hidden_channels=16, filter_channels=32, filter_channels_dp=16,
                             n_blocks_dec=2, n_layers_enc=1, n_block_layers=2, hidden_channels_enc=16,
                             hidden_channels_dec=16, p_dropout=0.0, p_dropout_dec=0.0, window_size=4, prenet=False)
    cfg = ref_cfg.TrainingConfig(model=mc, audio=ref_cfg.AudioConfig(mel_channels=8), warmup_steps=10)
    torch.manual_seed(1234)
    model, opt = R.models.setup_model(cfg, use_cuda=False)
    with torch.no_grad():
        for fl in model.decoder.flows:
            if hasattr(fl, "end"):
                fl.end.weight.copy_(0.05 * torch.randn_like(fl.end.weight))
                fl.end.bias.copy_(0.05 * torch.randn_like(fl.end.bias))
    model.train()
    gen = torch.Generator().manual_seed(5)
    xl = torch.tensor([7, 5, 4])
    yl = torch.tensor([30, 22, 18])
    x = torch.randint(1, 20, (3, 7), generator=gen) * (torch.arange(7)[None] < xl[:, None])
    y = torch.randn(3, 8, 30, generator=gen) * (torch.arange(30)[None, None] < yl[:, None, None])

    def one_step():
        opt.zero_grad()
        (z, z_m, z_logs, logdet, z_mask), _, (_a, logw, logw_) = model(x, xl, y, yl, g=None)
        loss = U.mle_loss(z, z_m, z_logs, logdet, z_mask) + U.duration_loss(logw, logw_, xl)
        loss.backward()
        U.clip_grad_value_(model.parameters(), cfg.grad_clip)
        opt.step()
        return float(loss)

    losses = [one_step(), one_step()]
    path = Path(OUT) / "host_ref_checkpoint.pth"
    ref_ckpt.save_checkpoint(ref_ckpt.Checkpoint(model=model, optimizer=opt, learning_rate=opt.cur_lr, global_step=3,
                                                 version=cfg.version), path)
    # what the reference does with that file: load it into fresh objects, then take one more step
    # (the file holds numpy scalars — cur_lr comes out of np.power, optimize.py:35-42 — which torch >= 2.6 refuses
    # under its default weights_only=True; the reference's torch.load call predates that default, so allow them here)
    torch.manual_seed(99)
    with torch.serialization.safe_globals(numpy_scalar_globals()):
        ck = ref_ckpt.load_checkpoint(path, cfg, use_cuda=False)
    model, opt = ck.model, ck.optimizer
    # the reference's Adam wrapper restarts its Noam counter at 1 on load (optimize.py:28, :60-61) and only torch's
    # own per-parameter step survives; record what the third step then does
    model.train()
    lr_before = opt.cur_lr
    loss3 = one_step()
    save("host_ref_checkpoint_expect", x=npy(x), x_lengths=npy(xl), y=npy(y), y_lengths=npy(yl),
         losses=np.asarray(losses + [loss3], np.float64), learning_rate=np.float64(ck.learning_rate),
         global_step=np.int64(ck.global_step), version=np.int64(ck.version), lr_before_step3=np.float64(lr_before),
         lr_after_step3=np.float64(opt.cur_lr), step_num_after=np.int64(opt.step_num),
         model_config=np.asarray(json.dumps(dataclasses.asdict(mc))),
         **{"sd_after3." + k: npy(v) for k, v in model.state_dict().items()})
    print("host_ref_checkpoint.pth:", os.path.getsize(path), "bytes")


def main():
    R = import_reference()
    torch.set_num_threads(1)
    torch.use_deterministic_algorithms(False)
    if sys.argv[1:] == ["host"]:            # only the §8f-row-4 fixtures; the others stay as committed
        gen_host(R)
        return
    if sys.argv[1:] == ["attention-long"]:  # only the two long-sequence attention fixtures (round 2)
        gen_attention(R, only_long=True)
        return
    if sys.argv[1:] == ["dropout"]:         # only the training step with recorded dropout decisions (round 4)
        gen_e2e_dropout(R)
        return
    gen_host(R)
    gen_mas(R)
    gen_flows(R)
    gen_wn_gate_squeeze(R)
    gen_attention(R)
    gen_losses(R)
    gen_e2e(R)
    gen_e2e_dropout(R)
    tot = sum(os.path.getsize(p) for p in glob.glob(os.path.join(OUT, "*.npz")))
    print("total fixture bytes:", tot)


if __name__ == "__main__":
    main()
