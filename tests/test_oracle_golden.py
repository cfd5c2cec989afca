"""CPU tests: the oracle (oracle/glow_oracle.py, oracle/mas_oracle.c) against golden vectors produced by the REAL
reference (oracle/make_golden.py).  This is what pins the oracle; the GPU tests then compare HIP against it.

Tolerance: 1e-5 abs/rel for fp32 CPU restatement vs reference; MAS exact.
"""
import glob
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, T, assert_close, load_golden, split_prefix
from oracle import glow_oracle as O

TOL = dict(rtol=2e-5, atol=2e-5)


def _grad_check(sd, golden, what, atol_of_max=0.0):
    """`atol_of_max`: extra absolute tolerance as a fraction of the gradient tensor's largest element — for sums of thousands of
    fp32 terms (the bias gradients of the 300- and 512-token attention fixtures), whose low bits depend on the summation order."""
    gg = split_prefix(golden, "grad.")
    assert gg, "fixture holds no grads"
    for k, want in gg.items():
        got = sd[k].grad
        assert got is not None, f"{what}: no grad for {k}"
        assert_close(got, want, what=f"{what} grad {k}", rtol=1e-4, atol=2e-5 + atol_of_max * float(np.abs(want).max()))


# ------------------------------------------------------------------------------------------------ MAS
def test_mas_oracle_vs_reference_golden():
    g = load_golden("mas_cases")
    n = int(g["n"])
    assert n >= 20
    for i in range(n):
        v, tx, ty, want = g[f"value{i}"], g[f"tx{i}"], g[f"ty{i}"], g[f"path{i}"]
        b, mx, my = v.shape
        mask = np.zeros_like(v)
        for j in range(b):
            mask[j, : tx[j], : ty[j]] = 1
        got = O.mas_numpy(v * mask, tx, ty)
        assert got.dtype == np.int32
        assert (got == want.astype(np.int32)).all(), f"MAS case {i} differs"
        # structural properties the maths implies
        for j in range(b):
            if tx[j] == 0 or ty[j] == 0:
                assert got[j].sum() == 0
                continue
            p = got[j, : tx[j], : ty[j]]
            assert got[j].sum() == ty[j]
            assert (p.sum(0) == 1).all()
            idx = p.argmax(0)
            assert (np.diff(idx) >= 0).all() and (np.diff(idx) <= 1).all()
            assert idx[-1] == tx[j] - 1
            if tx[j] <= ty[j]:
                assert idx[0] == 0 and (p.sum(1) >= 1).all()


def test_mas_c_matches_python_loops():
    rng = np.random.RandomState(7)
    for tx, ty in [(1, 1), (1, 5), (3, 3), (4, 9), (7, 8), (6, 20)]:
        v = np.round(rng.randn(tx, ty) * 2).astype(np.float32)  # ties on purpose
        want = O.mas_python_loops(v, tx, ty)
        got = O.mas_numpy(v[None], np.array([tx]), np.array([ty]))[0]
        assert (got == want).all()


def test_mas_oracle_vs_reference_kernel_live():
    """When oracle/_ref (the reference's own Cython kernel, built by oracle/Makefile) is present, compare live."""
    so = glob.glob(os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "core*.so"))
    if not so:
        pytest.skip("oracle/_ref not built (only possible where /root/reference exists)")
    import importlib.machinery
    import importlib.util

    loader = importlib.machinery.ExtensionFileLoader("core", so[0])
    spec = importlib.util.spec_from_loader("core", loader)
    core = importlib.util.module_from_spec(spec)
    loader.exec_module(core)
    rng = np.random.RandomState(99)
    for b, tx, ty in [(4, 37, 111), (2, 160, 800), (3, 64, 64)]:
        v = rng.randn(b, tx, ty).astype(np.float32)
        txs = rng.randint(1, tx + 1, size=b).astype(np.int32)
        tys = np.maximum(txs, rng.randint(1, ty + 1, size=b)).astype(np.int32)
        txs[0], tys[0] = tx, ty
        mask = np.zeros_like(v)
        for j in range(b):
            mask[j, : txs[j], : tys[j]] = 1
        vm = (v * mask).astype(np.float32)
        want = np.zeros_like(vm, dtype=np.int32)
        core.maximum_path_c(want, vm.copy(), txs, tys)
        got = O.mas_numpy(vm, txs, tys)
        assert (got == want).all()


# ------------------------------------------------------------------------------------------------ flows
@pytest.mark.parametrize("name", ["actnorm_c8", "actnorm_c160"])
def test_actnorm(name):
    g = load_golden(name)
    sd = split_prefix(g, "sd.", requires_grad=True)
    x = T(g["x"]).requires_grad_(True)
    mask = T(g["mask"])
    z, logdet = O.actnorm(x, mask, sd["logs"], sd["bias"])
    assert_close(z, g["z"], what="z", **TOL)
    assert_close(logdet, g["logdet"], what="logdet", **TOL)
    ((z * T(g["r"])).sum() + (logdet * T(g["s"])).sum()).backward()
    assert_close(x.grad, g["dx"], what="dx", **TOL)
    _grad_check(sd, g, name)
    xr, _ = O.actnorm(z.detach(), mask, sd["logs"].detach(), sd["bias"].detach(), reverse=True)
    assert_close(xr, g["x_rev"], what="x_rev", **TOL)
    assert_close(xr, x.detach() * mask, what="invertibility", rtol=1e-4, atol=1e-5)


def test_actnorm_ddi():
    g = load_golden("actnorm_ddi")
    x, mask = T(g["x"]), T(g["mask"])
    logs, bias = O.actnorm_init_stats(x, mask)
    assert_close(logs, g["sd.logs"], what="logs", **TOL)
    assert_close(bias, g["sd.bias"], what="bias", **TOL)
    z, logdet = O.actnorm(x, mask, logs, bias)
    assert_close(z, g["z"], what="z", **TOL)
    assert_close(logdet, g["logdet"], what="logdet", **TOL)


@pytest.mark.parametrize("name,n_split", [("invconv_c8_s4", 4), ("invconv_c8_s2", 2), ("invconv_c160_s4", 4)])
def test_invconv(name, n_split):
    g = load_golden(name)
    sd = split_prefix(g, "sd.", requires_grad=True)
    x = T(g["x"]).requires_grad_(True)
    mask = T(g["mask"])
    z, logdet = O.invconv(x, mask, sd["weight"], n_split)
    assert_close(z, g["z"], what="z", **TOL)
    assert_close(logdet, g["logdet"], what="logdet", **TOL)
    ((z * T(g["r"])).sum() + (logdet * T(g["s"])).sum()).backward()
    assert_close(x.grad, g["dx"], what="dx", **TOL)
    _grad_check(sd, g, name)
    xr, _ = O.invconv(z.detach(), mask, sd["weight"].detach(), n_split, reverse=True)
    assert_close(xr, g["x_rev"], what="x_rev", rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name,sig,gin,k,dil", [
    ("coupling_c8_h16_sig0_gin0", False, 0, 5, 1),
    ("coupling_c8_h16_sig0_gin8", False, 8, 5, 1),
    ("coupling_c8_h16_sig1_gin0", True, 0, 5, 1),
    ("coupling_c8_h16_sig1_gin8", True, 8, 5, 1),
    ("coupling_c8_h16_k3_d2", False, 0, 3, 2),
])
def test_coupling(name, sig, gin, k, dil):
    g = load_golden(name)
    hp = O.HParams(hidden_channels=16, kernel_size_dec=k, dilation_rate=dil, n_block_layers=3, gin_channels=gin,
                   sigmoid_scale=sig)
    sd = {"f." + kk: v for kk, v in split_prefix(g, "sd.", requires_grad=True).items()}
    x = T(g["x"]).requires_grad_(True)
    mask = T(g["mask"])
    gc = T(g["g"]).requires_grad_(True) if gin else None
    z, logdet = O.coupling(sd, "f", x, mask, gc, hp, 16)
    assert_close(z, g["z"], what="z", **TOL)
    assert_close(logdet, g["logdet"], what="logdet", **TOL)
    ((z * T(g["r"])).sum() + (logdet * T(g["s"])).sum()).backward()
    assert_close(x.grad, g["dx"], what="dx", rtol=1e-4, atol=2e-5)
    if gin:
        assert_close(gc.grad, g["dg"], what="dg", rtol=1e-4, atol=2e-5)
    _grad_check({kk[2:]: v for kk, v in sd.items()}, g, name, atol_of_max=2e-4 if g["x"].shape[2] > 256 else 0.0)
    with torch.no_grad():
        xr, _ = O.coupling(sd, "f", z.detach(), mask, gc, hp, 16, reverse=True)
    assert_close(xr, g["x_rev"], what="x_rev", rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("gin", [0, 8])
def test_wn(gin):
    g = load_golden(f"wn_h16_gin{gin}")
    sd = {"w." + kk: v for kk, v in split_prefix(g, "sd.", requires_grad=True).items()}
    x = T(g["x"]).requires_grad_(True)
    gc = T(g["g"]).requires_grad_(True) if gin else None
    out = O.wn(sd, "w", x, T(g["mask"]), gc, 16, 3, 1)
    assert_close(out, g["out"], what="out", **TOL)
    (out * T(g["r"])).sum().backward()
    assert_close(x.grad, g["dx"], what="dx", rtol=1e-4, atol=2e-5)
    _grad_check({kk[2:]: v for kk, v in sd.items()}, g, "wn")


def test_gate_and_squeeze():
    g = load_golden("gate_h16")
    a, b = T(g["a"]).requires_grad_(True), T(g["b"]).requires_grad_(True)
    acts = O.gate(a, b, 16)
    assert_close(acts, g["acts"], what="acts", **TOL)
    (acts * T(g["r"])).sum().backward()
    assert_close(a.grad, g["da"], what="da", **TOL)
    assert_close(b.grad, g["db"], what="db", **TOL)

    g = load_golden("squeeze_c6_t11")
    xs, ms = O.squeeze(T(g["x"]), T(g["mask"]), 2)
    assert_close(xs, g["x_sqz"], what="x_sqz", rtol=0, atol=0)
    assert_close(ms, g["mask_sqz"], what="mask_sqz", rtol=0, atol=0)
    xu, mu = O.unsqueeze(xs, ms, 2)
    assert_close(xu, g["x_unsqz"], what="x_unsqz", rtol=0, atol=0)
    assert_close(mu, g["mask_unsqz"], what="mask_unsqz", rtol=0, atol=0)


# ------------------------------------------------------------------------------------------------ attention
MHA_CASES = ["mha_t12_w4", "mha_t4_w4", "mha_t5_w4", "mha_t12_w4_blk3", "mha_t12_nowin", "mha_t70_w4", "mha_c32_t70_w4",
             "mha_c32_t12_w4_blk3", "mha_c32_t5_w4", "mha_c32_t40_nowin", "mha_c192_t160_w4",
             "mha_c192_t240_w4", "mha_c64_t256_w4", "mha_c192_t300_w4", "mha_c64_t512_w4"]


@pytest.mark.parametrize("name", MHA_CASES)
def test_attention(name):
    g = load_golden(name)
    win = int(g["window"])
    blk = int(g["block"])
    hp = O.HParams(hidden_channels=g["x"].shape[1], n_heads=2, window_size=None if win < 0 else win,
                   block_length=None if blk < 0 else blk)
    sd = {"a." + kk: v for kk, v in split_prefix(g, "sd.", requires_grad=True).items()}
    x = T(g["x"]).requires_grad_(True)
    mask = T(g["mask"])
    attn_mask = mask.unsqueeze(2) * mask.unsqueeze(-1)
    y, p = O.multi_head_attention(sd, "a", x, attn_mask, hp)
    assert_close(y, g["y"], what="y", **TOL)
    assert_close(p, g["p_attn"], what="p_attn", **TOL)
    (y * T(g["r"])).sum().backward()
    assert_close(x.grad, g["dx"], what="dx", rtol=1e-4, atol=2e-5)
    _grad_check({kk[2:]: v for kk, v in sd.items()}, g, name, atol_of_max=2e-4 if g["x"].shape[2] > 256 else 0.0)


# ------------------------------------------------------------------------------------------------ losses
def test_losses():
    g = load_golden("losses")
    z, m, logs = (T(g[k]).requires_grad_(True) for k in ("z", "m", "logs"))
    logdet = T(g["logdet"]).requires_grad_(True)
    loss = O.mle_loss(z, m, logs, logdet, T(g["mask"]))
    assert_close(loss, g["loss"], what="mle", **TOL)
    loss.backward()
    for t, k in ((z, "dz"), (m, "dm"), (logs, "dlogs"), (logdet, "dlogdet")):
        assert_close(t.grad, g[k], what=k, **TOL)
    logw = T(g["logw"]).requires_grad_(True)
    dl = O.duration_loss(logw, T(g["logw_"]), T(g["lengths"]))
    assert_close(dl, g["dur_loss"], what="dur", **TOL)
    dl.backward()
    assert_close(logw.grad, g["dlogw"], what="dlogw", **TOL)


# ------------------------------------------------------------------------------------------------ end to end
def _e2e_hp(tag):
    kw = dict(n_vocab=148, hidden_channels=32, filter_channels=64, filter_channels_dp=32, out_channels=80,
              kernel_size=3, n_heads=2, n_layers_enc=2, n_blocks_dec=2, kernel_size_dec=5, dilation_rate=1,
              n_block_layers=2, n_split=4, n_sqz=2, window_size=4, mean_only=True, prenet=True)
    if tag == "spk":
        kw.update(gin_channels=8, n_speakers=3, mean_only=False, sigmoid_scale=True)
    return O.HParams(**kw)


@pytest.mark.parametrize("tag", ["base", "spk"])
def test_e2e_train(tag):
    g = load_golden(f"e2e_{tag}_train")
    hp = _e2e_hp(tag)
    sd = split_prefix(g, "sd.", requires_grad=True)
    spk = T(g["speaker_ids"]) if "speaker_ids" in g else None
    x, xl, y, yl = T(g["x"]), T(g["x_lengths"]), T(g["y"]), T(g["y_lengths"])
    (z, z_m, z_logs, logdet, z_mask), (x_m, x_logs, x_mask), (attn, logw, logw_) = O.generator_forward(
        sd, hp, x, xl, y, yl, spk)
    for name, t in dict(z=z, z_m=z_m, z_logs=z_logs, logdet=logdet, z_mask=z_mask, x_m=x_m, x_logs=x_logs,
                        x_mask=x_mask, logw=logw, logw_=logw_).items():
        assert_close(t, g[name], what=name, rtol=1e-4, atol=5e-5)
    assert (attn.numpy().astype(np.int8) == g["attn"]).all(), "MAS alignment differs"
    l_mle = O.mle_loss(z, z_m, z_logs, logdet, z_mask)
    l_len = O.duration_loss(logw, logw_, xl)
    assert_close(l_mle, g["l_mle"], what="l_mle", **TOL)
    assert_close(l_len, g["l_length"], what="l_length", **TOL)
    (l_mle + l_len).backward()
    gg = split_prefix(g, "grad.")
    for k, want in gg.items():
        assert sd[k].grad is not None, k
        assert_close(sd[k].grad, want, what="grad " + k, rtol=5e-4, atol=5e-5)
    # clip + three Adam/Noam updates with these grads
    grads = {k: sd[k].grad.clone() for k in gg}
    tn = O.clip_grad_value(grads.values(), 5.0)
    assert abs(tn - float(g["total_norm"])) <= 1e-4 * float(g["total_norm"])
    params = {k: v.detach().clone() for k, v in sd.items()}
    opt = O.AdamNoam(params, dim_model=32, warmup_steps=4000, lr=1.0, betas=(0.9, 0.98), eps=1e-9)
    lrs = [opt.cur_lr]
    for _ in range(3):
        opt.step(grads)
        lrs.append(opt.cur_lr)
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-12)
    after = split_prefix(g, "sd_after3.")
    for k, want in after.items():
        assert_close(params[k], want, what="after3 " + k, rtol=1e-4, atol=1e-6)


def keep_masks_from_golden(g):
    """KeepMasks from a fixture's `keep.<site>` / `p.<site>` arrays (oracle/make_golden.py gen_e2e_dropout)."""
    return O.KeepMasks({k[5:]: (T(v), float(g["p." + k[5:]])) for k, v in g.items() if k.startswith("keep.")})


def test_e2e_train_with_recorded_dropout():
    """The training step WITH dropout (the state bench.py times): the reference's own run with its F.dropout decisions
    recorded; the oracle fed the same decisions reproduces all 11 outputs, both losses and every parameter gradient —
    pins every dropout SITE of the oracle (pre-net 0.5, encoder 0.1 x 4 per layer, duration predictor, WN 0.05)."""
    g = load_golden("e2e_dropout_train")
    hp = _e2e_hp("base")
    sd = split_prefix(g, "sd.", requires_grad=True)
    drop = keep_masks_from_golden(g)
    assert len(drop.masks) == 3 + 2 * 4 + 2 + 2 * 2
    x, xl, y, yl = T(g["x"]), T(g["x_lengths"]), T(g["y"]), T(g["y_lengths"])
    (z, z_m, z_logs, logdet, z_mask), (x_m, x_logs, x_mask), (attn, logw, logw_) = O.generator_forward(
        sd, hp, x, xl, y, yl, None, drop=drop)
    assert drop.used == set(drop.masks), sorted(set(drop.masks) - drop.used)
    for name, t in dict(z=z, z_m=z_m, z_logs=z_logs, logdet=logdet, z_mask=z_mask, x_m=x_m, x_logs=x_logs,
                        x_mask=x_mask, logw=logw, logw_=logw_).items():
        assert_close(t, g[name], what=name, rtol=1e-4, atol=5e-5)
    assert (attn.numpy().astype(np.int8) == g["attn"]).all(), "MAS alignment differs"
    l_mle = O.mle_loss(z, z_m, z_logs, logdet, z_mask)
    l_len = O.duration_loss(logw, logw_, xl)
    assert_close(l_mle, g["l_mle"], what="l_mle", **TOL)
    assert_close(l_len, g["l_length"], what="l_length", **TOL)
    (l_mle + l_len).backward()
    for k, want in split_prefix(g, "grad.").items():
        assert sd[k].grad is not None, k
        assert_close(sd[k].grad, want, what="grad " + k, rtol=5e-4, atol=5e-5)
    # and the masks matter: without them the outputs differ by far more than the tolerance
    with torch.no_grad():
        z0 = O.generator_forward(sd, hp, x, xl, y, yl, None)[0][0]
    assert float((z0 - T(g["z"])).abs().max()) > 1e-2


@pytest.mark.parametrize("tag", ["base", "spk"])
def test_e2e_generate(tag):
    gt = load_golden(f"e2e_{tag}_train")
    g = load_golden(f"e2e_{tag}_gen")
    hp = _e2e_hp(tag)
    sd = split_prefix(gt, "sd.")
    spk = T(g["speaker_ids"]) if "speaker_ids" in g else None
    with torch.no_grad():
        (y, z_m, z_logs, ld, z_mask), _, (attn, logw, logw_) = O.generator_generate(
            sd, hp, T(g["x"]), T(g["x_lengths"]), spk, T(g["noise"]), float(g["noise_scale"]), 1.0)
    assert ld is None
    assert (attn.numpy().astype(np.int8) == g["attn"]).all()
    for name, t in dict(y=y, z_m=z_m, z_logs=z_logs, z_mask=z_mask, logw=logw, logw_=logw_).items():
        assert_close(t, g[name], what=name, rtol=1e-4, atol=5e-5)


def test_decoder_invertible_and_logdet_is_jacobian():
    """Property tests the maths implies: reverse(forward(x)) == x, and logdet == log|det J| on a tiny shape."""
    hp = O.HParams(hidden_channels=8, out_channels=2, n_blocks_dec=2, n_block_layers=2, n_split=2, n_sqz=2,
                   n_layers_enc=1, filter_channels=8, filter_channels_dp=8)
    sd = O.init_state_dict(hp, seed=3)
    for k in sd:
        if k.endswith(".end.weight"):
            sd[k] = 0.3 * torch.randn_like(sd[k])
    t = 4
    x = torch.randn(1, 2, t, dtype=torch.float32)
    mask = torch.ones(1, 1, t)
    z, logdet = O.flow_decoder(sd, x, mask, None, hp)
    xr, _ = O.flow_decoder(sd, z, mask, None, hp, reverse=True)
    assert_close(xr, x, what="roundtrip", rtol=1e-4, atol=1e-5)
    sd64 = {k: v.double() for k, v in sd.items()}
    J = torch.autograd.functional.jacobian(
        lambda xx: O.flow_decoder(sd64, xx.view(1, 2, t), mask.double(), None, hp)[0].reshape(-1), x.double().reshape(-1))
    assert_close(logdet[0], torch.linalg.slogdet(J)[1], what="logdet", rtol=1e-4, atol=1e-4)
