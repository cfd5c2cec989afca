"""The opt-in bf16-plane arithmetic of the WN convolutions (csrc/convgemm_split.hip, glowtts_conv_math).

An fp32 value is the exact sum of three bf16 values; "bf16x6" forms the six products above 2^-24 on the bf16 matrix pipe
with fp32 accumulation, so its results must be as close to an fp64 reference as the native fp32 MFMA kernels are.
"bf16x3" (2^-16 products) and "bf16" are looser modes with their own bounds.  Native fp32 stays the default and the
parity reference; these tests pin what each mode promises.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import assert_close, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture()
def M():
    """Package handles + a guard that restores the package's default arithmetic after every test."""
    from glow_tts_train import _hip, convops, layers

    _hip.load()

    class NS:
        pass

    ns = NS()
    ns.hip, ns.convops, ns.layers = _hip, convops, layers
    before = convops.conv_math_name()
    yield ns
    _hip.conv_bind_planes(None)
    _PLANES.clear()
    convops.set_conv_math(before)


_PLANES = {}


def _split(M, *weights):
    """Split a packed buffer into caller-owned planes and bind them to this thread (the tests use one buffer at a time)."""
    for w in weights:
        planes = _PLANES.setdefault(w.data_ptr(), torch.empty(3 * w.numel(), device=w.device, dtype=torch.int16))
        M.hip.call("glowtts_conv_split_weights", M.hip.ptr(w), w.numel(), M.hip.ptr(planes))
        M.hip.conv_bind_planes(w, planes)


def _errors(fn, ref, M, weights, modes):
    out = {}
    scale = float(ref.abs().max())
    for mode in modes:
        M.convops.set_conv_math(mode)
        _split(M, *weights)
        got = fn().double()
        out[mode] = float((got - ref).abs().max()) / scale
    M.convops.set_conv_math("fp32")
    return out


MODES = ("fp32", "bf16x6+wrw", "bf16x3+wrw", "bf16+wrw")


def _check(err, what, must_split=True):
    native, x6, x3, b16 = (err[m] for m in MODES)
    assert native < 2e-5, (what, err)
    assert x6 <= 1.5 * native + 1e-7, (what, err)              # fp32-equivalent: no worse than the native fp32 MFMA path
    assert x6 <= x3 < 2e-4, (what, err)                        # products good to 2^-16
    assert x3 <= b16 < 3e-2, (what, err)
    if must_split:                                             # the split kernels really ran (production width)
        assert b16 > 20 * native, (what, err)


@pytest.mark.parametrize("b,h,t", [(4, 192, 400), (3, 192, 160), (2, 96, 96)])
def test_every_mode_against_fp64(M, b, h, t):
    """Gated in-conv (k = 5), res/skip 1x1, backward-data (k = 5) and both weight gradients at WN shapes."""
    call, ptr = M.hip.call, M.hip.ptr
    dev = "cuda"
    torch.manual_seed(b * 1000 + h + t)
    x = torch.randn(b, h, t, device=dev)
    lens = torch.randint(t // 2, t + 1, (b,))
    lens[0] = t
    m2 = (torch.arange(t)[None] < lens[:, None]).float().to(dev)
    v_in = torch.randn(2 * h, h, 5, device=dev) * 0.03
    b_in = torch.randn(2 * h, device=dev) * 0.1
    v_rs = torch.randn(2 * h, h, 1, device=dev) * 0.07
    wf_in, wb_in, _ = M.convops.pack_weight(v_in, None)
    wf_rs, _, _ = M.convops.pack_weight(v_rs, None)
    acts_in = torch.randn(b, h, t, device=dev) * 0.5
    skip_in = torch.randn(b, h, t, device=dev)
    d2 = torch.randn(b, 2 * h, t, device=dev)

    def gate():
        acts = torch.empty(b, h, t, device=dev)
        ts = torch.empty(b, 2 * h, t, device=dev)
        call("glowtts_conv_gate_fwd", ptr(x), ptr(wf_in), ptr(b_in), None, None, 1.0, ptr(acts), ptr(ts), b, h, t, 5, 1, 2)
        return torch.cat([acts, ts], 1)

    pre = F.conv1d(x.double(), v_in.double(), b_in.double(), padding=2)
    th, sg = torch.tanh(pre[:, :h]), torch.sigmoid(pre[:, h:])
    _check(_errors(gate, torch.cat([th * sg, th, sg], 1), M, (wf_in,), MODES), "gate", h == 192)

    def resskip():
        xo = torch.empty(b, h, t, device=dev)
        sk = torch.empty(b, h, t, device=dev)
        call("glowtts_conv_res_skip_fwd", ptr(acts_in), ptr(wf_rs), ptr(b_in), ptr(m2), ptr(x), ptr(skip_in), ptr(xo), ptr(sk),
             b, h, t, 0)
        return torch.cat([xo, sk], 1)

    rs = F.conv1d(acts_in.double(), v_rs.double(), b_in.double())
    ref = torch.cat([(x.double() + rs[:, :h]) * m2[:, None].double(), skip_in.double() + rs[:, h:]], 1)
    _check(_errors(resskip, ref, M, (wf_rs,), MODES), "res/skip", h == 192)

    def bwd_data():
        dx = torch.empty(b, h, t, device=dev)
        M.convops.conv_fwd(d2, wb_in, None, None, dx, 2 * h, h, 5, 1, 2, addend=skip_in)
        return dx

    ref = F.conv_transpose1d(d2.double(), v_in.double(), padding=2) + skip_in.double()
    _check(_errors(bwd_data, ref, M, (wb_in,), MODES), "backward-data", h == 192)

    def wrw5():
        dwp = torch.zeros(5, h, 2 * h, device=dev)
        db = torch.zeros(2 * h, device=dev)
        call("glowtts_conv_wrw", ptr(x), x.stride(0), ptr(d2), d2.stride(0), ptr(m2), None, ptr(dwp), ptr(db), b, h, 2 * h, t,
             5, 1, 2)
        return torch.cat([dwp.reshape(-1), db])

    dm = d2.double() * m2[:, None].double()
    dw = torch.nn.grad.conv1d_weight(x.double(), (2 * h, h, 5), dm, padding=2)
    _check(_errors(wrw5, torch.cat([dw.permute(2, 1, 0).reshape(-1), dm.sum((0, 2))]), M, (), MODES), "weight grad k=5")

    def wrw1():
        dwp = torch.zeros(1, h, 2 * h, device=dev)
        call("glowtts_conv_wrw", ptr(acts_in), acts_in.stride(0), ptr(d2), d2.stride(0), None, None, ptr(dwp), None, b, h, 2 * h,
             t, 1, 1, 0)
        return dwp.reshape(-1)

    _check(_errors(wrw1, torch.einsum("bot,bct->co", d2.double(), acts_in.double()).reshape(-1), M, (), MODES), "weight grad 1x1")


@pytest.mark.parametrize("b,h,m,t", [(32, 192, 192, 128), (32, 192, 384, 192), (40, 192, 192, 96), (32, 192, 384, 160)])
def test_unmasked_1x1_weight_gradient_with_several_chunks_per_workgroup(M, b, h, m, t):
    """The software-pipelined 1x1 weight gradient (convwrw_split_kernel<., 1, 4 | 5, 4>, AG form) when a workgroup holds
    MORE THAN ONE chunk: the split units of chunk c + 1 ride behind the MFMA groups of chunk c, and the 64-frame form has as
    many units as groups — its last unit (rows 48..63 of the d tile, the last bias partial) once ran in the prologue only, so
    every later chunk multiplied chunk 0's d planes.  T = 128 / 192 / 96 take the 64-frame form
    (nb = ceil(B * ceil(T / 64) / (512 / tiles)) >= 2 here), T = 160 the 80-frame one; dW and dbias against fp64."""
    call, ptr = M.hip.call, M.hip.ptr
    dev = "cuda"
    torch.manual_seed(b + h + m + t)
    x = torch.randn(b, h, t, device=dev) * 0.5
    d = torch.randn(b, m, t, device=dev)
    tiles = -(-h // 64) * -(-m // 64)
    ct = 80 if (t % 80 == 0 or -(-t // 80) * 80 <= -(-t // 64) * 64) else 64
    assert -(-(b * -(-t // ct)) // (512 // tiles)) >= 2, "the case must give a workgroup several chunks"

    def wrw1():
        dwp = torch.zeros(1, h, m, device=dev)
        db = torch.zeros(m, device=dev)
        call("glowtts_conv_wrw", ptr(x), x.stride(0), ptr(d), d.stride(0), None, None, ptr(dwp), ptr(db), b, h, m, t, 1, 1, 0)
        return torch.cat([dwp.reshape(-1), db])

    ref = torch.cat([torch.einsum("bot,bct->co", d.double(), x.double()).reshape(-1), d.double().sum((0, 2))])
    err = _errors(wrw1, ref, M, (), ("fp32", "bf16x6+wrw"))
    assert err["fp32"] < 2e-5 and err["bf16x6+wrw"] <= 1.5 * err["fp32"] + 1e-7, err


def test_planes_are_a_snapshot_and_only_bound_weights_switch(M):
    """The planes are a snapshot of the packed buffer they were made from (WNPackPlan re-splits after every pack), and only
    the buffer bound to the calling thread runs in the selected mode: everything else stays native, bit for bit."""
    call, ptr = M.hip.call, M.hip.ptr
    b, h, t = 2, 192, 160
    torch.manual_seed(5)
    x = torch.randn(b, h, t, device="cuda")
    v5 = torch.randn(2 * h, h, 5, device="cuda") * 0.03
    wf, _, _ = M.convops.pack_weight(v5, None)
    other, _, _ = M.convops.pack_weight(v5 * 1.5, None)
    acts = torch.empty(b, h, t, device="cuda")

    def gate(w):
        call("glowtts_conv_gate_fwd", ptr(x), ptr(w), None, None, None, 1.0, ptr(acts), None, b, h, t, 5, 1, 2)
        return acts.clone()

    native, native_other = gate(wf), gate(other)
    M.convops.set_conv_math("bf16")
    assert torch.equal(gate(wf), native)                       # mode on, nothing bound: native kernel
    _split(M, wf)
    coarse = gate(wf)
    assert not torch.equal(coarse, native) and rel_err(coarse, native) < 3e-2
    assert torch.equal(gate(other), native_other)              # a buffer that is not the bound one: native
    wf.mul_(2.0)                                               # weights change, planes do not: the old result persists
    assert torch.equal(gate(wf), coarse)
    _split(M, wf)
    assert rel_err(gate(wf), coarse) > 0.1
    M.hip.conv_bind_planes(None)
    M.convops.set_conv_math("fp32")
    doubled = gate(wf)
    M.convops.set_conv_math("bf16")
    assert torch.equal(gate(wf), doubled)                      # unbound: native again
    # a binding made under one mode is not used under another (the planes were written for the mode in force then)
    _split(M, wf)
    M.convops.set_conv_math("bf16x6")
    assert torch.equal(gate(wf), doubled)


@pytest.mark.parametrize("mode,tol", [("bf16x6", 1.0), ("bf16x6+wrw", 1.0), ("bf16x3+wrw", 20.0)])
def test_wn_stack_in_split_mode_matches_oracle(M, mode, tol):
    """The whole WN stack (native forward executor, layer-by-layer backward) in split arithmetic against the CPU oracle:
    bf16x6 within the tolerances of the native path (tests/test_hip_parity.py::test_wn_stack_shapes_vs_oracle)."""
    from oracle import glow_oracle as O

    b, h, t, k, nl = 2, 192, 160, 5, 4
    torch.manual_seed(77)
    wn = M.layers.WN(2 * h, h, kernel_size=k, dilation_rate=1, n_layers=nl, p_dropout=0.0).cuda().train()
    x0 = torch.randn(b, h, t)
    lens = torch.tensor([t, t - 37])
    mask = (torch.arange(t)[None] < lens[:, None]).float()[:, None]
    r = torch.randn(b, h, t)
    sd = {"wn." + k_: v.detach().cpu().clone().requires_grad_(True) for k_, v in wn.state_dict().items()}
    xo = (x0 * mask).clone().requires_grad_(True)
    yo = O.wn(sd, "wn", xo, mask, None, h, nl, 1)
    (yo * r).sum().backward()

    M.convops.set_conv_math(mode)
    for p in wn.parameters():
        p.grad = torch.zeros_like(p)
    x = (x0 * mask).cuda().requires_grad_(True)
    y = wn(x, mask.cuda())
    (y * r.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert_close(y, yo, what="y", rtol=2e-4 * tol, atol=2e-5 * tol * max(1.0, float(yo.abs().max())))
    assert_close(x.grad, xo.grad, what="dx", rtol=5e-4 * tol, atol=5e-5 * tol * max(1.0, float(xo.grad.abs().max())))
    for name, p in wn.named_parameters():
        want = sd["wn." + name].grad
        assert_close(p.grad, want, what=f"grad {name}", rtol=1e-3 * tol, atol=1e-4 * tol * max(1.0, float(want.abs().max())))


def test_training_step_bf16x6_tracks_native(M):
    """Three optimisation steps of a production-width model, native fp32 against bf16x6, from the same start: losses agree
    to fp32 noise and the parameters stay together (Adam's sign-like update only diverges where gradients are noise)."""
    from glow_tts_train import models, optimize
    from glow_tts_train.train import train_batch

    def run(mode):
        M.convops.set_conv_math(mode)
        torch.manual_seed(1234)
        model = models.FlowGenerator(n_vocab=148, hidden_channels=192, filter_channels=768, filter_channels_dp=256,
                                     out_channels=80, kernel_size=3, n_heads=2, n_layers_enc=2, p_dropout=0.0, n_blocks_dec=3,
                                     kernel_size_dec=5, dilation_rate=1, n_block_layers=4, p_dropout_dec=0.0, n_split=4,
                                     n_sqz=2, window_size=4, mean_only=True, prenet=True).cuda().train()
        for mod in model.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        with torch.no_grad():
            for f in model.decoder.flows:
                if hasattr(f, "end"):
                    f.end.weight.normal_(0, 0.01)
        opt = optimize.Adam(model.parameters(), scheduler="noam", dim_model=192, warmup_steps=4000, lr=1.0)
        gen = torch.Generator().manual_seed(9)
        bsz, tx, ty = 4, 40, 200
        xl = torch.tensor([40, 33, 30, 21]).cuda()
        yl = torch.tensor([200, 170, 150, 110]).cuda()
        x = (torch.randint(1, 148, (bsz, tx), generator=gen) * (torch.arange(tx)[None] < xl.cpu()[:, None])).cuda()
        y = (torch.randn(bsz, 80, ty, generator=gen) * (torch.arange(ty)[None, None] < yl.cpu()[:, None, None])).cuda()
        losses = [float(train_batch(model, opt, (x, xl, y, yl, None), 5.0)) for _ in range(3)]
        return losses, opt._optim.flat_p.clone()

    l0, p0 = run("fp32")
    l1, p1 = run("bf16x6+wrw")
    np.testing.assert_allclose(l1, l0, rtol=2e-5)
    close = ((p1 - p0).abs() <= 1e-5 + 1e-3 * p0.abs()).float().mean().item()
    assert close > 0.99, close


@pytest.mark.parametrize("b,t_mel,blocks", [(64, 1000, 12), (8, 400, 6)])
def test_bf16_mode_logdet_tolerance(M, b, t_mel, blocks):
    """BASELINE configs[2] ("bf16 ... log-det tolerance check") at its full size, and config 1's: the flow decoder with the
    WN convolutions on plain bf16 operands against the native fp32 path.  The invertible kernels (ActNorm, InvConvNear,
    affine apply) stay fp32 in every mode, so the log-det only moves through the couplings' predicted log-scales."""
    from glow_tts_train import models

    torch.manual_seed(3)
    dec = models.FlowSpecDecoder(80, 192, 5, 1, blocks, 4, p_dropout=0.0, n_split=4, n_sqz=2, sigmoid_scale=False,
                                 gin_channels=0).cuda().train()
    with torch.no_grad():
        for f in dec.flows:
            if hasattr(f, "end"):
                f.end.weight.normal_(0, 0.01)
            if hasattr(f, "logs") and hasattr(f, "bias"):
                f.logs.normal_(0, 0.1)
                f.bias.normal_(0, 0.1)
    yl = torch.linspace(t_mel, t_mel // 2, b).long()
    y = (torch.randn(b, 80, t_mel) * (torch.arange(t_mel)[None, None] < yl[:, None, None])).cuda()
    mask = (torch.arange(t_mel)[None, None] < ((yl // 2) * 2)[:, None, None]).float().cuda()

    def run(mode):
        M.convops.set_conv_math(mode)
        with torch.no_grad():
            return dec(y, mask)

    z0, ld0 = run("fp32")
    z6, ld6 = run("bf16x6")
    z1, ld1 = run("bf16")
    assert rel_err(z6, z0) < 1e-5 and rel_err(ld6, ld0) < 1e-6
    assert 1e-5 < rel_err(z1, z0) < 5e-2                            # really a different arithmetic, and a usable one
    assert rel_err(ld1, ld0) < 2e-3                                 # the log-det tolerance of the bf16 configuration


@pytest.mark.parametrize("b,h,t", [(4, 192, 400), (3, 192, 160), (2, 96, 96)])
def test_weight_gradient_from_presplit_planes(M, b, h, t):
    """glowtts_split_planes + glowtts_conv_wrw_planes: the weight gradient from operands split ONCE into three bf16 planes
    (groundwork for letting the producing kernels write the planes; not yet on the training path) is fp32-equivalent."""
    call, ptr = M.hip.call, M.hip.ptr
    dev = "cuda"
    torch.manual_seed(b + h + t)
    x = torch.randn(b, h, t, device=dev)
    d = torch.randn(b, 2 * h, t, device=dev)
    xp = torch.empty(3 * x.numel(), device=dev, dtype=torch.int16)
    dp = torch.empty(3 * d.numel(), device=dev, dtype=torch.int16)
    call("glowtts_split_planes", ptr(x), x.numel(), ptr(xp), 3)
    call("glowtts_split_planes", ptr(d), d.numel(), ptr(dp), 3)
    # the three planes sum back to the value exactly
    planes = xp.view(3, -1).to(torch.int32).bitwise_and(0xFFFF).bitwise_left_shift(16).view(torch.float32)
    assert torch.equal(planes[0] + planes[1] + planes[2], x.reshape(-1))
    for taps in (5, 1):
        dwp = torch.zeros(taps, h, 2 * h, device=dev)
        db = torch.zeros(2 * h, device=dev)
        call("glowtts_conv_wrw_planes", ptr(xp), x.numel(), h * t, ptr(dp), d.numel(), 2 * h * t, ptr(dwp), ptr(db), b, h, 2 * h, t,
             taps, 3)
        native = torch.zeros(taps, h, 2 * h, device=dev)
        call("glowtts_conv_wrw", ptr(x), x.stride(0), ptr(d), d.stride(0), None, None, ptr(native), None, b, h, 2 * h, t, taps, 1,
             (taps - 1) // 2)
        dw = torch.nn.grad.conv1d_weight(x.double(), (2 * h, h, taps), d.double(), padding=(taps - 1) // 2).permute(2, 1, 0)
        scale = float(dw.abs().max())
        e_planes = float((dwp.double() - dw).abs().max()) / scale
        e_native = float((native.double() - dw).abs().max()) / scale
        assert e_planes <= 1.5 * e_native + 1e-7, (taps, e_planes, e_native)
        assert rel_err(db, d.double().sum((0, 2))) < 1e-6
    with pytest.raises(RuntimeError, match="no kernel"):
        call("glowtts_conv_wrw_planes", ptr(xp), x.numel(), h * t, ptr(dp), d.numel(), 2 * h * t, ptr(dwp), None, b, h, 2 * h, t, 3, 3)


@pytest.mark.parametrize("b,t", [(8, 160), (32, 160), (48, 240), (64, 200)])
def test_encoder_ffn_convs_take_the_group_planes(M, b, t):
    """(b, t): the text encoder's shapes at BASELINE configs[1] (32 x 160), configs[4] (48 x 240: where the whole-step margin of one
    FFN weight gradient sits 2.8x closer to its tolerance in bf16x6 than in fp32 — profiles/r04_parity_margins.json) and
    configs[2] (64 x 200).
    Round 3: the text encoder's 3-tap FFN convolutions (attentions.py:347-381; 192 -> 768 -> 192 channels at T_text frames)
    run in the selected conv arithmetic when their ConvGroup keeps bf16 planes: forward, backward-data (32-frame tiles at
    T = 160) and the 3-tap weight gradient (convwrw_tr.hip) against torch fp64 — bf16x6 no worse than the native kernels, and
    bit-different from them (the plane kernels really ran)."""
    import torch.nn as nn

    torch.manual_seed(11)
    h, f = 192, 768
    c1, c2 = nn.Conv1d(h, f, 3, padding=1).cuda(), nn.Conv1d(f, h, 3, padding=1).cuda()
    for p_ in list(c1.parameters()) + list(c2.parameters()):
        p_.grad = torch.zeros_like(p_)
    grp = M.convops.ConvGroup([c1, c2], planes=True)
    x0 = torch.randn(b, h, t, device="cuda")
    r = torch.randn(b, h, t, device="cuda")
    m2 = torch.ones(b, t, device="cuda")

    xd = x0.double().requires_grad_(True)
    w1, w2 = c1.weight.detach().double().requires_grad_(True), c2.weight.detach().double().requires_grad_(True)
    yd = F.conv1d(F.conv1d(xd, w1, c1.bias.detach().double(), padding=1), w2, c2.bias.detach().double(), padding=1)
    (yd * r.double()).sum().backward()
    ref = {"y": yd.detach(), "dx": xd.grad, "dw1": w1.grad, "dw2": w2.grad}

    def run(mode):
        M.convops.set_conv_math(mode)
        for p_ in list(c1.parameters()) + list(c2.parameters()):
            p_.grad.zero_()
        x = x0.clone().requires_grad_(True)
        grp.begin()
        y = M.convops.conv1d(c2, M.convops.conv1d(c1, x, m2), m2)
        (y * r).sum().backward()
        M.convops.flush_groups()
        torch.cuda.synchronize()
        got = {"y": y.detach().clone(), "dx": x.grad.clone(), "dw1": c1.weight.grad.clone(), "dw2": c2.weight.grad.clone()}
        return {k: float((got[k].double() - ref[k]).abs().max() / ref[k].abs().max()) for k in ref}, got

    (native, t_native), (x6, t_x6) = run("fp32"), run("bf16x6+wrw")
    for k in ref:
        assert native[k] < 2e-5, (k, native)
        assert x6[k] <= 1.5 * native[k] + 2e-7, (k, x6, native)
        # the plane kernels really ran: the same numbers to fp32 accuracy, not the same bits (another summation order)
        assert not torch.equal(t_x6[k], t_native[k]), f"{k}: bit-identical to the native kernels"


@pytest.mark.parametrize("mode", ["bf16x6+wrw", "fp32"])
@pytest.mark.parametrize("b,t", [(3, 100), (2, 400), (4, 36), (2, 37)])       # (T % 4 != 0: the entry launches the problems one by one)
def test_multi_problem_1x1_weight_gradient_against_fp64(M, mode, b, t):
    """glowtts_conv_wrw1_multi (csrc/convwrw1.hip): the 1x1 weight gradients of a flow block — and problems of other shapes —
    in ONE launch (192 x 192 tiles, 32-frame steps) against an fp64 contraction: two-source output gradients, a slice of a wider
    tensor as x, masks on either operand, more than 192 channels either side, tiny problems; in native fp32 arithmetic the entry
    launches the problems one by one (same results).  Accumulates into dwp / dbias."""
    import ctypes

    torch.manual_seed(b * 1000 + t)
    dev = "cuda"
    M.convops.set_conv_math(mode)
    lens = torch.randint(t // 2, t + 1, (b,))
    lens[0] = t
    mask = (torch.arange(t)[None] < lens[:, None]).float().to(dev)
    wide = torch.randn(b, 160, t, device=dev)
    # (Cin, M, two-source split or 0, mask on d, mask on x, x tensor or None)
    specs = [(192, 384, 192, False, False, None), (192, 192, 0, False, False, None), (80, 192, 0, True, False, wide),
             (192, 160, 0, False, False, None), (48, 16, 0, False, True, None), (200, 200, 0, True, True, None),
             (192, 384, 192, False, False, None), (64, 320, 64, False, False, None)]
    if t % 4:                                          # (the two-source form needs 16-byte rows, as glowtts_conv_wrw2 does)
        specs = [sp for sp in specs if not sp[2]]
    probs = (M.hip.Wrw1Problem * len(specs))()
    keep, want = [], []
    for j, (cin, m, split, md, mx, xt) in enumerate(specs):
        x = xt if xt is not None else torch.randn(b, cin, t, device=dev)
        d = torch.randn(b, split if split else m, t, device=dev)
        d2 = torch.randn(b, m - split, t, device=dev) if split else None
        dwp = 0.5 * torch.randn(cin, m, device=dev)
        dbias = 0.5 * torch.randn(m, device=dev)
        xe = x[:, :cin].double() * (mask[:, None].double() if mx else 1.0)
        de = (torch.cat([d, d2], 1) if split else d).double() * (mask[:, None].double() if md else 1.0)
        want.append((dwp.double() + torch.einsum("bkt,bmt->km", xe, de), dbias.double() + de.sum((0, 2))))
        q = probs[j]
        q.x, q.d, q.d2 = x.data_ptr(), d.data_ptr(), (d2.data_ptr() if split else None)
        q.mask_d, q.mask_x = (mask.data_ptr() if md else None), (mask.data_ptr() if mx else None)
        q.dwp, q.dbias = dwp.data_ptr(), dbias.data_ptr()
        q.x_bs, q.d_bs, q.d2_bs = x.shape[1] * t, d.shape[1] * t, ((m - split) * t if split else 0)
        q.Cin, q.M, q.d_split = cin, m, split
        keep.append((x, d, d2, dwp, dbias))
    M.hip.call("glowtts_conv_wrw1_multi", len(specs), ctypes.addressof(probs), b, t)
    torch.cuda.synchronize()
    for j, ((x, d, d2, dwp, dbias), (w_ref, b_ref)) in enumerate(zip(keep, want)):
        scale = float(w_ref.abs().max())
        err = float((dwp.double() - w_ref).abs().max()) / scale
        assert err < 2e-5, f"problem {j} {specs[j][:3]}: dW off by {err:.2e} of its largest element ({mode})"
        errb = float((dbias.double() - b_ref).abs().max()) / float(b_ref.abs().max())
        assert errb < 2e-5, f"problem {j}: dbias off by {errb:.2e} ({mode})"


@pytest.mark.gpu
@pytest.mark.parametrize("b,h,t,drop,cond,want_ts", [(4, 192, 400, True, False, True), (3, 192, 160, False, True, True),
                                                     (5, 64, 36, True, True, False), (2, 128, 8, False, False, True),
                                                     (32, 192, 400, True, False, True)])
def test_winograd_gated_in_conv_against_fp64(M, b, h, t, drop, cond, want_ts):
    """The gated 5-tap in-conv (reference layers.py:146-153 + utils.py:31-38) in its Winograd F(4, 5) form (csrc/convwino.hip):
    against fp64 at the accuracy of the native fp32 kernel, with dropout keep bytes, conditioning rows, ragged tile counts
    (utterances whose tile count is not a multiple of a workgroup's 48, tiles of two utterances in one workgroup), and next to
    the direct bf16x6 kernel it replaces — which it must NOT equal bit for bit (that would mean the switch did nothing)."""
    call, ptr = M.hip.call, M.hip.ptr
    dev = "cuda"
    torch.manual_seed(b * 977 + h + t)
    x = torch.randn(b, h, t, device=dev) * torch.exp(torch.randn(1, h, 1, device=dev) * 0.5)
    v_in = torch.randn(2 * h, h, 5, device=dev) * 0.03
    b_in = torch.randn(2 * h, device=dev) * 0.1
    wf_in, _, _ = M.convops.pack_weight(v_in, None)
    keep = (torch.rand(b, 2 * h, t, device=dev) > 0.1).to(torch.uint8) if drop else None
    scale = 1.0 / 0.9 if drop else 1.0
    g = torch.randn(b, 2 * h, device=dev) * 0.3 if cond else None
    table = torch.tensor([[0, h // 16, 2 * h]], dtype=torch.int64, device=dev)
    n_u = M.hip.wino_plane_elems(wf_in.numel())
    u_planes = torch.zeros(3 * n_u, device=dev, dtype=torch.int16)

    def gate():
        acts = torch.full((b, h, t), float("nan"), device=dev)
        ts = torch.full((b, 2 * h, t), float("nan"), device=dev) if want_ts else None
        call("glowtts_conv_gate_fwd", ptr(x), ptr(wf_in), ptr(b_in), ptr(g), ptr(keep), scale, ptr(acts), ptr(ts), b, h, t, 5, 1, 2)
        return torch.cat([acts, ts], 1) if want_ts else acts

    pre = F.conv1d(x.double(), v_in.double(), b_in.double(), padding=2)
    if drop:
        pre = pre * keep.double() * scale
    if cond:
        pre = pre + g.double()[:, :, None]
    th, sg = torch.tanh(pre[:, :h]), torch.sigmoid(pre[:, h:])
    ref = torch.cat([th * sg, th, sg], 1) if want_ts else th * sg
    out = {}
    try:
        for name, mode, wino in (("native", "fp32", 0), ("direct", "bf16x6+wrw", 0), ("winograd", "bf16x6+wrw", 1)):
            M.convops.set_conv_math(mode)
            _split(M, wf_in)
            if wino:
                call("glowtts_wino_weights", ptr(wf_in), wf_in.numel(), ptr(table), 1, ptr(u_planes), n_u)
                M.hip.conv_bind_wino(wf_in, u_planes)
            M.hip.set_knob("WINO", wino)
            out[name] = gate()
    finally:
        M.hip.set_knob("WINO", 1)                        # (the package default)
        M.hip.conv_bind_wino(None)
    err = {k: float((v.double() - ref).abs().max()) for k, v in out.items()}
    assert torch.isfinite(out["winograd"]).all()
    assert not torch.equal(out["winograd"], out["direct"]), "the Winograd kernel did not run"
    try:                                                 # no atomics anywhere in it: launch-to-launch bit-identical
        M.convops.set_conv_math("bf16x6+wrw")
        _split(M, wf_in)
        M.hip.conv_bind_wino(wf_in, u_planes)
        M.hip.set_knob("WINO", 1)
        assert torch.equal(gate(), out["winograd"])
    finally:
        M.hip.set_knob("WINO", 1)
        M.hip.conv_bind_wino(None)
    assert err["winograd"] <= 4 * err["native"] + 2e-7, err         # gate outputs are O(1): absolute = relative
    assert err["winograd"] < 2e-5, err            # (the bound the full-size per-launch test holds the direct kernels to)


@pytest.mark.gpu
def test_flow_stack_takes_the_winograd_gated_in_conv_and_can_be_told_not_to(M):
    """The flow stack's forward launches the Winograd form of the gated in-conv by default (one launch per WN layer and forward
    chain; `glowtts_wino_launches` counts them), `glowtts_set_knob("WINO", 0)` puts the direct kernels back, and the two agree to
    the rounding of the transforms (z, log-det and every gradient within the tolerances of the arithmetic-mode comparison)."""
    from glow_tts_train import models

    torch.manual_seed(5)
    b, t, blocks, layers = 4, 64, 2, 4
    dec = models.FlowSpecDecoder(80, hidden_channels=192, kernel_size=5, dilation_rate=1, n_blocks=blocks, n_layers=layers,
                                 p_dropout=0.0, n_split=4, n_sqz=2).cuda().train()
    y0 = torch.randn(b, 80, t, device="cuda")
    lens = torch.tensor([t, t - 8, t - 20, t // 2], device="cuda")
    mask = (torch.arange(t, device="cuda")[None] < lens[:, None]).float()[:, None]
    with torch.no_grad():
        for f in dec.flows:
            if hasattr(f, "end"):
                f.end.weight.normal_(0, 0.02)
    for p in dec.parameters():                           # (the stack node writes gradients in place: they must exist)
        p.grad = torch.zeros_like(p)
    res = {}
    try:
        for wino in (1, 0):
            M.hip.set_knob("WINO", wino)
            for p in dec.parameters():
                p.grad.zero_()
            y = (y0 * mask).clone().requires_grad_(True)
            before = M.hip.wino_launches()
            z, ld = dec(y, mask)
            launched = M.hip.wino_launches() - before
            (z.square().sum() + ld.sum()).backward()
            res[wino] = (z.detach(), ld.detach(), y.grad.clone(), {k: p.grad.clone() for k, p in dec.named_parameters()}, launched)
    finally:
        M.hip.set_knob("WINO", 1)
    assert res[1][4] in (blocks * layers, 2 * blocks * layers), res[1][4]        # one or two forward chains
    assert res[0][4] == 0
    z1, l1, g1, p1, _ = res[1]
    z0, l0, g0, p0, _ = res[0]
    assert not torch.equal(z1, z0)
    assert float((z1 - z0).abs().max()) <= 2e-5 * max(1.0, float(z0.abs().max()))
    assert float((l1 - l0).abs().max()) <= 1e-4 * max(1.0, float(l0.abs().max()))
    assert float((g1 - g0).abs().max()) <= 1e-4 * float(g0.abs().max())
    for k in p0:
        assert float((p1[k] - p0[k]).abs().max()) <= 2e-4 * max(1e-3, float(p0[k].abs().max())), k
