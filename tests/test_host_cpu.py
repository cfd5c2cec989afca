"""CPU tests of the host side: C-ABI surface, state-dict schema, flat-buffer optimizer, data-parallel reducer over
gloo (world_size 2), config, and loud failure without a GPU.  No kernel is launched here."""
import ctypes
import os
import re
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT, T, assert_close, load_golden, split_prefix


# ------------------------------------------------------------------------------------------------ C ABI
def _header_functions():
    text = open(os.path.join(ROOT, "include", "glowtts_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(glowtts_[a-z0-9_]+)\s*\(", text)))


def test_cabi_exports_every_declared_symbol():
    from glow_tts_train import _hip

    so = _hip.library_path()
    assert os.path.exists(so), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(so)
    declared = _header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/glowtts_hip.h but not exported"
    # the Python binding table and the header agree exactly
    assert sorted(_hip.EXPORTED_SYMBOLS) == declared
    lib.glowtts_abi_version.restype = ctypes.c_int
    assert lib.glowtts_abi_version() == 1


def test_cabi_argument_errors_do_not_need_a_gpu():
    """Argument validation happens on the host before any launch: exercise it without a device."""
    from glow_tts_train import _hip

    lib = _hip.load()
    rc = lib.glowtts_invconv_prepare(None, None, None, 4, None)
    assert rc != 0 and b"null pointer" in lib.glowtts_last_error()
    rc = lib.glowtts_mas_path(1, 1, 1, 1, 1, 2100, 10, None)          # non-null dummies; Tx over the limit
    assert rc != 0 and b"2048" in lib.glowtts_last_error()
    rc = lib.glowtts_invconv_fwd(1, 1, 1, None, None, 1, None, 1, 12, 4, 3, None)      # odd group size (layers.py:227)
    assert rc != 0 and b"n_split" in lib.glowtts_last_error()
    assert lib.glowtts_actnorm_fwd(1, 1, 1, 1, None, 1, None, 0, 4, 0, 0, None) == 0   # empty batch: no launch


def test_product_fails_loudly_without_gpu():
    from glow_tts_train import layers, monotonic_align, utils

    f = layers.ActNorm(4)
    with pytest.raises(RuntimeError, match="CPU tensor|HIP"):
        f(torch.zeros(1, 4, 8), torch.ones(1, 1, 8))
    with pytest.raises(RuntimeError):
        monotonic_align.maximum_path(torch.zeros(1, 2, 3), torch.ones(1, 2, 3))
    with pytest.raises(RuntimeError):
        utils.mle_loss(*(torch.zeros(1, 2, 4) for _ in range(3)), torch.zeros(1), torch.ones(1, 1, 4))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "glow-tts-train_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{fn} mentions the oracle"
                assert "/root/reference" not in src


# ------------------------------------------------------------------------------------------------ schema
def _make(tag="base", **over):
    from glow_tts_train import models

    kw = dict(n_vocab=148, hidden_channels=32, filter_channels=64, filter_channels_dp=32, out_channels=80,
              kernel_size=3, n_heads=2, n_layers_enc=2, p_dropout=0.0, n_blocks_dec=2, kernel_size_dec=5,
              dilation_rate=1, n_block_layers=2, p_dropout_dec=0.0, n_speakers=0, gin_channels=0, n_split=4, n_sqz=2,
              sigmoid_scale=False, window_size=4, block_length=None, mean_only=True, hidden_channels_enc=32,
              hidden_channels_dec=32, prenet=True)
    if tag == "spk":
        kw.update(gin_channels=8, n_speakers=3, mean_only=False, sigmoid_scale=True)
    kw.update(over)
    return models.FlowGenerator(**kw)


@pytest.mark.parametrize("tag", ["base", "spk"])
def test_state_dict_schema_matches_reference(tag):
    g = load_golden(f"e2e_{tag}_train")
    want = {k: v.shape for k, v in split_prefix(g, "sd.", as_torch=False).items()}
    got = {k: tuple(v.shape) for k, v in _make(tag).state_dict().items()}
    assert list(got) == list(want), "state-dict key ORDER/NAMES differ from the reference"
    assert got == {k: tuple(s) for k, s in want.items()}
    m = _make(tag)
    m.load_state_dict(split_prefix(g, "sd."))                         # strict load works


def test_default_config_model_size_and_setup_model():
    from glow_tts_train import config, models

    cfg = config.TrainingConfig()
    cfg.model.num_symbols = 148
    model, opt = models.setup_model(cfg, use_cuda=False, create_optimizer=False)
    assert opt is None
    assert sum(p.numel() for p in model.parameters()) == 28_623_889
    assert len(list(model.parameters())) == 519
    assert len(model.decoder.flows) == 36 and model.n_sqz == 2
    flows = list(model.decoder.flows)
    assert [type(f).__name__ for f in flows[:3]] == ["ActNorm", "InvConvNear", "CouplingBlock"]
    assert all(hasattr(f, "store_inverse") for f in flows)
    assert hasattr(flows[0], "set_ddi")
    w = flows[1].weight.detach()
    assert_close(w @ w.t(), torch.eye(4), what="orthogonal init", rtol=0, atol=1e-5)
    assert torch.det(w) > 0
    assert flows[2].end.weight.abs().max() == 0 and flows[2].end.bias.abs().max() == 0
    d = cfg.to_dict()
    cfg2 = config.TrainingConfig.from_dict(d)
    assert cfg2 == cfg and cfg2.model.n_blocks_dec == 12 and cfg2.betas == (0.9, 0.98)


def test_ddi_helper_contract():
    """ddi.py:10-17 flips every flow that has set_ddi; ActNorm then initialises on its first forward."""
    m = _make()
    n = 0
    for f in m.decoder.flows:
        if getattr(f, "set_ddi", False):
            f.set_ddi(True)
            assert not f.initialized
            n += 1
    assert n == 2


def test_generate_path_and_sequence_mask():
    from glow_tts_train import utils
    from oracle import glow_oracle as O

    dur = torch.tensor([[2.0, 0.0, 3.0, 1.0], [1.0, 1.0, 0.0, 0.0]])
    mask = torch.zeros(2, 4, 7)
    mask[0, :4, :6] = 1
    mask[1, :2, :2] = 1
    p = utils.generate_path(dur, mask)
    assert_close(p, O.generate_path(dur, mask), rtol=0, atol=0)
    assert p[0].sum(0).tolist() == [1, 1, 1, 1, 1, 1, 0]
    assert p[0, 1].sum() == 0 and p[0, 2, 2:5].sum() == 3
    assert utils.sequence_mask(torch.tensor([1, 3]), 4).tolist() == [[True, False, False, False], [True, True, True, False]]
    assert utils.convert_pad_shape([[0, 0], [1, 2], [3, 4]]) == [3, 4, 1, 2, 0, 0]
    assert utils.intersperse([1, 2], 0) == [0, 1, 0, 2, 0]


# ------------------------------------------------------------------------------------------------ optimizer
def test_flat_adam_layout_and_noam_schedule():
    from glow_tts_train import optimize

    g = load_golden("e2e_base_train")
    m = _make()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    opt = optimize.Adam(m.parameters(), scheduler="noam", dim_model=32, warmup_steps=4000, lr=1.0)
    flat = opt._optim
    # parameters are now views of ONE buffer, values unchanged, 256-byte aligned starts
    for (k, v) in m.state_dict().items():
        assert torch.equal(v, before[k])
    base = flat.flat_p.data_ptr()
    for p, (o, n) in zip(m.parameters(), flat.slices()):
        assert p.data_ptr() == base + 4 * o and o % 64 == 0 and n == p.numel()
        assert p.grad is not None and p.grad.data_ptr() == flat.flat_g.data_ptr() + 4 * o
    assert flat.numel == sum(p.numel() for p in m.parameters())
    # autograd accumulates INTO the flat buffer, zero_grad keeps the aliasing
    loss = sum((p * p).sum() for p in m.parameters())
    loss.backward()
    assert flat.flat_g.abs().sum() > 0
    first = next(iter(m.parameters()))
    assert_close(first.grad, 2 * first.detach(), rtol=0, atol=0)
    opt.zero_grad()
    assert flat.flat_g.abs().sum() == 0 and first.grad.data_ptr() == flat.flat_g.data_ptr()
    # host mirror of the Noam schedule == the reference's learning rates
    lrs = [opt.cur_lr]
    for _ in range(3):
        opt._update_learning_rate()
        lrs.append(opt.cur_lr)
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-12)
    assert opt.get_lr() == lrs[-1] and opt._optim.param_groups[0]["lr"] == lrs[-1]
    # state-dict round trip in torch.optim.Adam's layout
    flat.flat_m.normal_()
    flat.flat_v.uniform_()
    sd = opt.state_dict()
    assert set(sd) == {"state", "param_groups"} and len(sd["state"]) == len(list(m.parameters()))
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    m2 = _make()
    opt2 = optimize.Adam(m2.parameters(), scheduler="noam", dim_model=32)
    opt2.load_state_dict(sd)
    for o, n in flat.slices():          # (padding between parameters is not part of the state)
        assert torch.equal(opt2._optim.flat_m[o:o + n], flat.flat_m[o:o + n])
        assert torch.equal(opt2._optim.flat_v[o:o + n], flat.flat_v[o:o + n])


# ------------------------------------------------------------------------------------------------ data parallel (gloo)
class _Toy(torch.nn.Module):
    """Same parameter naming scheme as FlowGenerator (encoder.*, decoder.flows.N.*, emb_g) on plain CPU layers."""

    def __init__(self):
        super().__init__()
        self.encoder = torch.nn.Sequential(torch.nn.Linear(6, 6), torch.nn.Linear(6, 6))
        self.decoder = torch.nn.Module()
        self.decoder.flows = torch.nn.ModuleList(torch.nn.Linear(6, 6) for _ in range(6))
        self.emb_g = torch.nn.Embedding(3, 6)      # never used in forward: a parameter without gradient

    def forward(self, x):
        h = self.encoder(x)
        for f in self.decoder.flows:
            h = torch.tanh(f(h))
        return h


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q):
    import sys

    sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
    from glow_tts_train import optimize, parallel

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)               # different initial weights per rank: broadcast must fix that
        model = _Toy()
        opt = optimize.Adam(model.parameters(), scheduler="noam", dim_model=6)
        red = parallel.FlowBlockReducer(model, opt)
        red.broadcast_parameters(0)
        keys = [b.key for b in red.buckets]
        torch.manual_seed(7)
        data = torch.randn(world * 4, 6)
        shard = data[rank * 4:(rank + 1) * 4]
        opt.zero_grad()
        model(shard).pow(2).mean().backward()
        launched_during_backward = sum(red._launched)
        red.finish()
        q.put((rank, keys, launched_during_backward, opt._optim.flat_p.numpy().copy(), opt._optim.flat_g.numpy().copy()))  # by value
    finally:
        dist.destroy_process_group()


def test_flow_block_reducer_gloo_world2():
    from glow_tts_train import optimize

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, keys0, l0, p0, g0), (_, keys1, l1, p1, g1) = res
    p0, g0, p1, g1 = (torch.from_numpy(a) for a in (p0, g0, p1, g1))
    assert keys0 == ["enc.tail", "dec000", "misc"] == keys1     # 6 flows = 2 blocks = ONE decoder bucket; toy encoder = one run
    assert l0 >= 1 and l1 >= 1          # the decoder bucket was reduced while backward was still running
    assert torch.equal(p0, p1), "parameters differ after broadcast"
    assert torch.equal(g0, g1), "averaged gradients differ between ranks"
    # single-process reference: same weights (rank 0's), mean of the per-rank losses == DDP semantics
    torch.manual_seed(100)
    model = _Toy()
    opt = optimize.Adam(model.parameters(), scheduler="noam", dim_model=6)
    torch.manual_seed(7)
    data = torch.randn(world * 4, 6)
    opt.zero_grad()
    (sum(model(data[r * 4:(r + 1) * 4]).pow(2).mean() for r in range(world)) / world).backward()
    assert_close(g0, opt._optim.flat_g, what="averaged grads", rtol=1e-5, atol=1e-7)
    assert g0[opt._optim.slices()[-1][0]:].abs().sum() == 0            # unused embedding: zeros, still reduced


def _ddp_worker(rank, world, port, q):
    """The reference's own wrap (`__main__.py:268-271`): torch DistributedDataParallel around a model whose parameters and
    gradients are views of FlatAdam's flat buffers."""
    import sys

    sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
    from glow_tts_train import convops, optimize, parallel

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    auto_before = convops.direct_grads_enabled()                # no process group yet: in place
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        auto_in_group = convops.direct_grads_enabled()          # a group and nobody listening: through autograd (DDP)
        torch.manual_seed(100 + rank)
        model = _Toy()
        for p in model.emb_g.parameters():                      # DDP wants a gradient for every parameter it manages
            p.requires_grad_(False)
        opt = optimize.Adam(model.parameters(), scheduler="noam", dim_model=6)
        red = parallel.FlowBlockReducer(model, opt)
        auto_with_reducer = convops.direct_grads_enabled()      # our reducer listens: in place again
        red.remove_hooks()
        auto_after_remove = convops.direct_grads_enabled()
        ddp = torch.nn.parallel.DistributedDataParallel(model)  # broadcasts rank 0's parameters INTO the flat buffer
        torch.manual_seed(7)
        data = torch.randn(world * 4, 6)
        opt.zero_grad()
        ddp(data[rank * 4:(rank + 1) * 4]).pow(2).mean().backward()
        in_place = opt._optim.grads_in_place()                  # DDP copies the reduced bucket back into the same views
        q.put((rank, (auto_before, auto_in_group, auto_with_reducer, auto_after_remove), in_place,
               opt._optim.flat_p.numpy().copy(), opt._optim.flat_g.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_unchanged_ddp_wrap_gloo_world2_matches_flow_block_reducer():
    """DDP semantics on the flat buffers == FlowBlockReducer's (mean of per-rank gradients, rank 0's parameters), and the
    operators' in-place-gradient shortcut switches itself off when a process group exists without our reducer."""
    from glow_tts_train import optimize

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, auto0, inpl0, p0, g0), (_, auto1, inpl1, p1, g1) = res
    assert auto0 == auto1 == (True, False, True, False)
    assert inpl0 and inpl1, "DDP must leave .grad as the optimizer's views of the flat buffer"
    p0, g0, p1, g1 = (torch.from_numpy(a) for a in (p0, g0, p1, g1))
    assert torch.equal(p0, p1) and torch.equal(g0, g1)
    torch.manual_seed(100)
    model = _Toy()
    opt = optimize.Adam(model.parameters(), scheduler="noam", dim_model=6)
    torch.manual_seed(7)
    data = torch.randn(world * 4, 6)
    opt.zero_grad()
    (sum(model(data[r * 4:(r + 1) * 4]).pow(2).mean() for r in range(world)) / world).backward()
    assert_close(g0, opt._optim.flat_g, what="DDP-averaged grads", rtol=1e-5, atol=1e-7)


def test_clip_falls_back_to_live_gradients_when_one_was_replaced(monkeypatch):
    """`utils.clip_grad_value_` may clamp the flat buffer in one launch only while every `.grad` still is the
    optimizer's view of it; a replaced `.grad` must be clamped itself (the kernel is stubbed: selection logic only)."""
    from glow_tts_train import optimize, utils

    model = _Toy()
    opt = optimize.Adam(model.parameters(), scheduler="noam", dim_model=6)
    flat = opt._optim
    clamped = []

    def fake_call(name, t, n, clip, sumsq, **kw):
        assert name == "glowtts_clip_grad_value"
        clamped.append(t)
        sumsq += (t.reshape(-1)[:n] ** 2).sum()
        t.clamp_(-clip, clip)

    monkeypatch.setattr(utils, "call", fake_call)
    monkeypatch.setattr(utils, "ptr", lambda t: t)
    opt.zero_grad()
    flat.flat_g.fill_(3.0)
    norm = utils.clip_grad_value_(model.parameters(), 1.0)
    assert len(clamped) == 1 and clamped[0] is flat.flat_g and float(flat.flat_g.max()) == 1.0
    assert float(norm) == pytest.approx(3.0 * flat.numel_padded ** 0.5)
    # a foreign tensor takes the place of one gradient
    clamped.clear()
    victim = next(iter(model.parameters()))
    victim.grad = torch.full_like(victim, 7.0)
    assert not flat.grads_in_place() and flat.clip_grad_value_(1.0) is None
    utils.clip_grad_value_(model.parameters(), 2.0)
    assert len(clamped) == len([p for p in model.parameters()]), "per-tensor path expected"
    assert float(victim.grad.max()) == 2.0, "the live gradient was not clamped"


def test_flat_adam_load_state_dict_rejects_mismatched_state():
    from glow_tts_train import optimize

    m = _make()
    opt = optimize.Adam(m.parameters(), scheduler="noam", dim_model=32)
    sd = opt.state_dict()
    opt.load_state_dict({"state": {}, "param_groups": sd["param_groups"]})      # saved before the first update: fine
    broken = {"state": dict(sd["state"]), "param_groups": sd["param_groups"]}
    del broken["state"][3]
    # torch.optim.Adam has no state for a parameter that never received a gradient: accepted (zero moments) with a warning
    opt._optim.flat_m.fill_(1.0)
    with pytest.warns(UserWarning, match="no state for parameters"):
        opt.load_state_dict(broken)
    o3, n3 = opt._optim.offsets[3], opt._optim._params[3].numel()
    assert float(opt._optim.flat_m[o3:o3 + n3].abs().max()) == 0.0
    broken = {"state": dict(sd["state"]), "param_groups": sd["param_groups"]}
    broken["state"][0] = dict(broken["state"][0], exp_avg=torch.zeros(5))
    with pytest.raises(ValueError, match="shape"):
        opt.load_state_dict(broken)
    other = optimize.Adam(list(m.parameters())[:4], scheduler="noam", dim_model=32)
    with pytest.raises(ValueError, match="parameters"):
        other.load_state_dict(sd)


def test_reducer_single_process_is_a_noop():
    from glow_tts_train import optimize, parallel

    model = _Toy()
    opt = optimize.Adam(model.parameters(), scheduler="noam", dim_model=6)
    red = parallel.FlowBlockReducer(model, opt)
    assert red.world == 1 and not red._hooks
    model(torch.randn(2, 6)).sum().backward()
    red.finish()
    red.broadcast_parameters()
    sizes = [(b.key, b.hi - b.lo) for b in red.buckets]
    assert sizes[0][0].startswith("enc") and all(s > 0 for _, s in sizes)
    # buckets tile the flat buffer in order without overlap
    for a, b in zip(red.buckets, red.buckets[1:]):
        assert a.hi <= b.lo


def test_default_bucket_keys_cut_the_encoder_per_ffn_layer():
    """DP buckets: one per PAIR of flow blocks, and four inside the text encoder (head incl. the attention stack, FFN layers
    0-2, FFN layers 3-5, tail) — contiguous runs of the parameter order, so each is a slice of the flat gradient buffer."""
    from glow_tts_train import config, models, parallel

    cfg = config.TrainingConfig()
    cfg.model.num_symbols = 148
    model, _ = models.setup_model(cfg, use_cuda=False, create_optimizer=False)
    runs = []
    for name, p in model.named_parameters():
        k = parallel.default_bucket_key(name)
        if not runs or runs[-1][0] != k:
            runs.append([k, 0])
        runs[-1][1] += p.numel()
    keys = [k for k, _ in runs]
    assert len(keys) == len(set(keys)), "a bucket key must name ONE contiguous run of parameters"
    assert keys[:4] == ["enc.head", "enc.ffn0", "enc.ffn1", "enc.tail"]
    assert keys[4:] == [f"dec{i:03d}" for i in range(6)]
    sizes = dict(runs)
    assert all(13e6 < 4 * sizes[f"dec{i:03d}"] < 15e6 for i in range(6))       # ~14.3 MB per decoder bucket
    assert sum(n for _, n in runs) == sum(p.numel() for p in model.parameters())


def test_conv_math_mode_names():
    """The arithmetic switch is host state of the library: selectable (and rejected when misspelt) without a GPU."""
    from glow_tts_train import _hip

    before = _hip.conv_math(None)
    try:
        assert _hip.conv_math("bf16x6+wrw") == before and _hip.conv_math(None) == 3 + 4 * 3
        assert _hip.conv_math("bf16") == 15 and _hip.conv_math(None) == 1
        with pytest.raises(ValueError, match="bf16x6"):
            _hip.conv_math("bf16x9")
        with pytest.raises(ValueError):
            _hip.conv_math("bf16x6+all")
        assert _hip.conv_math("fp32") == 1 and _hip.conv_math(None) == 0
    finally:
        _hip.conv_math(before)


def test_tuning_switches_are_latched_listed_and_settable():
    """the library's tuning switches.  (1) No kernel-launch path calls getenv: the ONLY getenv of csrc/ is the
    one-time table fill in error.hip.  (2) Every switch name of that table is documented in include/glowtts_hip.h, and the header
    names no switch the table lacks.  (3) The two skip-work experiment switches exist under GLOWTTS_TRACE only: the shipped
    library does not contain their names.  (4) glowtts_set_knob / glowtts_get_knob work without a GPU and reject unknown names."""
    from glow_tts_train import _hip

    csrc = os.path.join(ROOT, "glow-tts-train_amd", "csrc")
    sites = []
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".hpp")):
            for i, line in enumerate(open(os.path.join(csrc, fn)), 1):
                code = line.split("//")[0]
                if re.search(r"\bgetenv\s*\(|\benv_knob\s*\(", code):
                    sites.append((fn, i))
    assert [f for f, _ in sites] == ["error.hip"], f"getenv outside the one-time table fill: {sites}"

    err = open(os.path.join(csrc, "error.hip")).read()
    table = err[err.index("kKnobs[K_COUNT] = {"):err.index("};", err.index("kKnobs[K_COUNT] = {"))]
    shipped_part, _, trace_part = table.partition("#ifdef GLOWTTS_TRACE")
    shipped = re.findall(r'\{"([A-Z0-9_]+)",', shipped_part)
    trace_only = re.findall(r'\{"([A-Z0-9_]+)",', trace_part)
    assert len(shipped) >= 10 and sorted(trace_only) == ["BND_EXP", "CONV_EXP", "WRW1_EXP"]
    common = open(os.path.join(csrc, "common.hpp")).read()
    enum = common[common.index("enum Knob {"):common.index("K_COUNT", common.index("enum Knob {"))]
    assert re.findall(r"\bK_([A-Z0-9_]+)", enum) == shipped + trace_only, "enum Knob and the name table must have one order"

    header = open(os.path.join(ROOT, "include", "glowtts_hip.h")).read()
    conventions = header[:header.index("#ifndef GLOWTTS_HIP_H")]
    documented = set(re.findall(r"GLOWTTS_([A-Z0-9_]+)\s+\[-?\d+\]", conventions))
    assert documented == set(shipped) | set(trace_only), (documented ^ (set(shipped) | set(trace_only)))
    integration = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in shipped:
        assert "GLOWTTS_" + name in conventions

    blob = open(_hip.library_path(), "rb").read()
    for name in trace_only:
        assert ("GLOWTTS_" + name).encode() not in blob and (name + "\0").encode() not in blob, f"{name} in the shipped library"
    assert b"WRW_TR_PRIO\0" in blob

    before = _hip.get_knob("GLOWTTS_WRW_TR_MT")
    try:
        _hip.set_knob("WRW_TR_MT", 2)
        assert _hip.get_knob("GLOWTTS_WRW_TR_MT") == 2
    finally:
        _hip.set_knob("GLOWTTS_WRW_TR_MT", before)
    with pytest.raises(RuntimeError, match="no tuning switch"):
        _hip.set_knob("GLOWTTS_BND_EXP", 1)
    with pytest.raises(RuntimeError, match="no tuning switch"):
        _hip.get_knob("GLOWTTS_NOT_A_SWITCH")
    assert "glowtts_set_knob" in integration


def test_bench_pmc_keys_exist_in_the_newest_profiles():
    """the bench line's counter fields must be re-derivable from profiles/.  Every kernel key bench.py looks
    up (`_PMC_KERNEL`) is a key of the NEWEST profiles/rNN_pmc.json, its kernel name a row of the kernel trace of the SAME round,
    and the recorded HBM-side traffic at least 0.9 x the launch's algorithmic bytes (a record far below them belongs to another
    kernel — round 4's driver line carried round 1's).  A kernel renamed without a fresh counter pass fails here."""
    import csv
    import json

    import bench

    pmc_path, stats_path = bench.newest_profile("_pmc.json"), bench.newest_profile("_kernel_stats.csv")
    assert pmc_path and stats_path
    assert os.path.basename(pmc_path)[:3] == os.path.basename(stats_path)[:3], "counter passes and kernel trace of one round"
    table = json.load(open(pmc_path))
    stat_names = [r["Name"].replace(" ", "") for r in csv.DictReader(open(stats_path))]
    B, T, H, C = 32, 400, 192, 160
    other_bytes = {"glowtts_flow_boundary_fwd": 4.0 * B * T * (2 * H + 3 * C), "glowtts_coupling_actnorm_invconv_bwd": 5 * 4.0 * B * C * T}
    assert len(bench._PMC_KERNEL) >= 4
    for (tag, math), key in bench._PMC_KERNEL.items():
        assert key in table, f"{tag}: {key!r} is not a key of {os.path.basename(pmc_path)}"
        name = key.split(" grid=")[0]
        assert any("glowtts::" + name + "(" in n or "glowtts::" + name + "<" in n for n in stat_names), f"{name} not in {os.path.basename(stats_path)}"
        m = re.match(r"(glowtts_conv\w*)\[M(\d+) K(\d+)x(\d+) N(\d+)x(\d+)\]", tag)
        if m:
            alg = bench.conv_algorithmic_bytes(m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)),
                                               int(m.group(5)) * int(m.group(6)), H)
            alg *= bench._PMC_PROBLEMS_PER_LAUNCH.get(tag, 1)
        else:
            alg = other_bytes[tag]
        traffic = table[key]["traffic_bytes"]
        assert traffic >= 0.9 * alg, f"{tag}: {traffic / 1e6:.1f} MB of counter traffic for {alg / 1e6:.1f} MB algorithmic"
        assert bench.pmc_entry(tag, math)["pmc_key"] == key and bench.pmc_traffic(tag, math) == traffic
    # a tag without a record gives null, not another kernel's bytes
    assert bench.pmc_traffic("glowtts_conv_wrw[M384 K192x5 N32x400]", "fp32") is None
    assert bench.pmc_entry("glowtts_no_such_kernel") == {}


def test_committed_parity_margins_belong_to_this_code():
    """the committed parity margins (profiles/rNN_parity_margins.json, written on the MI355X by
    tests/test_full_size_parity.py) carry the digest of the kernel sources they were measured on; it must be the digest of the
    sources in this tree, every whole-step case of both arithmetics must be there, and none may sit above the margin the GPU test
    enforces."""
    import json

    import bench
    from helpers import source_digest

    path = bench.newest_profile("_parity_margins.json")
    assert path, "no profiles/rNN_parity_margins.json"
    data = json.load(open(path))
    if "source_digest" not in data:
        pytest.skip(f"{os.path.basename(path)} predates the digest (round 4's file): re-run the full-size tests on the GPU box")
    assert data["source_digest"] == source_digest(), (
        f"{os.path.basename(path)} was measured on other kernel sources ({data['source_digest']} != {source_digest()}): run "
        "tests/test_full_size_parity.py on the GPU box and commit gpurun_out/parity_margins.json as the new profiles/ copy")
    for case in ("train_step_config2", "train_step_config2_dropout", "train_step_config2_dropout_onechain", "train_step_config5",
                 "train_step_config5_dropout"):
        for arith in ("fp32", "bf16x6+wrw"):
            f = data[case][arith]
            assert f["loss_err_over_tol"] <= 1.0 and f["worst_grad_err_over_tol"] <= 1.0, (case, arith, f)
            if f["n_alignment_frames_differing"] == 0:          # (same path on both sides: the drift guard of the GPU test)
                assert f["worst_grad_err_over_tol"] <= 0.5, (case, arith, f)


def test_fastcall_binding_matches_the_ctypes_table():
    """`_glowtts_fastcall` (generated by csrc/gen_fastcall.py, built by `make`) wraps exactly the entry points of
    `_hip._SIGNATURES`, is bound to the addresses of the library ctypes opened, validates its arguments, and `call()` uses
    it; GLOWTTS_FASTCALL=0 or a missing module falls back to ctypes (a binding, never a compute path)."""
    from glow_tts_train import _hip

    lib = _hip.load()
    assert _hip.FASTCALL, "run __graft_entry__.build() (make builds lib/_glowtts_fastcall*.so)"
    assert sorted(_hip._fn_cache) == sorted(_hip._SIGNATURES)
    fast = _hip._fn_cache["glowtts_actnorm_fwd"]
    args = (1, 1, 1, 1, None, 1, None, 0, 4, 0, 0, None)                    # empty batch: validated, nothing launched
    assert fast(*args) == lib.glowtts_actnorm_fwd(*args) == 0
    assert fast(1, 1, 1, 1, None, 1, None, np.int64(0), True, 0, 0, None) == 0     # anything with __index__ is an int
    assert _hip._fn_cache["glowtts_mas_path"](1, 1, 1, 1, 1, 2100, 10, None) != 0 and b"2048" in lib.glowtts_last_error()
    with pytest.raises(TypeError):
        fast(1, 1, 1, 1, None, 1, None, "0", 4, 0, 0, None)
    with pytest.raises(TypeError, match="expected 12"):
        fast(1, 2)


def test_library_has_no_packed_fp32_valu(tmp_path):
    """DESIGN.md lesson 12: on gfx950 a wave's packed-fp32 arithmetic (v_pk_add / v_pk_mul / v_pk_fma_f32) goes wrong in lanes
    48-63 while another wave of the CU runs bf16 MFMAs with ds_write_b64 stores (tools/pk_hazard_repro.py).  The library is built
    with that instruction class removed from instruction selection (csrc/Makefile: NOPK); this disassembles what was built."""
    import shutil
    import subprocess

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    lib = os.path.join(ROOT, "glow-tts-train_amd", "lib", "libglowtts_hip.so")
    if not (os.path.exists(objdump) and os.path.exists(lib)):
        pytest.skip("needs the built library and the ROCm llvm-objdump")
    copy = tmp_path / "lib.so"
    shutil.copy(lib, copy)
    subprocess.run([objdump, "--offloading", str(copy)], check=True, capture_output=True, cwd=tmp_path)
    objs = sorted(p for p in tmp_path.iterdir() if "gfx950" in p.name)
    assert objs, "no gfx950 code object in the library"
    import re

    bad, mfma, aggressors = 0, 0, []
    for o in objs:
        text = subprocess.run([objdump, "-d", str(o)], check=True, capture_output=True, text=True).stdout
        bad += sum(text.count(op) for op in ("v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32"))
        mfma += text.count("v_mfma_f32_16x16x32_bf16")
        # the other half of the hazard: no kernel of ours is an AGGRESSOR either (bf16 MFMAs together with 64-bit LDS stores —
        # ds_write_b64 and ds_write2st64_b64 were measured to trigger it; ds_write2_b32 / b16 / b32 / b96 / b128 do not): a
        # collective or a framework kernel sharing a CU with it may contain packed-fp32 code
        for kernel in re.split(r"\n(?=[0-9a-f]+ <)", text):
            if "bf16" in kernel and "v_mfma" in kernel and re.search(r"ds_write(2|2st64)?_b64\b", kernel):
                aggressors.append(kernel.split(">:")[0].split("<")[-1][:80])
    assert mfma > 1000, "disassembly looks empty"
    assert bad == 0, f"{bad} packed-fp32 VALU instructions in libglowtts_hip.so"
    assert not aggressors, f"bf16-MFMA kernels with 64-bit LDS stores: {aggressors[:4]}"


def test_library_has_no_early_mfma_result_read_behind_a_branch():
    """csrc/common.hpp mfma_settle(): ROCm 7.2 pads the distance between an MFMA and the first read of its accumulators in
    layout order; behind a TAKEN forward branch that distance can be shorter than the matrix pipe needs (no interlock), and
    the read returns the accumulator before the last pass has landed.  tools/mfma_hazard_scan.py walks the control-flow graph
    of every kernel of the built library; round 4's attention kernels had 185 such paths (24 kernels), wrong results in ~25 %
    of launches at T = 128 / 160 / 192."""
    import importlib.util

    lib = os.path.join(ROOT, "glow-tts-train_amd", "lib", "libglowtts_hip.so")
    if not (os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump") and os.path.exists(lib)):
        pytest.skip("needs the built library and the ROCm llvm-objdump")
    spec = importlib.util.spec_from_file_location("mfma_hazard_scan", os.path.join(ROOT, "tools", "mfma_hazard_scan.py"))
    scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scan)
    found, stats, n_kernels = scan.scan(lib)
    assert n_kernels > 100 and len(stats["linear"]) >= 3, "disassembly looks empty"
    assert not found, [(f[0][:60], hex(f[1]), f[2], f[6], f[7]) for f in found[:6]]
