"""Data-parallel path on the real operators (-m gpu): two ranks (gloo, sharing the one GPU of the test box) train one step
with `parallel.FlowBlockReducer`; the reduced gradient buffer must equal the mean of the two ranks' own gradients — which
it only does if every bucket's all-reduce is launched AFTER the last gradient of the bucket is complete (in-place
gradients written on side streams, deferred un-packing) and on the stream that produced them."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(models, optimize, full=False):
    kw = dict(n_vocab=148, hidden_channels=192, filter_channels=256, filter_channels_dp=64, out_channels=80,
              kernel_size=3, n_heads=2, n_layers_enc=2, p_dropout=0.0, n_blocks_dec=2, kernel_size_dec=5, dilation_rate=1,
              n_block_layers=2, p_dropout_dec=0.0, n_split=4, n_sqz=2, window_size=4, mean_only=True, prenet=True)
    if full:       # BASELINE configs[1] / [3] model: 12 flow blocks x 4 WN layers, 6 encoder layers -> the 6 + 4 = 10 real buckets
        kw.update(filter_channels=768, filter_channels_dp=256, n_layers_enc=6, n_blocks_dec=12, n_block_layers=4)
    model = models.FlowGenerator(**kw).cuda().train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    with torch.no_grad():
        for f in model.decoder.flows:
            if hasattr(f, "end"):
                f.end.weight.normal_(0, 0.01)
    opt = optimize.Adam(model.parameters(), scheduler="noam", dim_model=192, warmup_steps=4000, lr=1.0, betas=(0.9, 0.98), eps=1e-9)
    return model, opt


def _batch(rank, dev, full=False):
    g = torch.Generator().manual_seed(500 + rank)
    if full:       # B = 8 per rank at config-1 lengths (T_text 100, T_mel 400), ragged
        b, tx, ty = 8, 100, 400
        yl = torch.linspace(ty, ty // 2, b).long()
        xl = (yl // 4).clamp(min=1)
        x = torch.randint(1, 148, (b, tx), generator=g) * (torch.arange(tx)[None] < xl[:, None])
        y = torch.randn(b, 80, ty, generator=g) * (torch.arange(ty)[None, None] < yl[:, None, None])
        return x.to(dev), xl.to(dev), y.to(dev), yl.to(dev)
    b, tx, ty = 4, 24, 96
    x = torch.randint(1, 148, (b, tx), generator=g).to(dev)
    xl = torch.tensor([24, 20, 17, 12]).to(dev)
    y = torch.randn(b, 80, ty, generator=g).to(dev)
    yl = torch.tensor([96, 88, 70, 52]).to(dev)
    return x, xl, y, yl


def _grads_of(model, opt, batch, reducer):
    from glow_tts_train._hip import join_side_streams, zero_scope
    from glow_tts_train.convops import flush_groups
    from glow_tts_train.utils import duration_loss, mle_loss

    x, xl, y, yl = batch
    opt.zero_grad()
    with zero_scope(y.device):
        (z, z_m, z_logs, logdet, z_mask), _, (_a, logw, logw_) = model(x, xl, y, yl)
        loss = mle_loss(z, z_m, z_logs, logdet, z_mask) + duration_loss(logw, logw_, xl)
        loss.backward()
        join_side_streams()
        flush_groups()
        launched = sum(reducer._launched) if reducer is not None else 0
        if reducer is not None:
            reducer.finish()
    torch.cuda.synchronize()
    return opt._optim.flat_g.detach().cpu().numpy().copy(), launched


def _worker(rank, world, port, q, full=False):
    import sys
    import torch.distributed as dist

    sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from glow_tts_train import models, optimize, parallel

    if os.environ.get("DP_TEST_OLD_SIGNALS") == "1":        # self-check of this test: the pre-fix double counting must fail it
        def _old(self, p):
            self._seen.discard(id(p))
            parallel.FlowBlockReducer._on_grad(self, p)
        parallel.FlowBlockReducer._on_hook = _old
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(1234)                          # same initial weights on both ranks (broadcast also runs)
        model, opt = _build(models, optimize, full)
        dev = torch.device("cuda", 0)
        first = parallel.FlowBlockReducer(model, opt)
        first.broadcast_parameters(0)
        first.remove_hooks()                             # no reducer listens while the reference gradients are computed
        own = []
        for r in range(world):                           # every rank's un-reduced gradients, computed locally
            g, _ = _grads_of(model, opt, _batch(r, dev, full), None)
            own.append(g)
        red = parallel.FlowBlockReducer(model, opt)
        reduced, launched = _grads_of(model, opt, _batch(rank, dev, full), red)
        again, launched2 = _grads_of(model, opt, _batch(rank, dev, full), red)   # second step: hooks of announced parameters are gone
        mean = np.mean(own, axis=0)
        scale = float(np.abs(mean).max())
        assert np.abs(again - reduced).max() <= 1e-5 * scale, "second reduced step differs from the first"
        q.put((rank, reduced, mean, min(launched, launched2), len(red.buckets)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("full", [False, True], ids=["small", "config1-model-10-buckets"])
def test_flow_block_reducer_on_real_operators_two_ranks(full):
    """`full`: the BASELINE configs[1] / [3] MODEL (12 blocks, 6 encoder layers: 6 + 4 = 10 buckets of 14 / 7 MB) with B = 8 per
    rank — bucket completion order, the three producer streams and `finish()` with the real bucket list."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, full)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, red0, mean0, launched0, nb), (_, red1, mean1, launched1, _) = res
    assert nb == (10 if full else nb)
    assert launched0 >= nb - 3 and launched1 >= nb - 3, "buckets should be reduced while backward is still running"
    scale = float(np.abs(mean0).max())
    np.testing.assert_allclose(red0, red1, rtol=0, atol=1e-6 * scale, err_msg="ranks disagree after the all-reduce")
    # the reference is computed with the same kernels, so only the atomics' summation order differs
    np.testing.assert_allclose(red0, mean0, rtol=2e-3, atol=2e-5 * scale, err_msg="reduced gradients != mean of per-rank gradients")


# ------------------------------------------------------------------------------------------------ ActNorm DDI across ranks
def _ddi_worker(rank, world, port, q):
    import sys
    import torch.distributed as dist

    sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from glow_tts_train import layers

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        xs = [(torch.randn(3, 16, 40, generator=g) * (1 + r) + r).cuda() for r in range(world)]       # rank r's first batch
        lens = torch.tensor([40, 31, 22]).cuda()
        mask = (torch.arange(40).cuda()[None, None] < lens[:, None, None]).float()
        out = {}
        for flag in (False, True):
            an = layers.ActNorm(16, ddi=True).cuda()
            an.ddi_all_reduce = flag
            an(xs[rank], mask)
            out[flag] = (an.logs.detach().flatten().cpu().numpy(), an.bias.detach().flatten().cpu().numpy())
        both = layers.ActNorm(16, ddi=True).cuda()       # what one process sees on the concatenated batch
        both(torch.cat(xs), torch.cat([mask] * world))
        # numpy arrays travel by value: a torch tensor in a Queue is a handle to the SENDER's shared memory, gone when it exits
        q.put((rank, out, (both.logs.detach().flatten().cpu().numpy(), both.bias.detach().flatten().cpu().numpy())))
    finally:
        dist.destroy_process_group()


def test_actnorm_ddi_all_reduce_flag_two_ranks():
    """SURVEY Q10 / hard part 6: by default every rank initialises ActNorm from its OWN first batch and rank 0's parameters win
    at the broadcast (the reference's behaviour); with `ddi_all_reduce` the masked sums are all-reduced first, so every rank
    gets the statistics of the global batch."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddi_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, out0, glob), (_, out1, _) = res
    tt = torch.from_numpy
    assert not torch.allclose(tt(out0[False][0]), tt(out1[False][0]), atol=1e-3), "default: per-rank statistics"
    for a, b, c in zip(out0[True], out1[True], glob):
        assert torch.allclose(tt(a), tt(b), atol=1e-6) and torch.allclose(tt(a), tt(c), atol=1e-5), "flag: statistics of the global batch"


# ------------------------------------------------------------------------------------------------ RCCL on one card
def _nccl_worker(port, q):
    """`backend="nccl"` is RCCL on ROCm.  A group of ONE rank on the test GPU runs the reducer's real launch path: slices
    of the flat gradient buffer all-reduced (AVG) in place from the "comm" stream as their buckets complete during
    backward, `broadcast_parameters`, and `finish()`'s stream joins (this branch had never executed)."""
    import sys
    import torch.distributed as dist

    sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from glow_tts_train import convops, models, optimize, parallel

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        torch.manual_seed(1234)
        model, opt = _build(models, optimize)
        dev = torch.device("cuda", 0)
        batch = _batch(0, dev)
        convops.set_direct_grads(True)                   # the un-reduced reference takes the same in-place route
        plain, _ = _grads_of(model, opt, batch, None)
        convops.set_direct_grads(None)
        red = parallel.FlowBlockReducer(model, opt, force=True, measure=True)
        assert red.backend == "nccl" and red._use_avg and red._active and red._hooks
        before = opt._optim.flat_p.clone()
        red.broadcast_parameters(0)
        torch.cuda.synchronize()
        same_params = bool(torch.equal(before, opt._optim.flat_p))
        reduced, launched = _grads_of(model, opt, batch, red)
        again, launched2 = _grads_of(model, opt, batch, red)     # second step: counters were reset by finish()
        exposed = red.exposed_comm_ms()
        q.put(("ok", plain, reduced, again, launched, launched2, len(red.buckets), same_params, exposed))
    except Exception as exc:                                      # surface the failure in the parent
        import traceback
        q.put(("error", traceback.format_exc(), repr(exc)))
    finally:
        dist.destroy_process_group()


def test_flow_block_reducer_runs_on_rccl_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    proc = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    proc.start()
    res = q.get(timeout=300)
    proc.join(120)
    assert res[0] == "ok", res[1]
    assert proc.exitcode == 0
    _, plain, reduced, again, launched, launched2, nb, same_params, exposed = res
    assert same_params, "broadcast from rank 0 to a group of one must leave the parameters alone"
    assert launched >= nb - 3 and launched2 >= nb - 3, f"only {launched}/{nb} buckets were reduced while backward ran"
    assert len(exposed) == 2 and all(0.0 <= t < 50.0 for t in exposed)
    scale = float(np.abs(plain).max())
    # AVG over one rank is the identity: only the float atomics' summation order separates the runs
    np.testing.assert_allclose(reduced, plain, rtol=2e-3, atol=2e-5 * scale)
    np.testing.assert_allclose(again, plain, rtol=2e-3, atol=2e-5 * scale)


# ------------------------------------------------------------------------------------------------ the reference's DDP wrap
def _ddp_worker(rank, world, port, q):
    """Reference `__main__.py:268-271` unchanged: `DistributedDataParallel(model, device_ids=[local_rank], ...)` around the
    model built by `setup_model` — no environment variable, no call into convops."""
    import sys
    import torch.distributed as dist

    sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("GLOWTTS_DIRECT_GRADS", None)
    from glow_tts_train import convops, models, optimize
    from glow_tts_train.utils import clip_grad_value_

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(1234 + rank)                   # DDP's construction-time broadcast must make rank 0's weights win
        model, opt = _build(models, optimize)
        dev = torch.device("cuda", 0)
        assert not convops.direct_grads_enabled()
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], output_device=0)
        params_after_wrap = opt._optim.flat_p.detach().cpu().numpy().copy()
        with ddp.no_sync():                              # every rank's un-reduced gradients, computed locally
            own = [_grads_of(ddp, opt, _batch(r, dev), None)[0] for r in range(world)]
        reduced, _ = _grads_of(ddp, opt, _batch(rank, dev), None)
        in_place = opt._optim.grads_in_place()
        clip_grad_value_(ddp.parameters(), 5.0)          # what train.py:145 calls on the wrapped model
        opt.step()
        torch.cuda.synchronize()
        q.put((rank, reduced, np.mean(own, axis=0), params_after_wrap, in_place,
               opt._optim.flat_p.detach().cpu().numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_unchanged_ddp_wrap_on_real_operators_two_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, red0, mean0, w0, inpl0, after0), (_, red1, mean1, w1, inpl1, after1) = res
    np.testing.assert_array_equal(w0, w1, err_msg="DDP's parameter broadcast did not reach the flat buffer")
    assert inpl0 and inpl1
    scale = float(np.abs(mean0).max())
    np.testing.assert_allclose(red0, red1, rtol=0, atol=1e-6 * scale, err_msg="ranks disagree after DDP's all-reduce")
    np.testing.assert_allclose(red0, mean0, rtol=2e-3, atol=2e-5 * scale, err_msg="DDP gradients != mean of per-rank gradients")
    np.testing.assert_allclose(after0, after1, rtol=0, atol=1e-6, err_msg="parameters diverge after one update")
    assert np.isfinite(after0).all() and np.abs(after0 - w0).max() > 0


def test_clip_clamps_a_replaced_gradient_on_device():
    import sys
    sys.path[:0] = [os.path.join(ROOT, "glow-tts-train_amd"), ROOT]
    from glow_tts_train import models, optimize, utils

    model, opt = _build(models, optimize)
    opt.zero_grad()
    opt._optim.flat_g.fill_(0.5)
    victim = next(iter(model.parameters()))
    victim.grad = torch.full_like(victim, 9.0)
    norm = utils.clip_grad_value_(model.parameters(), 2.0)
    torch.cuda.synchronize()
    assert float(victim.grad.max()) == 2.0
    n_rest = opt._optim.numel - victim.numel()
    assert float(norm) == pytest.approx((81.0 * victim.numel() + 0.25 * n_rest) ** 0.5, rel=1e-4)
    opt.step()                                            # folds the foreign gradient back into the flat buffer
    assert opt._optim.grads_in_place()


# ============================================================ the driver's SCALE command, rehearsed
def _run_scale_command(n, tmp_path, extra_env=None):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    --gpus N --steps 3 --warmup 1` — the exact launch line of the driver's scaling bench — as a CHILD process, with the
    collectives staged by gloo because the ranks share this box's one card (BENCH_DIST_BACKEND=gloo: a rehearsal of the
    N > 1 control flow, never a measurement).  Returns (returncode, stdout, stderr)."""
    import subprocess
    import sys

    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    (tmp_path / f"scale_{n}.stderr").write_text(r.stderr)
    return r.returncode, r.stdout, r.stderr


@pytest.mark.parametrize("n", [2, 4])
def test_bench_scale_command_control_flow(n, tmp_path):
    """What the driver will run first on an 8-GPU node, here with N ranks on one card: the launcher exits 0 (every rank
    leaves — rank != 0 does not tear the group down under a rank still working), stdout carries exactly ONE JSON line with
    n_gpus = N, the whole-job frame count, weak scaling, the 10 real buckets of the configs[1] model of which >= 7 are
    launched during backward, the per-bucket timing rows, one core share per rank, and no one-GPU diagnostic legs."""
    import json

    rc, out, err = _run_scale_command(n, tmp_path)
    assert rc == 0, err[-3000:]
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["metric"] == "mel_frames_per_sec" and rec["higher_is_better"] is True and rec["dtype"] == "f32"
    assert rec["config"]["global_batch"] == 32 * n and rec["config"]["parallelism"] == f"dp{n}"
    assert "configs[1]" in rec["config"]["workload"]
    assert abs(rec["value"] - n * 32 * 800 / (rec["ms_per_step"] * 1e-3)) <= 1e-6 * rec["value"]
    comm = rec["comm"]
    assert comm["buckets"] == 10 and comm["buckets_launched_during_backward"] >= 7, comm
    assert comm["backend"] == "gloo" and comm["rccl_ranks"] == 0
    assert abs(comm["grad_MB_per_step"] - 114.5) < 1.0
    assert len(comm["per_bucket"]) == 10 and all(b["ready_to_start_ms"] >= 0 for b in comm["per_bucket"])
    assert rec["host_cores_per_rank"] is not None and rec["host_cores_per_rank"] >= 1
    assert "roofline" not in rec and "native_fp32" not in rec and "cpu_baseline" not in rec
    assert np.isfinite(rec["config"]["final_loss"])
