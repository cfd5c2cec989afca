#!/usr/bin/env python3
"""bench.py — mel-frames/s of ONE full Glow-TTS training step (BASELINE.json metric) on N MI355X of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" = zero_grad + FlowGenerator.forward (text encoder, 12-block flow decoder, device-resident MAS) + mle_loss +
duration_loss + backward + (N>1: bucketed RCCL gradient all-reduce overlapped with backward) + clip_grad_value_ +
Adam/Noam, on a synthetic batch already resident in HBM.  Workload = BASELINE.json configs[1]: per-GPU B=32,
T_text=160, T_mel=800, 80 mels, 12 flow blocks, n_split=4, fp32, ModelConfig defaults (dropout 0.1 / 0.05 active),
random-init weights, data-dependent ActNorm init done before timing.  Weak scaling: every rank gets its own B=32.

Rank 0 prints ONE JSON line.  Besides the driver's contract it carries
  "roofline"     : the dominant hand-written HIP launch of the step, chosen MECHANICALLY: arg-max of launches per step x mean
                   duration over the table of launches the step makes (`as_launched_top5`).  Mean durations come from an
                   instrumented pass after the timed region (HIP events on the launch stream around every launch); the two
                   families the step batches (a WN stack's four 5-tap weight gradients = one glowtts_conv_wrw_batch launch, a
                   block's 1x1 weight gradients = one glowtts_conv_wrw1_multi launch) are re-priced with that batched launch
                   timed back to back.  In the default arithmetic ("bf16x6+wrw": each fp32 operand as three bf16 planes, six
                   products per fp32 product on v_mfma_f32_16x16x32_bf16) the object carries THREE fractions:
                     frac_algorithmic = algorithmic FLOPs (2*M*K*taps*columns) / t / 2.5 PFLOP/s dense bf16 peak,
                     frac_pipe        = 6 x that (the MFMA work the pipe really does) = `frac`,
                     mfma_busy_measured = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs) from the NEWEST committed
                                        counter pass (profiles/rNN_pmc.json); `traffic` = HBM-side bytes per launch from the same
                                        record (`pmc_source`, `pmc_key` say which).  Both are null when that file has no record
                                        under exactly this kernel's name and grid — never another kernel's, never an older round's.
                   "wrw5_batch_as_launched_in_the_step": the batched 5-tap weight gradient (the dominant launch until round 4)
                   whichever launch leads; "wrw1_multi_as_launched_in_the_step": a flow block's six 1x1 weight gradients.
                   Every MFMA kernel is listed with its algorithmic bytes and FLOPs, its time at each roof and `bound: hbm|mfma`
                   (the 1x1 convolutions are byte-bound: reported against 8 TB/s); every streaming kernel with GB/s against 8 TB/s;
                   plus SURVEY.md 8d(i)'s invertible subset (un-fused kernels of the instrumented pass, and
                   `as_launched_in_the_step`: the block-boundary launch forward / the fused kernel backward) and the decoder
                   alone (8d(ii));
  "other_configs": BASELINE configs[2] (bf16 tensors, B=64, T_mel=1000) and configs[4] (speaker-conditioned, B=48, T_mel=1200,
                   20 blocks) timed after everything else, 6 warm-up + 8 steps each on fresh models (N=1 only; never `value`);
  "cpu_baseline" : the CPU oracle (oracle/glow_oracle.py, a port) on this host's cores as BASELINE.md section 3 prescribes:
                   3 warm-up + 10 timed full steps, median, all usable cores, plus a 1-thread run on a shorter sample; the
                   CPU model is stated.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "glow-tts-train_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2, dense, = fp32 vector peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (AMD's 5 PF headline includes 2:1 sparsity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE config 2: 32)")
    ap.add_argument("--t-mel", type=int, default=800)
    ap.add_argument("--t-text", type=int, default=0, help="default T_mel / 5 (SURVEY.md §8)")
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--speakers", type=int, default=0, help="speaker-conditioned couplings (BASELINE config 5: 4)")
    ap.add_argument("--gin", type=int, default=64, help="speaker embedding width when --speakers > 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the leg that times BASELINE configs[2] and configs[4] for a few steps after the main workload")
    ap.add_argument("--no-split-math", action="store_true",
                    help="skip the extra leg that times the same step with the WN convolutions in bf16x6 split arithmetic")
    ap.add_argument("--cpu-steps", type=int, default=10, help="timed CPU-oracle steps on all cores (BASELINE.md section 3: 10)")
    ap.add_argument("--cpu-warmup", type=int, default=3)
    ap.add_argument("--cpu-steps-1thread", type=int, default=2, help="timed CPU-oracle steps of the 1-thread run (0 = skip)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16: add a leg (`bf16_io`, beside `value`, never instead of it) that times the same step with the flow "
                         "decoder's activation tensors kept in HBM as bf16 (BASELINE configs[2]'s arithmetic; use with --batch 64 "
                         "--t-mel 1000 for its sizes)")
    ap.add_argument("--rccl-self", action="store_true",
                    help="N=1 rehearsal: open a one-rank RCCL group and run the DP reducer's real launch path (buckets "
                         "all-reduced from the comm stream during backward); reported under `comm`, for inspection only")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a captured hipGraph (GraphedTrainStep); default is eager launches — the step is "
                         "GPU-bound, so replay buys <1 %% at this shape")
    return ap.parse_args()


def algorithmic_bytes(B, C, H, Ts, T_text, n_params_padded):
    """Ideal-fusion HBM bytes per LAUNCH of each hand-written kernel at this shape (fp32; N = B*Ts squeezed columns).
    Every operand read once, every result written once; masks / parameters (KB) ignored.  DESIGN.md §Kernels."""
    N, e = B * Ts, 4
    X = C * N * e            # one squeezed flow tensor
    Hb = H * N * e           # one hidden tensor
    return {
        "glowtts_actnorm_fwd": 2 * X,
        "glowtts_actnorm_bwd": 3 * X,
        "glowtts_invconv_fwd": 2 * X,
        "glowtts_invconv_bwd": 3 * X,
        "glowtts_actnorm_invconv_fwd": 2 * X,   # the two elementwise flows of a block in one pass
        "glowtts_actnorm_invconv_bwd": 3 * X,
        "glowtts_coupling_fwd": 3 * X,
        "glowtts_coupling_bwd": 4 * X,          # reads x1, logs (0.5 X each) + dz (X); writes dx, dout (X each)
        "glowtts_gate_fwd": 3 * Hb,
        "glowtts_gate_bwd": 5 * Hb,
        "glowtts_gate_bwd_ts": 5 * Hb,           # reads tanh/sigmoid (2 Hb) + dacts (Hb), writes d(pre-activation) (2 Hb)
        "glowtts_res_skip_fwd": 5.25 * Hb,       # 3 x (5..6 Hb) non-last + 1 x 3 Hb last, mean over the 4 layers
        "glowtts_res_skip_bwd": 3.5 * Hb,        # 3 x 4 Hb + 1 x 2 Hb
        "glowtts_squeeze": 2 * X,
        "glowtts_unsqueeze": 2 * X,
        "glowtts_mle_fwd": 3 * X,
        "glowtts_mle_bwd": 6 * X,
        "glowtts_clip_grad_value": 2 * n_params_padded * e,
        "glowtts_adam_noam": 7 * n_params_padded * e,
        "glowtts_mas_path": B * T_text * (Ts * 2) * 8,
    }


INVERTIBLE_SUBSET = ("glowtts_actnorm_fwd", "glowtts_actnorm_bwd", "glowtts_invconv_fwd", "glowtts_invconv_bwd",
                     "glowtts_actnorm_invconv_fwd", "glowtts_actnorm_invconv_bwd", "glowtts_coupling_fwd",
                     "glowtts_coupling_bwd")


# bench tag -> (HIP kernel, grid size[#duration class]) in the NEWEST committed counter passes (profiles/rNN_pmc.json), per
# arithmetic of the WN convolutions.  tests/test_host_cpu.py::test_bench_pmc_keys_exist_in_the_newest_profiles holds every value to a
# key of that file, its kernel name to a row of the newest rNN_kernel_stats.csv, and its traffic to >= 0.9 x the algorithmic bytes:
# a kernel that is renamed (a new template argument) without a new counter pass fails the CPU suite instead of silently
# reporting another kernel's bytes.
_PMC_KERNEL = {
    # the launch the step makes for a WN stack's four 5-tap weight gradients: one round of 216 workgroups (the pre-net's single
    # 5-tap problems have the same grid size: tools/rocpd_summary.py separates the two duration classes)
    ("glowtts_conv_wrw[M384 K192x5 N32x400]", "bf16x6+wrw"): "convwrw_tr_kernel<3,5,4,false,2> grid=110592#long",
    # the gated in-conv in its Winograd F(4, 5) form (csrc/convwino.hip): 201 workgroups of 256 threads for the whole batch
    ("glowtts_conv_gate_fwd[M384 K192x5 N32x400]", "bf16x6+wrw"): "wino_gate_fwd_kernel<0> grid=51456",
    ("glowtts_conv_fwd[M192 K384x5 N32x400]", "bf16x6+wrw"): "convgemm_split_kernel<3,1,5,4,5,0,3> grid=122880",
    ("glowtts_conv_gate_bwd[M192 K384x1 N32x400]", "bf16x6+wrw"): "convgemm_split_kernel<3,1,5,5,1,0,3> grid=122880",
    ("glowtts_conv_res_skip_fwd[M384 K192x1 N32x400]", "bf16x6+wrw"): "convgemm_split_kernel<3,2,5,2,1,0,3> grid=122880",
    ("glowtts_flow_boundary_fwd", "bf16x6+wrw"): "flow_boundary_fwd_kernel<4,32> grid=106496",
    ("glowtts_coupling_actnorm_invconv_bwd", "bf16x6+wrw"): "coupling_ai_bwd_kernel<4,4> grid=51200",
}
# kernels whose launch the step makes with SEVERAL problems: the record is per launch, the algorithmic bytes per problem
_PMC_PROBLEMS_PER_LAUNCH = {"glowtts_conv_wrw[M384 K192x5 N32x400]": 4}


def newest_profile(suffix):
    """Path of profiles/rNN<suffix> with the largest NN (None when there is none): the bench line reads committed counter /
    kernel-trace records of the NEWEST round only — an older round's file describes older kernels."""
    import re as _re

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    best = None
    try:
        for fn in os.listdir(here):
            m = _re.fullmatch(r"r(\d+)" + _re.escape(suffix), fn)
            if m and (best is None or int(m.group(1)) > best[0]):
                best = (int(m.group(1)), os.path.join(here, fn))
    except OSError:
        return None
    return best[1] if best else None


def pmc_entry(kernel_tag, math="bf16x6+wrw"):
    """The committed rocprofv3 PMC record of `kernel_tag` (FETCH_SIZE / WRITE_SIZE / MFMA-busy cannot be collected from inside
    this process): the newest profiles/rNN_pmc.json — tools/pmc_passes.sh over this same bench command, one counter group per
    run, combined per kernel (exact name AND grid size) by tools/pmc_combine.py as MI355X_MICROARCH.md prescribes (FETCH_SIZE
    doubled on gfx950).  {} when that file has no record under exactly this key: never another kernel's, never an older round's."""
    key = _PMC_KERNEL.get((kernel_tag, math))
    path = newest_profile("_pmc.json")
    if key is None or path is None:
        return {}
    try:
        table = json.load(open(path))
    except Exception:
        return {}
    if key not in table:
        log(f"no PMC record for {kernel_tag!r} (key {key!r}) in {os.path.basename(path)}: traffic / mfma_busy reported as null")
        return {}
    return dict(table[key], source="profiles/" + os.path.basename(path), pmc_key=key)


def subset_from_kernel_trace(x_bytes, survey_gb, blocks):
    """The kernels that carry the invertible subset (ActNorm, InvConvNear, affine apply) in the newest committed rocprofv3 kernel
    trace (profiles/rNN_kernel_stats.csv, one stream: the launches the step makes), beside the HIP-event figures of this run
    (which include each launch's latency).  What the step launches since round 4: forward — ONE flow_boundary_fwd_kernel per
    block boundary (end conv + affine apply + ActNorm + InvConv + start conv: the subset's forward half cannot be timed apart from
    the two contractions it shares the launch with, so that launch is counted WHOLE, with the 4.4 X it moves); backward —
    coupling_ai_bwd_kernel (5 X); the first block's ActNorm + InvConv and the last block's coupling on the un-fused kernels.
    `frac` = bytes of the launches found / their time / 8 TB/s; `survey_frac` = SURVEY 8d(i)'s un-fused 19.5 X per block over that
    time — an UNDER-estimate of the subset's own fraction, since the forward launch's time includes two convolutions.  A fraction
    above 1 would mean launches were missed: None is reported instead."""
    import csv

    path = newest_profile("_kernel_stats.csv")
    names = {"actnorm_invconv_fwd_kernel": 2.0, "actnorm_invconv_bwd_kernel": 3.0, "coupling_fwd_kernel": 3.0, "coupling_bwd_kernel": 4.0,
             "coupling_ai_fwd_kernel": 3.0, "coupling_ai_bwd_kernel": 5.0,
             # skip (H) + y in; out, y, h0 (H) out = (2 H + 3 C) / C = 5.4 X at H = 192, C = 160
             "flow_boundary_fwd_kernel": 5.4, "flow_boundary_bwd_kernel": 8.6}          # X per launch
    try:
        rows = list(csv.DictReader(open(path)))
    except Exception:
        return None
    steps = next((int(r["Calls"]) for r in rows if "glowtts::adam_kernel<" in r["Name"]), 0)
    if not steps:
        return None
    ns, xs, us, calls = 0.0, 0.0, {}, {}
    for r in rows:
        for n, nx in names.items():
            if "glowtts::" + n + "<" in r["Name"]:
                ns += float(r["TotalDurationNs"])
                xs += nx * int(r["Calls"])
                us[n] = round(float(r["AverageNs"]) / 1e3, 2)
                calls[n] = round(int(r["Calls"]) / steps, 2)
    if not ns:
        return None
    ms = ns / steps / 1e6
    launched = xs / steps * x_bytes
    frac, sfrac = launched / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, survey_gb / (ms * 1e-3) / HBM_PEAK_GBS
    complete = calls.get("flow_boundary_fwd_kernel", 0) + calls.get("coupling_ai_fwd_kernel", 0) + calls.get("actnorm_invconv_fwd_kernel", 0) >= blocks - 0.5
    return {"source": "profiles/" + os.path.basename(path), "mean_us": us, "launches_per_step": calls, "ms_per_step": round(ms, 3),
            "alg_GB_as_launched": round(launched / 1e9, 3),
            "frac": round(frac, 4) if frac <= 1 and complete else None,
            "survey_frac": round(sfrac, 4) if sfrac <= 1 and complete else None,
            "note": "forward: the whole flow_boundary_fwd launch (two 1x1 contractions inside) is counted, so survey_frac under-states "
                    "the subset's own fraction"}


def fused_flows_time(args, dev, iters=30):
    """What the step's flow stack launches BETWEEN the WN stacks of blocks k and k + 1 (convops.FlowStackFn, fp32 tensors), HIP
    events on the launch stream around back-to-back launches at the benchmark's shape (each figure includes a launch's latency):
    fwd — `glowtts_flow_boundary_fwd` (csrc/flow_boundary.hip: end conv(k) + affine apply(k) + ActNorm + InvConv(k + 1) + start
          conv(k + 1) in ONE launch; algorithmic bytes skip + y in, out + y + h0 out = (2 H + 3 C) N e);
    bwd — `glowtts_coupling_actnorm_invconv_bwd` (ActNorm + InvConv(k + 1) backward + coupling(k) backward: 5 X), between the
          separately launched backward-data convolutions (the one-launch backward boundary is opt-in: no gain in the step)."""
    from glow_tts_train import _hip

    b, t, c, h, ns = args.batch, args.t_mel // 2, 160, 192, 4
    f = lambda *sh: torch.randn(*sh, device=dev)                                                       # noqa: E731
    y_prev, out_prev, dz = f(b, c, t), f(b, c, t) * 0.1, f(b, c, t)
    mask, x_len = torch.ones(b, t, device=dev), torch.full((b,), float(t), device=dev)
    logs, bias, w = f(c) * 0.1, f(c) * 0.1, torch.linalg.qr(f(ns, ns))[0].contiguous()
    w_inv, logdet_w = torch.inverse(w).contiguous(), torch.zeros(1, device=dev)
    ld_prev, ld, dld = torch.zeros(b, device=dev), torch.zeros(b, device=dev), f(b)
    dy_prev, dout_prev = torch.empty(b, c, t, device=dev), torch.empty(b, c, t, device=dev)
    dlogs, dbias, dw = torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.zeros(ns, ns, device=dev)
    skip = f(b, h, t)
    wp_end, b_end = f(h // 16, c, 16) * 0.05, f(c) * 0.1
    wp_start, b_start = f((c // 2 + 15) // 16, h, 16) * 0.05, f(h) * 0.1
    out, y, h0 = torch.empty(b, c, t, device=dev), torch.empty(b, c, t, device=dev), torch.empty(b, h, t, device=dev)
    P = lambda x: x.data_ptr()                                                                         # noqa: E731
    fwd = lambda: _hip.call("glowtts_flow_boundary_fwd", P(skip), P(wp_end), P(b_end), P(y_prev), P(mask), P(logs), P(bias), P(w),   # noqa: E731
                            P(logdet_w), P(x_len), P(wp_start), P(b_start), P(out), P(y), P(h0), P(ld_prev), P(ld), b, c, h, t, ns, 0)
    bwd = lambda: _hip.call("glowtts_coupling_actnorm_invconv_bwd", P(y_prev), P(out_prev), P(mask), P(logs), P(bias), P(w),   # noqa: E731
                            P(w_inv), P(dz), P(dld), P(x_len), P(dy_prev), P(dout_prev), P(dlogs), P(dbias), P(dw), b, c, t, ns, 0)
    cur = torch.cuda.current_stream(dev)
    res = {}
    x_bytes = 4.0 * b * c * t
    fwd_bytes = 4.0 * b * t * (2 * h + 3 * c)
    for name, fn, nbytes, tag in (("fwd", fwd, fwd_bytes, "glowtts_flow_boundary_fwd"),
                                  ("bwd", bwd, 5 * x_bytes, "glowtts_coupling_actnorm_invconv_bwd")):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for _ in range(iters):
            fn()
        e1.record(cur)
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / iters
        pe = pmc_entry(tag)
        res[name] = {"entry": tag, "mean_us": round(us, 2), "alg_MB": round(nbytes / 1e6, 2),
                     "hbm_frac": round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                     "traffic": pe.get("traffic_bytes"), "pmc_source": pe.get("source")}
    res["fwd"]["what"] = ("end conv + affine apply + ActNorm + InvConv + start conv of a block boundary in one launch: the subset's "
                          "forward half is not a launch of its own any more")
    nb = args.blocks
    res["ms_per_step_fused_part"] = round((nb - 1) * (res["fwd"]["mean_us"] + res["bwd"]["mean_us"]) / 1e3, 3)
    res["alg_GB_fused_part"] = round((nb - 1) * (fwd_bytes + 5 * x_bytes) / 1e9, 3)
    res["note"] = ("per step: blocks - 1 launches of each, plus the first block's ActNorm + InvConv and the last block's coupling "
                   "on the un-fused kernels (timed in the instrumented pass)")
    return res


def pmc_traffic(kernel_tag, math="bf16x6+wrw"):
    """HBM-side bytes per launch from the newest counter pass, or None (never a value measured on another kernel)."""
    return pmc_entry(kernel_tag, math).get("traffic_bytes")


def conv_algorithmic_bytes(name, M, K, taps, cols, H):
    """Ideal-fusion HBM bytes of one launch of a convolution kernel (fp32 tensors; weights included): every operand read
    once, every result written once.  cols = B x T columns."""
    e = 4
    w = M * K * taps * e
    per_col = {
        "glowtts_conv_gate_fwd": K + H + 2 * H,            # x in; acts out; tanh / sigmoid kept for the backward
        "glowtts_conv_res_skip_fwd": K + (4 * H if M == 2 * H else 2 * H),   # acts, x, skip in; x, skip out (last layer: skip only)
        "glowtts_conv_gate_bwd": K + 2 * H + 2 * H,        # d_rs (two sources); tanh / sigmoid; d(pre-activation) out
        "glowtts_conv_wrw": K + M, "glowtts_conv_wrw2": K + M,
    }.get(name, K + M + (M if name == "glowtts_conv_fwd" and taps == 5 and M == H else 0))   # 5-tap backward-data adds a tensor
    return per_col * cols * e + w


def batched_wrw_time(tag, dev, n=4, iters=20):
    """`glowtts_conv_wrw_batch` on n problems of the dominant kernel's shape (distinct random operands per problem, as the layers of a
    WN stack have), HIP events on the launch stream around `iters` back-to-back launches: microseconds per PROBLEM and the pipe
    fraction that gives.  The roofline's `frac` stays the single launch measured in the instrumented pass."""
    import ctypes
    import re as _re

    from glow_tts_train._hip import call, ptr

    m_, k_, taps_, b_, t_ = (int(v) for v in _re.match(r"\w+\[M(\d+) K(\d+)x(\d+) N(\d+)x(\d+)\]", tag).groups())
    xs = [torch.randn(b_, k_, t_, device=dev) for _ in range(n)]
    ds = [torch.randn(b_, m_, t_, device=dev) for _ in range(n)]
    dw = [torch.zeros(taps_, k_, m_, device=dev) for _ in range(n)]
    keep = []

    def parr(ts):
        a = (ctypes.c_void_p * len(ts))(*[x.data_ptr() for x in ts])
        keep.append(a)
        return ctypes.addressof(a)

    ax, ad, aw = parr(xs), parr(ds), parr(dw)

    def run():
        call("glowtts_conv_wrw_batch", n, ax, xs[0].stride(0), ad, ds[0].stride(0), None, 0, 0, None, None, aw, None, b_, k_, m_, t_,
             taps_, 1, (taps_ - 1) // 2)

    for _ in range(3):
        run()
    cur = torch.cuda.current_stream(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    for _ in range(iters):
        run()
    e1.record(cur)
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters / n
    flop = 2.0 * m_ * k_ * taps_ * b_ * t_
    return {"problems_per_launch": n, "us_per_problem": round(us, 2),
            "frac_pipe": round(6.0 * flop / (us * 1e-6) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
            "note": "one glowtts_conv_wrw_batch launch per WN stack in the timed step; alone, back-to-back, distinct operands"}


def wino_gate_time(tag, dev, iters=40):
    """The gated in-conv of `tag`'s shape as the timed step launches it when the Winograd form is on (csrc/convwino.hip): the
    instrumented pass drives every convolution through the per-operator path, which has no Winograd-domain planes and therefore times
    the DIRECT kernel.  Weights packed, split and transformed as the flow stack's arena does, then HIP events around `iters`
    back-to-back launches with dropout keep bytes (as in the step).  None when the launches did not take the Winograd kernel."""
    import re as _re

    from glow_tts_train import _hip, convops
    from glow_tts_train._hip import call, ptr

    m_, k_, taps_, b_, t_ = (int(v) for v in _re.match(r"\w+\[M(\d+) K(\d+)x(\d+) N(\d+)x(\d+)\]", tag).groups())
    h_ = k_
    x = torch.randn(b_, h_, t_, device=dev)
    wf, _, _ = convops.pack_weight(torch.randn(m_, k_, taps_, device=dev) * 0.03, None)
    bias = torch.zeros(m_, device=dev)
    keep = (torch.rand(b_, m_, t_, device=dev) > 0.05).to(torch.uint8)
    acts, ts = torch.empty(b_, h_, t_, device=dev), torch.empty(b_, m_, t_, device=dev)
    planes = torch.empty(3 * wf.numel(), device=dev, dtype=torch.int16)
    call("glowtts_conv_split_weights", ptr(wf), wf.numel(), ptr(planes))
    n_u = _hip.wino_plane_elems(wf.numel())
    u = torch.zeros(3 * n_u, device=dev, dtype=torch.int16)
    table = torch.tensor([[0, k_ // 16, m_]], dtype=torch.int64, device=dev)
    call("glowtts_wino_weights", ptr(wf), wf.numel(), ptr(table), 1, ptr(u), n_u)
    _hip.conv_bind_planes(wf.reshape(-1), planes)
    _hip.conv_bind_wino(wf.reshape(-1), u)
    try:
        def run():
            call("glowtts_conv_gate_fwd", ptr(x), ptr(wf), ptr(bias), None, ptr(keep), 1.0 / 0.95, ptr(acts), ptr(ts), b_, h_, t_, taps_, 1,
                 (taps_ - 1) // 2)

        before = _hip.wino_launches()
        for _ in range(5):
            run()
        if _hip.wino_launches() - before != 5:
            return None
        cur = torch.cuda.current_stream(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        for _ in range(iters):
            run()
        e1.record(cur)
        torch.cuda.synchronize()
    finally:
        _hip.conv_bind_planes(None)
        _hip.conv_bind_wino(None)
    return {"us_per_launch": round(1e3 * e0.elapsed_time(e1) / iters, 2),
            "note": "the Winograd F(4,5) kernel alone, back to back, with dropout keep bytes; the instrumented pass times the direct kernel"}


def wrw1_multi_time(args, dev, iters=20):
    """`glowtts_conv_wrw1_multi` on a flow block's six 1x1 weight-gradient problems (three two-source res/skip gradients, the last
    layer's, the end conv's, the start conv's) at the benchmark's shape, as the block executor launches them: HIP events on the
    launch stream around back-to-back launches.  The kernel is bound by the rate at which a compute unit can take in its operands
    (fp32 rows of x and d, x once per 128 output channels): both roofs are reported."""
    import ctypes

    from glow_tts_train import _hip

    b, t, h, c = args.batch, args.t_mel // 2, 192, 160
    specs = [(h, 2 * h, h, False, 0)] * 3 + [(h, h, 0, False, 0), (h, c, 0, False, 0), (c // 2, h, 0, True, c)]
    probs = (_hip.Wrw1Problem * len(specs))()
    keep = []
    mask = torch.ones(b, t, device=dev)
    flop = 0.0
    alg_bytes = 0.0
    for j, (cin, m, split, md, xw) in enumerate(specs):
        x = torch.randn(b, xw or cin, t, device=dev)
        d = torch.randn(b, split if split else m, t, device=dev)
        d2 = torch.randn(b, m - split, t, device=dev) if split else None
        dwp, dbias = torch.zeros(cin, m, device=dev), torch.zeros(m, device=dev)
        q = probs[j]
        q.x, q.d, q.d2 = x.data_ptr(), d.data_ptr(), (d2.data_ptr() if split else None)
        q.mask_d, q.mask_x = (mask.data_ptr() if md else None), None
        q.dwp, q.dbias = dwp.data_ptr(), dbias.data_ptr()
        q.x_bs, q.d_bs, q.d2_bs = x.shape[1] * t, d.shape[1] * t, ((m - split) * t if split else 0)
        q.Cin, q.M, q.d_split = cin, m, split
        keep.append((x, d, d2, dwp, dbias))
        flop += 2.0 * cin * m * b * t
        alg_bytes += 4.0 * ((cin + m) * b * t + cin * m)
    run = lambda: _hip.call("glowtts_conv_wrw1_multi", len(specs), ctypes.addressof(probs), b, t)   # noqa: E731
    for _ in range(3):
        run()
    cur = torch.cuda.current_stream(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    for _ in range(iters):
        run()
    e1.record(cur)
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    return {"problems_per_launch": len(specs), "mean_us": round(us, 2), "alg_GFLOP": round(flop / 1e9, 3),
            "alg_MB": round(alg_bytes / 1e6, 2),
            "frac_pipe": round(6.0 * flop / (us * 1e-6) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
            "hbm_frac": round(alg_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "note": "one launch per flow block in the timed step (replaces one 3-problem batch and three single launches of the "
                    "frame-packed kernel); split-K sized for half the CUs, as the step launches it; alone, back-to-back"}


def decoder_alone(model, batch, cfg, n_iter=5):
    """SURVEY.md 8(d) secondary metric and headline (ii): FlowSpecDecoder forward + backward alone, one stream (outside a
    training step's scope nothing is put on side streams), HIP events on that stream around each half, median of
    `n_iter`.  frac = max(bytes_alg / 8 TB/s, flops_alg / 157.3 TFLOP/s) / t with SURVEY's per-column figures:
    bytes fwd = (6.5 C + 28 H) e per squeezed column per block, FLOPs fwd = 2(C/2)H + L 2H 2H k + (L-1) 2H 2H + 2HH + 2HC,
    both x3 for fwd + bwd."""
    _, _, y, y_lengths, speaker_ids = batch
    mc = cfg.model
    g = None
    if speaker_ids is not None:
        with torch.no_grad():
            g = torch.nn.functional.normalize(model.emb_g(speaker_ids)).unsqueeze(-1)
    B, _, T = y.shape
    z_mask = torch.ones(B, 1, T, device=y.device)
    r = torch.randn_like(y)
    cur = torch.cuda.current_stream()
    fwd, bwd = [], []
    for _ in range(n_iter + 1):
        yy = y.clone().requires_grad_(True)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(cur)
        z, logdet = model.decoder(yy, z_mask, g=g, reverse=False)
        e1.record(cur)
        loss = (z * r).sum() + logdet.sum()
        e1b = torch.cuda.Event(enable_timing=True)
        e1b.record(cur)
        loss.backward()
        from glow_tts_train.convops import flush_groups
        flush_groups()
        e2.record(cur)
        torch.cuda.synchronize()
        fwd.append(e0.elapsed_time(e1))
        bwd.append(e1b.elapsed_time(e2))
    fwd, bwd = sorted(fwd[1:]), sorted(bwd[1:])
    f_ms, b_ms = fwd[len(fwd) // 2], bwd[len(bwd) // 2]
    C, H = cfg.audio.mel_channels * mc.n_sqz, (mc.hidden_channels_dec or mc.hidden_channels)
    L, k, nb = mc.n_block_layers, mc.kernel_size_dec, mc.n_blocks_dec
    N = B * (T // mc.n_sqz)
    alg_bytes = 3 * (6.5 * C + 28 * H) * 4 * N * nb
    alg_flops = 3 * (2 * (C // 2) * H + L * 2 * H * 2 * H * k + (L - 1) * 2 * H * 2 * H + 2 * H * H + 2 * H * C) * N * nb
    t_roof_ms = 1e3 * max(alg_bytes / (HBM_PEAK_GBS * 1e9), alg_flops / (FP32_MFMA_PEAK_TFLOPS * 1e12))
    t = f_ms + b_ms
    # the arithmetic the step really runs in ("bf16x6": six bf16 products per fp32 product on the 2.5 PFLOP/s pipe) has a higher
    # ceiling than the fp32 MFMA SURVEY's roof is priced on: 2.5 PF / 6 = 416.7 TFLOP/s fp32-equivalent
    t_roof6_ms = 1e3 * max(alg_bytes / (HBM_PEAK_GBS * 1e9), 6.0 * alg_flops / (BF16_MFMA_PEAK_TFLOPS * 1e12))
    return {"fwd_ms": round(f_ms, 3), "bwd_ms": round(b_ms, 3), "fwd_bwd_ms": round(t, 3),
            "alg_GB": round(alg_bytes / 1e9, 3), "alg_TFLOP": round(alg_flops / 1e12, 4),
            "bound": "mfma (fp32)" if alg_flops / (FP32_MFMA_PEAK_TFLOPS * 1e12) > alg_bytes / (HBM_PEAK_GBS * 1e9) else "hbm",
            "roof_ms": round(t_roof_ms, 3), "frac": round(t_roof_ms / t, 4),
            "roof_ms_bf16x6_ceiling": round(t_roof6_ms, 3), "frac_vs_bf16x6_ceiling": round(t_roof6_ms / t, 4),
            "what_frac_is": "frac: SURVEY 8d(ii)'s roof (fp32 MFMA peak 157.3 TFLOP/s); frac_vs_bf16x6_ceiling: the ceiling of the "
                            "arithmetic the default step runs in (2.5 PFLOP/s bf16 pipe / 6 products = 416.7 TFLOP/s fp32-equivalent)",
            "mel_frames_per_s_decoder_only": round(B * T / (t * 1e-3))}


_T0 = time.perf_counter()


def usable_cores() -> int:
    """Cores this process may actually use: scheduler affinity capped by the cgroup CPU quota (os.cpu_count() reports
    the whole host, which over-subscribes the oracle's thread pool on a shared GPU box)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, 64))


def log(msg):
    """Progress line on stderr (the JSON line on stdout stays alone)."""
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def build_workload(args, dev, rank):
    """BASELINE.json configs[1] (or the sizes on the command line): random-init model + optimizer, one synthetic batch
    resident on `dev`, data-dependent ActNorm init done.  Returns (model, optimizer, batch, cfg)."""
    from glow_tts_train import config, models

    B, T_mel = args.batch, args.t_mel
    T_text = args.t_text or T_mel // 5
    cfg = config.TrainingConfig()
    cfg.model.num_symbols = 148
    cfg.model.n_blocks_dec = args.blocks
    if args.speakers > 0:
        cfg.model.n_speakers, cfg.model.gin_channels = args.speakers, args.gin
    torch.manual_seed(cfg.seed)
    model, opt = models.setup_model(cfg, use_cuda=True)
    # non-degenerate couplings for the benchmark: the zero-initialised end convs would make logs == 0 everywhere
    with torch.no_grad():
        for f in model.decoder.flows:
            if hasattr(f, "end"):
                f.end.weight.normal_(0, 0.01)
    model.train()
    log(f"model built: {sum(p.numel() for p in model.parameters())} parameters")

    gen = torch.Generator().manual_seed(cfg.seed + rank)
    x = torch.randint(1, 148, (B, T_text), generator=gen).to(dev)
    x_lengths = torch.full((B,), T_text, dtype=torch.long, device=dev)
    y = torch.randn(B, cfg.audio.mel_channels, T_mel, generator=gen).to(dev)
    y_lengths = torch.full((B,), T_mel, dtype=torch.long, device=dev)
    speaker_ids = (torch.arange(B) % args.speakers).to(dev) if args.speakers > 0 else None
    batch = (x, x_lengths, y, y_lengths, speaker_ids)

    # data-dependent ActNorm initialisation on the first batch (ddi.py:20-39), untimed; rank 0's result wins (Q10)
    for f in model.decoder.flows:
        if getattr(f, "set_ddi", False):
            f.set_ddi(True)
    with torch.no_grad():
        model(x, x_lengths, y, y_lengths, g=speaker_ids)
    torch.cuda.synchronize()
    log("data-dependent init forward done")
    return model, opt, batch, cfg


def other_configs_leg(args, dev, warm=6, steps=8):
    """BASELINE.json configs[2] (bf16 tensors in HBM, B=64, T_mel=1000) and configs[4] (speaker-conditioned couplings, B=48,
    T_mel=1200, 20 flow blocks, 4 speakers, gin 64): the same full training step, `warm` + `steps` steps each on fresh models,
    so that the driver's run times them too.  Never `value`."""
    import copy
    import gc

    from glow_tts_train.attentions import MultiHeadAttention
    from glow_tts_train.train import train_batch

    res = {}
    for key, over, bf16 in (("configs[2]", dict(batch=64, t_mel=1000, blocks=12, speakers=0), True),
                            ("configs[4]", dict(batch=48, t_mel=1200, blocks=20, speakers=4), False)):
        a2 = copy.copy(args)
        for k, v in over.items():
            setattr(a2, k, v)
        a2.t_text = 0
        try:
            model, opt, batch, cfg = build_workload(a2, dev, 0)
            if bf16:
                model.decoder.io_bf16 = "all"
                for m in model.modules():
                    if isinstance(m, MultiHeadAttention):
                        m.bf16_mma = True
            loss = None
            for _ in range(warm):
                loss = train_batch(model, opt, batch, cfg.grad_clip, None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = train_batch(model, opt, batch, cfg.grad_clip, None)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[key] = {"workload": f"B={a2.batch}, T_text={a2.t_mel // 5}, T_mel={a2.t_mel}, {a2.blocks} flow blocks"
                                    + (f", {a2.speakers} speakers (gin {a2.gin})" if a2.speakers else "")
                                    + (", bf16 tensors in HBM for the flow decoder + bf16-MFMA attention, fp32 accumulate" if bf16 else ", fp32"),
                        "dtype": "bf16 tensors / fp32 accumulate" if bf16 else "f32",
                        "ms_per_step": round(1e3 * dt / steps, 3), "value": round(a2.batch * a2.t_mel * steps / dt),
                        "unit": "mel-frames/s", "steps": steps, "warmup": warm, "final_loss": float(loss)}
            log(f"other_configs {key}: {res[key]['ms_per_step']:.2f} ms/step")
        except Exception as exc:                    # an orientation leg must never cost the run its result
            log(f"other_configs {key} failed ({type(exc).__name__}: {exc}); reported as null")
            res[key] = None
        finally:
            model = opt = batch = None
            gc.collect()
            torch.cuda.empty_cache()
    return res


def main():
    args = parse()
    # stdout carries ONE JSON line: libraries that print there (RCCL's version banner at communicator creation) are sent
    # to stderr for the duration of the run, and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    # one rank per GPU; BENCH_DIST_BACKEND=gloo with fewer GPUs than ranks is a rehearsal of the N>1 control flow on a
    # one-GPU box (ranks share the card, collectives staged by gloo) — never a measurement
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"{world} ranks but {torch.cuda.device_count()} GPUs visible")
    local %= max(1, torch.cuda.device_count())
    # N > 1: every rank (and the autograd thread it spawns later, which inherits the mask) gets its own share of the cores
    # this job may use — 8 ranks x (Python thread + backward thread) must not migrate over each other (host enqueue time is
    # within 2x of the step's GPU time: DESIGN.md 4f).  BENCH_NO_PIN=1 leaves the affinity alone.
    pinned = None
    if world > 1 and os.environ.get("BENCH_NO_PIN") != "1" and hasattr(os, "sched_setaffinity"):
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // world
        if per >= 1:
            pinned = cores[local * per:(local + 1) * per]
            os.sched_setaffinity(0, pinned)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend=backend, init_method="env://")
    elif args.rccl_self:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=backend, rank=0, world_size=1)

    from glow_tts_train import _hip, parallel
    from glow_tts_train.train import GraphedTrainStep, train_batch

    _hip.load()
    model, opt, batch, cfg = build_workload(args, dev, rank)
    B, T_mel = args.batch, args.t_mel
    T_text = args.t_text or T_mel // 5
    x, x_lengths, y, y_lengths, _ = batch
    reducer = parallel.FlowBlockReducer(model, opt, force=args.rccl_self, measure=True) if (world > 1 or args.rccl_self) else None
    if reducer is not None:
        reducer.broadcast_parameters(0)

    # single GPU: the step is captured into a hipGraph (hand-written + library kernels alike) and replayed; with a
    # process group the collectives stay outside a graph and the step is launched eagerly
    step_fn = lambda: train_batch(model, opt, batch, cfg.grad_clip, reducer)      # noqa: E731
    mode = "eager"
    if world == 1 and args.graph:
        try:
            graphed = GraphedTrainStep(model, opt, cfg.grad_clip, batch, warmup=2)
            step_fn = lambda: graphed()                                            # noqa: E731
            mode = "hipgraph"
        except Exception as exc:                                                   # capture unsupported: say so, run eagerly
            log(f"hipGraph capture failed ({type(exc).__name__}: {exc}); running eagerly")
    log(f"step mode: {mode}")
    for i in range(args.warmup):
        step_fn()
        torch.cuda.synchronize()
        log(f"warm-up step {i + 1}/{args.warmup} done")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    loss = None
    host_s = 0.0                                    # time the host spends queueing a step (no synchronisation inside step_fn)
    for _ in range(args.steps):
        h0 = time.perf_counter()
        loss = step_fn()
        host_s += time.perf_counter() - h0
    fence()
    dt = time.perf_counter() - t0
    host_ms = torch.tensor([1e3 * host_s / args.steps], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(host_ms, op=dist.ReduceOp.MAX)
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    loss_val = float(loss)
    log(f"timed region: {args.steps} steps in {dt:.3f} s, loss {loss_val:.4f}")
    comm = None
    if reducer is not None:
        # time the compute stream spent in reducer.finish() waiting for collectives backward did not hide (HIP events on
        # the compute stream), the timed steps only; max over ranks like the step time
        per_bucket = reducer.bucket_timings(last_steps=args.steps)       # this rank's (rank 0 prints its own)
        exposed = reducer.exposed_comm_ms()[-args.steps:]
        ex = torch.tensor([sum(exposed) / max(1, len(exposed))], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(ex, op=dist.ReduceOp.MAX)
        comm = {"backend": reducer.backend + (" (RCCL)" if reducer.backend == "nccl" else ""),
                "rccl_ranks": dist.get_world_size() if reducer.backend == "nccl" else 0,
                "buckets": len(reducer.buckets), "buckets_launched_during_backward": reducer.launched_in_backward,
                "grad_MB_per_step": round(4e-6 * sum(b.hi - b.lo for b in reducer.buckets), 1),
                "exposed_comm_ms_per_step": round(float(ex), 3),
                # rank 0's buckets in launch order: ready -> start = the collective waiting for work queued on the OTHER streams
                # (every bucket waits on all side streams), start -> end = queueing behind earlier collectives + the wire
                "per_bucket": per_bucket}
    frames = world * B * T_mel * args.steps
    ms_per_step = 1e3 * dt / args.steps

    sizes = (B, T_mel, args.blocks, args.speakers)
    which = {(32, 800, 12, 0): "BASELINE configs[1]", (48, 1200, 20, 4): "BASELINE configs[4]"}.get(
        sizes, "custom sizes (not a BASELINE configuration)")
    out = {
        "metric": "mel_frames_per_sec", "value": frames / dt, "unit": "mel-frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{which}: full training step, per-GPU B={B}, T_text={T_text}, T_mel={T_mel}, "
                               f"80 mels, {args.blocks} flow blocks, n_split=4, n_sqz=2, H=192, fp32, dropout 0.1/0.05, "
                               + (f"{args.speakers} speakers (gin {args.gin}), " if args.speakers > 0 else "")
                               + "random-init weights, synthetic resident batch",
                   "global_batch": world * B, "parallelism": f"dp{world}", "launch": mode, "final_loss": loss_val},
        # wall time the slowest rank's host needed to QUEUE a step (Python + C launch calls of forward, backward thread,
        # optimizer); when it approaches ms_per_step the step is host-bound.  It can exceed the GPU time only transiently.
        "host_enqueue_ms_per_step": round(float(host_ms), 3),
        "host_cores_per_rank": (len(pinned) if pinned is not None else None),
    }
    if comm is not None:
        out["comm"] = comm

    # ---- second arithmetic leg (not `value`).  The WN-stack convolutions have two fp32 forms: the native fp32 MFMA
    # (v_mfma_f32_16x16x4_f32) and "bf16x6+wrw" — each fp32 operand split EXACTLY into three bf16 planes, the six products
    # above 2^-24 formed on the bf16 matrix pipe, fp32 accumulation (csrc/convgemm_split.hip): fp32-equivalent results
    # (tests/test_conv_math.py: error against fp64 no larger than the native kernels'; every golden / oracle GPU test runs in
    # both forms at the same tolerances).  `value` is the package default (config.conv_math names it); this leg is the other.
    from glow_tts_train import convops

    default_math = convops.conv_math_name()
    other_math = "fp32" if default_math != "fp32" else "bf16x6+wrw"
    out["config"]["conv_math"] = default_math
    # N > 1 (the driver's SCALE runs): `value` only.  The legs below are one-GPU diagnostics that the N=1 line carries; with
    # several ranks they would run on rank 0 alone while the others tear their communicator down (BENCH_EXTRA_LEGS=1 forces them).
    extra_legs = world == 1 or os.environ.get("BENCH_EXTRA_LEGS") == "1"
    if not extra_legs:
        args.no_split_math = args.no_roofline = True
    if not args.no_split_math and mode == "eager":
        previous = convops.set_conv_math(other_math)
        key = "native_fp32" if other_math == "fp32" else "split_math"
        try:
            for _ in range(3):
                step_fn()
            fence()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                loss = step_fn()
            fence()
            dts = time.perf_counter() - t1
            if world > 1:
                tt = torch.tensor([dts], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dts = float(tt)
            out[key] = {"mode": other_math, "value": frames / dts, "unit": "mel-frames/s",
                        "ms_per_step": 1e3 * dts / args.steps, "loss_after_these_further_steps": float(loss),
                        "arithmetic": ("WN convolutions on the native fp32 MFMA (v_mfma_f32_16x16x4_f32)" if other_math == "fp32" else
                                       "WN convolutions (forward, backward-data, weight gradient): fp32 operands as 3 bf16 planes, "
                                       "6 products per pair on v_mfma_f32_16x16x32_bf16, fp32 accumulate") + "; everything else as in `value`"}
            log(f"{key} leg ({other_math}): {out[key]['ms_per_step']:.2f} ms/step")
        except Exception as exc:                    # the extra leg must never cost the run its result
            log(f"{key} leg failed ({type(exc).__name__}: {exc}); reported as null")
            out[key] = None
        finally:
            convops.set_conv_math(previous)         # the roofline pass below measures the default form again

    # ---- extra leg (not `value`): bf16 activation tensors in HBM for the flow decoder (decoder.io_bf16 = "all")
    if args.dtype == "bf16" and mode == "eager":
        from glow_tts_train.attentions import MultiHeadAttention
        dec = model.decoder
        dec.io_bf16 = "all"
        mhas = [m for m in model.modules() if isinstance(m, MultiHeadAttention)]
        for m in mhas:
            m.bf16_mma = True                          # encoder attention contractions on v_mfma_f32_16x16x16_bf16
        try:
            for _ in range(3):
                step_fn()
            fence()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                loss = step_fn()
            fence()
            dtb = time.perf_counter() - t1
            if world > 1:
                tt = torch.tensor([dtb], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dtb = float(tt)
            out["bf16_io"] = {"value": frames / dtb, "unit": "mel-frames/s", "ms_per_step": 1e3 * dtb / args.steps,
                              "dtype": "bf16 tensors / fp32 accumulate", "loss_after_these_further_steps": float(loss),
                              "what": "flow decoder: squeezed flow tensor and every hidden tensor of the coupling networks bf16 in "
                                      "HBM, v_mfma_f32_16x16x32_bf16 with fp32 accumulation, fp32 parameters / (m, logs) / "
                                      "log-determinants / parameter gradients; text encoder: attention contractions on "
                                      "v_mfma_f32_16x16x16_bf16 (fp32 softmax and tensors); losses as in `value`"}
            log(f"bf16-tensor leg: {out['bf16_io']['ms_per_step']:.2f} ms/step")
        except Exception as exc:
            log(f"bf16-tensor leg failed ({type(exc).__name__}: {exc}); reported as null")
            out["bf16_io"] = None
        finally:
            dec.io_bf16 = False
            for m in mhas:
                m.bf16_mma = False

    # ---- roofline leg: HIP events around every hand-written kernel launch, instrumented pass after the timed region
    if reducer is not None:
        reducer.remove_hooks()                           # the legs below run rank-local steps without collectives
        # ... on the SAME operator path as `value`: with the listener gone and a process group still alive, "auto" would send
        # every parameter gradient back through autograd (no whole-block executors) — pin in-place gradients instead
        convops.set_direct_grads(True)
    if rank == 0 and not args.no_roofline:
        import re

        C = cfg.audio.mel_channels * cfg.model.n_sqz
        H = cfg.model.hidden_channels_dec
        Ts = T_mel // cfg.model.n_sqz
        alg = algorithmic_bytes(B, C, H, Ts, T_text, opt._optim.numel_padded)
        n_inst = 3
        wino_before = _hip.wino_launches()
        _hip.enable_timing()
        for _ in range(n_inst):
            train_batch(model, opt, batch, cfg.grad_clip, None)
        times = _hip.disable_timing()
        wino_used = wino_before > 0                        # the timed steps ran the gated in-conv in its Winograd form (csrc/convwino.hip)
        log("instrumented pass done")
        dec = decoder_alone(model, batch, cfg, n_iter=5)
        log(f"decoder alone: fwd {dec['fwd_ms']:.2f} ms, bwd {dec['bwd_ms']:.2f} ms")
        hbm, mfma, other = {}, {}, {}
        for name, ms in times.items():
            mean_ms = sum(ms) / len(ms)
            row = {"launches_per_step": len(ms) // n_inst, "mean_us": round(1e3 * mean_ms, 2),
                   "total_ms_per_step": round(sum(ms) / n_inst, 3)}
            m = re.match(r"(glowtts_conv\w*)\[M(\d+) K(\d+)x(\d+) N(\d+)x(\d+)\]", name)
            if m:       # dense contraction on the MFMA: algorithmic (fp32) FLOPs = 2 * M * K * taps * columns
                M_, K_, taps_, cols_ = int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)) * int(m.group(6))
                flops = 2.0 * M_ * K_ * taps_ * cols_
                nbytes = conv_algorithmic_bytes(m.group(1), M_, K_, taps_, cols_, H)
                row.update(alg_GFLOP=round(flops / 1e9, 3), TFLOPs=round(flops / (mean_ms * 1e-3) / 1e12, 2),
                           alg_MB=round(nbytes / 1e6, 2), GBps=round(nbytes / (mean_ms * 1e-3) / 1e9, 1),
                           hbm_frac=round(nbytes / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
                mfma[name] = row
            elif name in alg:
                row.update(alg_MB=round(alg[name] / 1e6, 3), GBps=round(alg[name] / (mean_ms * 1e-3) / 1e9, 1))
                hbm[name] = row
            else:
                other[name] = row
        sub_ms = sum(hbm[k]["total_ms_per_step"] for k in INVERTIBLE_SUBSET if k in hbm)
        sub_bytes = sum(alg[k] * hbm[k]["launches_per_step"] for k in INVERTIBLE_SUBSET if k in hbm)
        survey_gb = 3 * 6.5 * C * 4 * B * Ts * cfg.model.n_blocks_dec / 1e9
        conv_ms = sum(v["total_ms_per_step"] for v in mfma.values())
        conv_flop = sum(v["alg_GFLOP"] * v["launches_per_step"] for v in mfma.values())

        def on_bf16_pipe(tag):
            """Does this launch run as a bf16-plane kernel in the mode being measured?  Mirrors the library's dispatch
            (csrc/convgemm.hip dispatch_convgemm, convgemm_split.hip dispatch_split_ns / conv_wrw_split_dispatch,
            convwrw_tr.hip conv_wrw_tr_dispatch): forward-type launches need planes bound to their weights (the WN stack; the
            encoder's FFN convolutions unless GLOWTTS_ENC_PLANES=0), weight gradients need nothing but the mode."""
            base, _, wrw = default_math.partition("+")
            if base != "bf16x6":
                return False
            m_ = re.match(r"(glowtts_conv\w*)\[M(\d+) K(\d+)x(\d+) N(\d+)x(\d+)", tag)
            name_, M_, K_, taps_ = m_.group(1), int(m_.group(2)), int(m_.group(3)), int(m_.group(4))
            B_, T_ = int(m_.group(5)), int(m_.group(6))
            if name_ in ("glowtts_conv_gate_fwd", "glowtts_conv_res_skip_fwd", "glowtts_conv_gate_bwd"):
                return True
            if name_ == "glowtts_conv_fwd":
                if taps_ == 5 and K_ == 2 * H and M_ == H:
                    return True                                     # backward-data of the WN stack's 5-tap convolution
                Fc = cfg.model.filter_channels
                if taps_ == 3 and {M_, K_} == {H, Fc} and os.environ.get("GLOWTTS_ENC_PLANES", "1") != "0":
                    big = M_ % 128 == 0 or M_ > 192                 # 128-row workgroups
                    wg5 = -(-T_ // 80) * B_ * -(-M_ // (128 if big else 64))
                    use32 = wg5 < 440 and -(-T_ // 32) * 32 * 10 <= T_ * 11
                    return not (use32 and big)                      # that one case stays on the fp32 MFMA kernel
                return False
            if name_ in ("glowtts_conv_wrw", "glowtts_conv_wrw2"):
                return wrw == "wrw" and (taps_ == 1 or (taps_ in (3, 5) and M_ % 32 == 0))
            return False

        def winograd(tag):
            """Does this launch run in the Winograd F(4, 5) form (csrc/convwino.hip)?  The gated 5-tap in-conv of the flow stack when
            the library launched that kernel during this process (the counter says so) and the shape is the kernel's: 8 products per
            (row, channel, 4 frames) instead of 20, i.e. 0.4 of the direct form's MFMAs for the same result."""
            m_ = re.match(r"glowtts_conv_gate_fwd\[M(\d+) K(\d+)x(\d+) N(\d+)x(\d+)", tag)
            return bool(m_ and wino_used and int(m_.group(3)) == 5 and int(m_.group(2)) % 64 == 0 and int(m_.group(5)) % 4 == 0
                        and default_math.startswith("bf16x6"))

        def pipe_mult(tag):                        # MFMA flops the pipe really does per algorithmic (fp32, direct-form) flop
            return (6.0 * 8.0 / 20.0 if winograd(tag) else 6.0) if on_bf16_pipe(tag) else 1.0

        # The instrumented pass drives every convolution through the per-operator path (no flow-stack arena, hence no Winograd-domain
        # planes): its row for the gated in-conv is the DIRECT kernel.  Where the timed step launches the Winograd form, the row is
        # re-priced with that kernel timed alone (the direct kernel's time stays beside it).
        wino_t = {}
        for tag, row in mfma.items():
            if winograd(tag):
                try:
                    wt = wino_gate_time(tag, dev)
                except Exception as exc:
                    wt = None
                    log(f"Winograd gate timing failed ({type(exc).__name__}: {exc})")
                if wt is not None:
                    wino_t[tag] = wt
                    row["direct_kernel_mean_us"] = row["mean_us"]
                    us = wt["us_per_launch"]
                    row.update(mean_us=us, total_ms_per_step=round(row["launches_per_step"] * us / 1e3, 3),
                               TFLOPs=round(row["alg_GFLOP"] / us * 1e3, 2), GBps=round(row["alg_MB"] / us * 1e3, 1),
                               hbm_frac=round(row["alg_MB"] / us * 1e3 / HBM_PEAK_GBS, 4), form=wt["note"])
        wino_used = bool(wino_t)
        for tag, row in mfma.items():              # which roof bounds each contraction, and how close it is to that roof
            split = on_bf16_pipe(tag)
            peak = BF16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
            t_mfma = pipe_mult(tag) * row["alg_GFLOP"] / peak            # us at the pipe's peak (GFLOP / TFLOP/s = ms*1e-3)
            t_hbm = row["alg_MB"] / HBM_PEAK_GBS                                    # MB / (GB/s) = ms*1e-3 likewise
            row.update(pipe=("bf16 MFMA x6, Winograd F(4,5): 0.4 of the direct form's products" if winograd(tag) else "bf16 MFMA x6") if split else "fp32 MFMA",
                       mfma_frac_algorithmic=round(row["TFLOPs"] / peak, 4),
                       mfma_frac_pipe=round(pipe_mult(tag) * row["TFLOPs"] / peak, 4),
                       bound="hbm" if t_hbm > t_mfma else "mfma",
                       frac_of_bound=round(max(t_hbm, t_mfma) * 1e3 / row["mean_us"], 4))
        # ---- which launch is the dominant one: arg-max of (launches per step x mean duration) over the table of launches THE STEP
        # MAKES.  The instrumented pass above times every convolution as a launch of its own; the step batches two families
        # (csrc/wn_stack.hip): a WN stack's four 5-tap weight gradients are ONE glowtts_conv_wrw_batch launch, a block's 1x1 weight
        # gradients ONE glowtts_conv_wrw1_multi launch.  Their rows are re-priced with the batched launch timed here (HIP events on
        # its stream, back to back), then the largest total wins — no kernel is named by hand.
        kd, nl = cfg.model.kernel_size_dec, cfg.model.n_blocks_dec * cfg.model.n_block_layers
        wrw5_tag = next((t for t in mfma if re.match(rf"glowtts_conv_wrw\[M{2 * H} K{H}x{kd} ", t)
                         and mfma[t]["launches_per_step"] == nl and on_bf16_pipe(t)), None)
        bt = None
        if wrw5_tag is not None:
            try:
                bt = batched_wrw_time(wrw5_tag, dev, n=cfg.model.n_block_layers)
            except Exception as exc:
                log(f"batched weight-gradient timing failed ({type(exc).__name__}: {exc})")
        as_launched = {}
        for tag, row in mfma.items():
            us = row["mean_us"]
            if tag == wrw5_tag and bt is not None:
                us = bt["us_per_problem"]

            as_launched[tag] = {"launches_per_step": row["launches_per_step"], "us_per_launch_or_problem": round(us, 2),
                                "total_ms_per_step": round(row["launches_per_step"] * us / 1e3, 3)}
        for tag, row in hbm.items():
            as_launched[tag] = {"launches_per_step": row["launches_per_step"], "us_per_launch_or_problem": row["mean_us"],
                                "total_ms_per_step": row["total_ms_per_step"]}
        dom = max(as_launched, key=lambda k: as_launched[k]["total_ms_per_step"]) if as_launched else None
        top5 = dict(sorted(as_launched.items(), key=lambda kv: -kv[1]["total_ms_per_step"])[:5])

        def mfma_roofline(tag, mean_us, problems=1):
            """Roofline object of one MFMA launch: `problems` problems of `tag`'s shape in `mean_us` microseconds."""
            row = mfma[tag]
            split = on_bf16_pipe(tag)
            peak = BF16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
            mult = pipe_mult(tag)
            tf = problems * row["alg_GFLOP"] * 1e9 / (mean_us * 1e-6) / 1e12          # fp32-equivalent TFLOP/s
            alg_mb = problems * row["alg_MB"]
            hbm_frac = alg_mb * 1e6 / (mean_us * 1e-6) / 1e9 / HBM_PEAK_GBS
            pe = pmc_entry(tag, default_math)
            r = {"bound": "mfma", "kernel": tag + (f" x{problems} (one launch, as the step makes it)" if problems > 1 else ""),
                 "achieved": round(mult * tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": mult * tf / peak,
                 "traffic": pe.get("traffic_bytes"), "pmc_source": pe.get("source"), "pmc_key": pe.get("pmc_key"),
                 "mfma_busy_measured": pe.get("mfma_util"),
                 "pipe": ("bf16 MFMA, Winograd F(4,5) form: 8 of the direct form's 20 products per 4 frames, six bf16 MFMAs each"
                          if winograd(tag) else "bf16 MFMA, 6 products per fp32 product (bf16x6)") if split else "fp32 MFMA",
                 "frac_algorithmic": round(tf / peak, 4), "frac_pipe": round(mult * tf / peak, 4),
                 "mean_us": round(mean_us, 2), "alg_GFLOP": round(problems * row["alg_GFLOP"], 3), "alg_MB": round(alg_mb, 2),
                 "hbm_frac_on_algorithmic_bytes": round(hbm_frac, 4),
                 "traffic_over_algorithmic": (round(pe["traffic_bytes"] / (alg_mb * 1e6), 3) if pe.get("traffic_bytes") else None),
                 "fp32_equivalent_TFLOPs": round(tf, 2),
                 "fp32_equivalent_vs_fp32_mfma_peak": round(tf / FP32_MFMA_PEAK_TFLOPS, 4)}
            if split:
                r["roof_TFLOPs_fp32_equivalent"] = round(BF16_MFMA_PEAK_TFLOPS / 6.0, 1)
                r["what_frac_is"] = ("frac = frac_pipe: the six bf16 MFMAs the kernel issues per fp32 product, counted as work, over "
                                     "the dense bf16 peak; frac_algorithmic counts the fp32 product once")
            if row.get("bound") == "hbm":               # a byte-bound contraction (1x1): report it against the HBM roof
                r.update(bound="hbm", achieved=round(alg_mb * 1e6 / (mean_us * 1e-6) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                         frac=hbm_frac)
            return r

        if dom is not None and dom in mfma:
            if dom == wrw5_tag and bt is not None:
                n_p = bt["problems_per_launch"]
                out["roofline"] = mfma_roofline(dom, bt["us_per_problem"] * n_p, n_p)
                out["roofline"]["single_launch"] = {k: v for k, v in mfma_roofline(dom, mfma[dom]["mean_us"]).items()
                                                    if k in ("mean_us", "frac_pipe", "frac_algorithmic", "fp32_equivalent_TFLOPs")}
            else:
                out["roofline"] = mfma_roofline(dom, mfma[dom]["mean_us"])
                if re.match(r"glowtts_conv_(gate_fwd|res_skip_fwd)", dom):
                    out["roofline"]["as_launched_note"] = (
                        "the timed step launches the decoder's forward as two half-batch chains on two streams (B/2 per launch, "
                        "DESIGN.md lesson 37); this is the whole-batch launch, the quantity the instrumented pass and the "
                        "one-stream rocprofv3 passes (profiles/) both time")
        elif dom is not None:
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": hbm[dom]["GBps"], "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": hbm[dom]["GBps"] / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, default_math)}
        else:
            out["roofline"] = {"bound": None, "kernel": None, "frac": None, "traffic": None}
        out["roofline"]["dominant_by"] = ("arg-max of launches_per_step x mean duration over the launches the step makes "
                                          "(`as_launched_top5`; batched families re-priced with the batched launch)")
        out["roofline"]["as_launched_top5"] = top5
        if wrw5_tag is not None and bt is not None:       # the batched 5-tap weight gradient stays on the line whichever kernel leads
            n_p = bt["problems_per_launch"]
            out["roofline"]["wrw5_batch_as_launched_in_the_step"] = dict(
                mfma_roofline(wrw5_tag, bt["us_per_problem"] * n_p, n_p), us_per_problem=bt["us_per_problem"], note=bt["note"])
        if default_math.endswith("+wrw"):
            try:        # the flow block's six 1x1 weight gradients: one multi-problem launch (csrc/convwrw1.hip)
                out["roofline"]["wrw1_multi_as_launched_in_the_step"] = wrw1_multi_time(args, dev)
            except Exception as exc:
                log(f"multi-problem 1x1 weight-gradient timing failed ({type(exc).__name__}: {exc})")
        try:        # coupling(k) + ActNorm + InvConv(k + 1): the two kernels the step's flow stack launches between blocks
            fused = fused_flows_time(args, dev)
        except Exception as exc:
            fused = None
            log(f"fused flow timing failed ({type(exc).__name__}: {exc})")
        if fused:
            # the subset as the step launches it, all by HIP events: the boundary launch (forward) and the fused backward kernel
            # between blocks (timed just above) plus the first block's ActNorm + InvConv and the last block's coupling on the
            # un-fused kernels (instrumented pass of this run).  The forward launch also carries two 1x1 contractions, so
            # `survey_frac_lower_bound` under-states the subset's own fraction.
            ends_k = ("glowtts_actnorm_invconv_fwd", "glowtts_actnorm_invconv_bwd", "glowtts_coupling_fwd", "glowtts_coupling_bwd")
            ends = sum(hbm[k]["mean_us"] for k in ends_k if k in hbm)
            ms_l = fused["ms_per_step_fused_part"] + ends / 1e3
            gb_l = fused["alg_GB_fused_part"] + sum(alg[k] for k in ends_k if k in hbm) / 1e9
            fused.update({"ms_per_step": round(ms_l, 3), "alg_GB_as_launched": round(gb_l, 3),
                          "frac": round(gb_l / (ms_l * 1e-3) / HBM_PEAK_GBS, 4),
                          "survey_frac_lower_bound": round(survey_gb / (ms_l * 1e-3) / HBM_PEAK_GBS, 4)})
        out["roofline"].update({
            "conv_math": default_math,
            "mfma_contractions": {"ms_per_step": round(conv_ms, 3), "alg_TFLOP_per_step": round(conv_flop / 1e3, 3),
                                  "TFLOPs": round(conv_flop / conv_ms, 2) if conv_ms else None,
                                  "vs_fp32_mfma_peak": round(conv_flop / conv_ms / FP32_MFMA_PEAK_TFLOPS, 4) if conv_ms else None},
            # leads with the bytes the fused kernels actually have to move (`frac`); SURVEY's un-fused numerator second
            "invertible_subset": {"ms_per_step": round(sub_ms, 3), "alg_GB": round(sub_bytes / 1e9, 3),
                                  "GBps": round(sub_bytes / (sub_ms * 1e-3) / 1e9, 1) if sub_ms else None,
                                  "frac": round(sub_bytes / (sub_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if sub_ms else None,
                                  # the un-fused per-flow byte count (6.5 C e per column per block,
                                  # x3 for fwd+bwd) over the measured subset time -- fusing flows lowers the time only
                                  "survey_alg_GB": round(survey_gb, 3),
                                  "survey_frac": round(survey_gb / (sub_ms * 1e-3) / HBM_PEAK_GBS, 4) if sub_ms else None,
                                  # HIP events around a 6-15 us launch include its launch latency (~4 us); the committed
                                  # rocprofv3 kernel trace of this same command times the kernels themselves
                                  "rocprofv3": subset_from_kernel_trace(4.0 * C * B * Ts, survey_gb, args.blocks),
                                  "what_these_are": "ms_per_step / frac / survey_frac: the UN-FUSED kernels of the instrumented pass "
                                                    "(one launch per flow); `as_launched_in_the_step`: what the timed step launches",
                                  "as_launched_in_the_step": fused},
            "step_ms": round(ms_per_step, 3),
            "decoder": dec,
            "mfma_kernels": dict(sorted(mfma.items(), key=lambda kv: -kv[1]["total_ms_per_step"])),
            "hbm_kernels": dict(sorted(hbm.items(), key=lambda kv: -kv[1]["total_ms_per_step"])),
            "other_kernels": dict(sorted(other.items(), key=lambda kv: -kv[1]["total_ms_per_step"])),
        })

    # ---- the other single-GPU BASELINE configurations, a few steps each (orientation beside `value`, timed by the same driver run)
    if rank == 0 and world == 1 and mode == "eager" and not args.no_other_configs and reducer is None and which == "BASELINE configs[1]":
        out["other_configs"] = other_configs_leg(args, dev)

    # ---- CPU baseline leg: the oracle (a port of the reference path) on this host's cores, bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import glow_oracle as O

        cores = usable_cores()
        hp = O.HParams(n_vocab=148, n_blocks_dec=args.blocks, n_speakers=args.speakers,
                       gin_channels=args.gin if args.speakers > 0 else 0)
        sd = {k: v.requires_grad_(True) for k, v in O.init_state_dict(hp, seed=cfg.seed).items()}
        oopt = O.AdamNoam(sd, dim_model=hp.hidden_channels)
        cb = tuple(None if t is None else t.cpu() for t in batch)
        model_name = "unknown"
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    model_name = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass

        def cpu_run(threads, warm, steps):
            torch.set_num_threads(threads)
            for _ in range(warm):                                        # thread pools, allocator, first-touch
                O.train_step(sd, hp, oopt, cb, cfg.grad_clip)
            ts = []
            for _ in range(steps):
                t0 = time.perf_counter()
                O.train_step(sd, hp, oopt, cb, cfg.grad_clip)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            return ts[len(ts) // 2], ts

        log(f"cpu baseline: oracle on {cores} threads ({model_name})")
        med, ts = cpu_run(cores, args.cpu_warmup, args.cpu_steps)
        log(f"cpu baseline: median {med:.2f} s/step over {len(ts)} steps")
        out["cpu_baseline"] = {"value": B * T_mel / med, "unit": "mel-frames/s", "cores": cores, "kind": "port",
                               "cpu_model": model_name, "threads": [cores],
                               "s_per_step_median": round(med, 3), "s_per_step_min_max": [round(ts[0], 3), round(ts[-1], 3)],
                               "sample": f"{args.cpu_steps} full training steps (median) after {args.cpu_warmup} warm-up steps of the "
                                         f"same B={B}, T_mel={T_mel}, {args.blocks}-block batch (oracle/glow_oracle.py, torch CPU "
                                         f"fp32 + C MAS) on {cores} threads of {model_name}"}
        if args.cpu_steps_1thread > 0:
            med1, ts1 = cpu_run(1, 1, args.cpu_steps_1thread)
            out["cpu_baseline"]["threads"] = [cores, 1]
            out["cpu_baseline"]["one_thread"] = {"value": B * T_mel / med1, "unit": "mel-frames/s", "cores": 1,
                                                 "s_per_step_median": round(med1, 3),
                                                 "sample": f"{args.cpu_steps_1thread} steps after 1 warm-up step, same batch, "
                                                           "torch.set_num_threads(1)"}
            log(f"cpu baseline, 1 thread: median {med1:.2f} s/step")

    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()                     # all ranks leave together: nobody tears RCCL down under a peer still working
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
