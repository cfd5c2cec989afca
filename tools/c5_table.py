#!/usr/bin/env python3
"""the per-kernel table of BASELINE configs[4] (B=48, T_mel=1200 -> N = 48 x 600 columns, 20 flow blocks, speaker
conditioning) from the committed rocprofv3 capture (profiles/r05_c5_*): launches per step, mean duration, and — for the kernels
whose shape is known from their template arguments — the fraction of the roof that bounds them.

  python tools/c5_table.py [tag]          # tag = r05_c5
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05_c5"
P = os.path.join(ROOT, "profiles")
B, Ts, H, C, NB, L = 48, 600, 192, 160, 20, 4
N = B * Ts
PEAK_BF16, PEAK_HBM = 2500.0, 8000.0      # TFLOP/s dense, GB/s (MI355X_MICROARCH.md)


def stats(name):
    rows = {}
    for r in csv.DictReader(open(os.path.join(P, f"{tag}_{name}.csv"))):
        rows[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return rows


s10, s5 = stats("kernel_stats"), stats("kernel_stats_5steps")
pmc = json.load(open(os.path.join(P, f"{tag}_pmc.json")))


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("glowtts::", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    return n.replace(", ", ",")


def pmc_rows(k):
    return [(key, v) for key, v in pmc.items() if key.split(" grid=")[0] == k]


# (M, K, taps, problems per launch, what) for the WN-stack kernels at this shape; flops are the six bf16 products per fp32 product
SHAPES = {
    "convgemm_split_kernel<3,2,4,1,5,0,3>": (384, 192, 5, 1, "gated in-conv (layers.py:146 + utils.py:31-38)"),
    "convgemm_split_kernel<3,1,4,4,5,0,3>": (192, 384, 5, 1, "5-tap backward-data + residual"),
    "convgemm_split_kernel<3,1,4,0,5,0,3>": (192, 384, 5, 1, "5-tap backward-data, first layer"),
    "convgemm_split_kernel<3,1,4,5,1,0,3>": (192, 384, 1, 1, "gate backward (1x1 backward-data of res/skip + gate')"),
    "convgemm_split_kernel<3,2,4,2,1,0,3>": (384, 192, 1, 1, "res/skip 1x1 (layers.py:155-161)"),
    "convgemm_split_kernel<3,1,4,3,1,0,3>": (192, 192, 1, 1, "last layer's skip 1x1"),
    "convwrw_tr_kernel<3,5,4,false,2>": (384, 192, 5, 4, "5-tap weight gradients, four per launch"),
    "wino_gate_fwd_kernel<0>": (384, 192, 5, 1, "gated in-conv, Winograd F(4,5) form (0.4 of the direct form's MFMAs)"),
}
out = []
tot = 0.0
for name, (c10, t10) in sorted(s10.items(), key=lambda kv: -kv[1][1]):
    c5, t5 = s5.get(name, (0, 0.0))
    per_step = (c10 - c5) / 5.0
    ms_step = (t10 - t5) / 5.0 / 1e6
    tot += ms_step
    if ms_step < 0.15:
        continue
    k = short(name)
    mean_us = t10 / c10 / 1e3
    cell = ""
    if k in SHAPES:
        M, K, taps, npl, what = SHAPES[k]
        gf = 2.0 * M * K * taps * N * 6 * npl / 1e9
        us = (t10 - t5) / max(c10 - c5, 1) / 1e3
        busy = [v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(v.get("SQ_BUSY_CYCLES", 1), 1) for _, v in pmc_rows(k)]
        tr = [v.get("traffic_bytes") for _, v in pmc_rows(k) if v.get("traffic_bytes")]
        tf = gf / us * 1e3
        if k.startswith("wino"):
            cell = (f"{what}: {0.4 * tf:.0f} TFLOP/s issued on the bf16 pipe = **{0.4 * tf / PEAK_BF16:.2f}** "
                    f"({tf:.0f} = {tf / PEAK_BF16:.2f} counted as the direct form's products)")
        else:
            cell = f"{what}: {tf:.0f} TFLOP/s on the bf16 pipe = **{tf / PEAK_BF16:.2f}**"
        if tr:
            cell += f"; HBM traffic {max(tr) / 1e6:.0f} MB per launch"
    else:
        tr = [(v.get("traffic_bytes"), v.get("duration_us_under_pmc")) for _, v in pmc_rows(k) if v.get("traffic_bytes")]
        if tr:
            t, d = max(tr)
            cell = f"{t / 1e6:.1f} MB per launch (PMC) -> {t / (mean_us * 1e-6) / 1e9 / PEAK_HBM:.2f} of HBM"
    out.append(f"| `{k}` | {per_step:.0f} | {mean_us:.1f} | {ms_step:.2f} | {cell} |")
print(f"single-stream kernel time per step: {tot:.1f} ms")
print("| kernel | launches / step | mean µs | ms / step | bound and fraction |")
print("|---|---|---|---|---|")
print("\n".join(out))
