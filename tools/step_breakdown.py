#!/usr/bin/env python3
"""Per-step breakdown of a rocprofv3 --kernel-trace CSV of bench.py: last steady-state step, by kernel and by origin."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = glob.glob(f"{d}/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
step = rows[adam[-2] + 1: adam[-1] + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0])
cat = collections.defaultdict(lambda: [0, 0])


def origin(n):
    if "glowtts::conv" in n:
        return "glowtts conv MFMA"
    if "glowtts::attn" in n:
        return "glowtts attention"
    if "glowtts" in n:
        return "glowtts other"
    if n.startswith("Cijk"):
        return "rocBLAS"
    if any(k in n for k in ("igemm", "miopen", "naive_conv", "batched_transpose", "SubTensor", "Im2d", "Col2Im", "ck::", "kernel_grouped", "kernel_batched")):
        return "MIOpen/CK"
    if "at::native" in n:
        return "torch aten"
    return "other"


for r in step:
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg[r["Kernel_Name"][:130]]
    a[0] += 1
    a[1] += dur
    c = cat[origin(r["Kernel_Name"])]
    c[0] += 1
    c[1] += dur
print(f"last step: wall {(t1 - t0) / 1e6:.2f} ms, {len(step)} launches, busy {sum(v[1] for v in agg.values()) / 1e6:.2f} ms")
for k, v in sorted(cat.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:22s} n={v[0]:5d} {v[1] / 1e6:7.2f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k:130s} n={v[0]:4d} {v[1] / 1e3:8.1f} us  avg {v[1] / v[0] / 1e3:7.1f}")
