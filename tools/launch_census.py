#!/usr/bin/env python3
"""Launches per training step by kernel family, as the DIFFERENCE of two kernel-trace runs of the same command with different
step counts (so that model construction, the data-dependent init and warm-up do not count):
  python tools/launch_census.py <stats_a.csv> <steps_a> <stats_b.csv> <steps_b>
(stats CSVs from tools/rocpd_summary.py kernels)"""
import csv
import sys


def load(path):
    return {r["Name"]: (int(r["Calls"]), int(r["TotalDurationNs"])) for r in csv.DictReader(open(path))}


a, na, b, nb = load(sys.argv[1]), float(sys.argv[2]), load(sys.argv[3]), float(sys.argv[4])
d = nb - na
rows = []
for name in set(a) | set(b):
    ca, ta = a.get(name, (0, 0))
    cb, tb = b.get(name, (0, 0))
    if cb != ca:
        rows.append(((cb - ca) / d, (tb - ta) / d / 1e6, name))
ours = [r for r in rows if "glowtts::" in r[2]]
other = [r for r in rows if "glowtts::" not in r[2]]
print(f"hand-written launches / step: {sum(r[0] for r in ours):.1f}  ({sum(r[1] for r in ours):.3f} ms of kernel time / step)")
print(f"other launches / step: {sum(r[0] for r in other):.1f}  ({sum(r[1] for r in other):.3f} ms / step)")
for n, ms, name in sorted(other, reverse=True)[:40]:
    print(f"  {n:7.1f}  {ms:7.3f} ms  {name[:120]}")
