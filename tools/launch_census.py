#!/usr/bin/env python3
"""Launches per training step by kernel family, from a kernel-stats CSV of tools/rocpd_summary.py.
  python tools/launch_census.py <kernel_stats.csv> <steps-in-the-profiled-run>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
ours = sum(int(r["Calls"]) for r in rows if "glowtts::" in r["Name"]) / steps
other = [(int(r["Calls"]) / steps, int(r["TotalDurationNs"]) / steps / 1e6, r["Name"][:110]) for r in rows if "glowtts::" not in r["Name"]]
print(f"hand-written launches / step: {ours:.0f}")
print(f"other launches / step: {sum(o[0] for o in other):.0f}  ({sum(o[1] for o in other):.3f} ms / step)")
for n, ms, name in sorted(other, reverse=True)[:40]:
    print(f"  {n:7.1f}  {ms:7.3f} ms  {name}")
