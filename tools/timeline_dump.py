#!/usr/bin/env python3
"""Kernels of one step in a time window, with queue ids and gaps (companion of tools/timeline.py).
Usage: python tools/timeline_dump.py kernel_trace.csv from_us to_us [step_index_from_end=1]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
lo, hi = float(sys.argv[2]), float(sys.argv[3])
back = int(sys.argv[4]) if len(sys.argv) > 4 else 1
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows))
marks = [e for e in ev if "adam_kernel" in e[3]]
t0, t1 = marks[-back - 1][1], marks[-back][1]
print(f"step of {(t1 - t0) / 1e3:.1f} us")
prev_end = None
for s, e, q, n in ev:
    a = (s - t0) / 1e3
    if a < lo or a > hi or s < t0 - 2_000_000:
        continue
    n = re.sub(r"^void ", "", n).replace("glowtts::", "").replace("at::native::", "")[:90]
    gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:7.1f}"
    print(f"+{a:9.1f} q{q} {(e - s) / 1e3:7.1f} us  (gap {gap})  {n}")
    prev_end = max(prev_end or e, e)
