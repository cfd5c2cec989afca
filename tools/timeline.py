#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 kernel trace (…_kernel_trace.csv): per-queue busy time, the union of busy
intervals, and the idle gaps of the whole device with the kernels either side.  Steps are delimited by the optimizer kernel.
Usage: python tools/timeline.py kernel_trace.csv [step_index_from_end=1] [min_gap_us=15]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows))
marks = [e for e in ev if "adam_kernel" in e[3]]
assert len(marks) > back, "not enough steps in the trace"
t0, t1 = marks[-back - 1][1], marks[-back][1]
step = [e for e in ev if t0 <= e[0] < t1]
print(f"step of {(t1 - t0) / 1e3:.1f} us, {len(step)} kernels")


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"glowtts::", "", n)
    return n[:70]


byq = defaultdict(list)
for s, e, q, n in step:
    byq[q].append((s, e, n))
for q, v in sorted(byq.items()):
    busy = sum(e - s for s, e, _ in v)
    print(f"  queue {q}: {len(v):4d} kernels, busy {busy / 1e3:8.1f} us, from +{(v[0][0] - t0) / 1e3:8.1f} to +{(v[-1][1] - t0) / 1e3:8.1f}")
# union + gaps
cur_end, union, gaps, last_name = t0, 0, [], "(previous step's optimizer)"
for s, e, q, n in step:
    if s > cur_end:
        gaps.append((s - cur_end, cur_end - t0, last_name, n))
        union += e - s
        cur_end, last_name = e, n
    elif e > cur_end:
        union += e - cur_end
        cur_end, last_name = e, n
print(f"  device busy (union) {union / 1e3:.1f} us, idle {(t1 - t0 - union) / 1e3:.1f} us in {len(gaps)} gaps")
for g, at, a, b in sorted(gaps, reverse=True)[:25]:
    if g / 1e3 >= min_gap:
        print(f"    gap {g / 1e3:7.1f} us at +{at / 1e3:8.1f}: after {short(a)}  -> before {short(b)}")
small = sum(g for g, *_ in gaps if g / 1e3 < min_gap)
print(f"  gaps < {min_gap:.0f} us: {sum(1 for g, *_ in gaps if g / 1e3 < min_gap)} totalling {small / 1e3:.1f} us")
# concurrency profile: time with exactly k queues busy
pts = []
for s, e, q, n in step:
    pts += [(s, 1), (e, -1)]
pts.sort()
depth, last, hist = 0, t0, defaultdict(int)
for t, d in pts:
    hist[depth] += t - last
    depth, last = depth + d, t
print("  time by number of kernels in flight: " + ", ".join(f"{k}: {v / 1e3:.0f} us" for k, v in sorted(hist.items())))
