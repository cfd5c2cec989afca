#!/usr/bin/env python3
"""Timeline summary of the last training step in a rocprofv3 (rocpd sqlite) kernel trace of bench.py: per-queue busy time,
time with >= 1 kernel running, idle gaps, and the kernels that run while nothing else does (the critical path's makeup).
Usage: python tools/timeline.py <results.db> [top]"""
import collections
import sqlite3
import sys

db = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
step = rows[adam[-2] + 1: adam[-1] + 1]
t0, t1 = step[0][1], max(r[2] for r in step)
print(f"last step: {len(step)} launches, wall {(t1 - t0) / 1e6:.3f} ms")
per_q = collections.defaultdict(int)
for n, s, e, q in step:
    per_q[q] += e - s
for q, v in per_q.items():
    print(f"  queue {q}: busy {v / 1e6:.3f} ms, {sum(1 for r in step if r[3] == q)} launches")
for q in per_q:
    agg = collections.defaultdict(lambda: [0, 0])
    for n, s_, e, qq in step:
        if qq == q:
            agg[n[:100]][0] += 1
            agg[n[:100]][1] += e - s_
    print(f"queue {q}: kernels by total time")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"  {v[1] / 1e6:7.3f} ms  n={v[0]:4d}  avg {v[1] / v[0] / 1e3:7.1f} us  {k}")
# sweep: coverage and exclusive time per kernel
events = []
for i, (n, s, e, q) in enumerate(step):
    events.append((s, 1, i))
    events.append((e, -1, i))
events.sort()
active = set()
last = t0
covered = 0
excl = collections.defaultdict(int)       # time during which exactly one kernel runs, by kernel name
shared = 0
for t, kind, i in events:
    if active:
        covered += t - last
        if len(active) == 1:
            excl[step[next(iter(active))][0][:100]] += t - last
        else:
            shared += t - last
    last = t
    if kind == 1:
        active.add(i)
    else:
        active.discard(i)
print(f"  >=1 kernel running {covered / 1e6:.3f} ms, idle {(t1 - t0 - covered) / 1e6:.3f} ms, >=2 kernels running {shared / 1e6:.3f} ms")
print("time with exactly ONE kernel running, by kernel:")
for k, v in sorted(excl.items(), key=lambda kv: -kv[1])[:top]:
    print(f"  {v / 1e6:7.3f} ms  {k}")
