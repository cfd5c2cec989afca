#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.x, rocpd sqlite output) run into the CSVs kept under profiles/.

  python tools/rocpd_summary.py kernels  <results.db> <out.csv>        per-kernel calls / total / mean / min / max (ns), % of GPU time
  python tools/rocpd_summary.py counters <results.db> <out.csv> [substr]  per-kernel mean counter value per dispatch (PMC runs)

rocprofv3 --kernel-trace --stats writes the same numbers into its `top_kernels` view; the view is used when present."""
import csv
import sqlite3
import sys


def kernels(db, out):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
        "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for r in rows:
            w.writerow([r[0][:160], r[1], r[2], round(r[3], 1), r[4], r[5], round(100.0 * r[2] / tot, 3)])
    print(f"{out}: {len(rows)} kernels, {tot / 1e6:.2f} ms of kernel time")


def counters(db, out, substr=""):
    """Per (kernel, grid size): one kernel name covers several shapes (a conv template runs at the decoder's and at the
    encoder's sizes), and a mean over them describes none; the grid size separates them.  Two shapes can still share a grid
    size (round 4: a WN stack's four 5-tap weight gradients in one launch and a pre-net layer's single one are both 216
    workgroups): when the dispatches of one (kernel, grid) fall into two duration classes more than 2.5x apart, the long class is
    reported as grid `<n>#long`."""
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    grid = "grid_size" if "grid_size" in cols else "0"
    raw = c.execute(
        f"select {name_col}, {grid}, counter_name, value, end - start from counters_collection where {name_col} like ?",
        (f"%{substr}%",)).fetchall()
    groups = {}
    for name, g, cn, val, dur in raw:
        groups.setdefault((name, g), []).append((cn, val, dur))
    rows = []
    for (name, g), items in sorted(groups.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        durs = [d for _, _, d in items]
        lo, hi = min(durs), max(durs)
        cut = (lo * hi) ** 0.5 if lo > 0 and hi > 2.5 * lo else None
        if cut is not None:                                  # two populations only if both sides are well populated
            n_long = sum(d > cut for d in durs)
            if n_long < 0.1 * len(durs) or n_long > 0.9 * len(durs):
                cut = None
        classes = {}
        for cn, val, dur in items:
            label = f"{g}#long" if (cut is not None and dur > cut) else str(g)
            classes.setdefault((label, cn), []).append((val, dur))
        for (label, cn), vs in sorted(classes.items()):
            vals = [v for v, _ in vs]
            rows.append((name[:160], label, cn, len(vs), sum(vals) / len(vals), min(vals), max(vals), sum(d for _, d in vs) / len(vs)))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "GridSize", "Counter", "Dispatches", "MeanPerDispatch", "Min", "Max", "MeanDurationNs"])
        w.writerows(rows)
    print(f"{out}: {len(rows)} rows")


if __name__ == "__main__":
    mode, db, out = sys.argv[1:4]
    if mode == "kernels":
        kernels(db, out)
    else:
        counters(db, out, sys.argv[4] if len(sys.argv) > 4 else "")
