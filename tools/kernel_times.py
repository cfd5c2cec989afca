#!/usr/bin/env python3
"""Print per-kernel launch times from bench.py's JSON line (roofline tables): python tools/kernel_times.py file.json [substr]"""
import json
import sys

d = json.load(open(sys.argv[1]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
r = d["roofline"]
print(f"step {d['ms_per_step']:.2f} ms   decoder {r['decoder']['fwd_bwd_ms']} ms")
for grp in ("mfma_kernels", "hbm_kernels", "other_kernels"):
    for k, v in r[grp].items():
        if sub in k and v["total_ms_per_step"] >= 0.08:
            print(f"  {k:52s} {v['launches_per_step']:4d} x {v['mean_us']:7.2f} us = {v['total_ms_per_step']:6.3f} ms")
