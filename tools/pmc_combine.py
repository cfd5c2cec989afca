#!/usr/bin/env python3
"""Combine the per-counter CSVs of tools/pmc_passes.sh into one JSON per kernel and shape (profiles/<tag>_pmc.json):

  traffic_bytes = 2 * FETCH_SIZE + WRITE_SIZE   (KiB -> bytes; FETCH_SIZE doubled: gfx950 tallies 128-B read requests at
                  64 B, MI355X_MICROARCH.md "HBM"; WRITE_SIZE as read)
  mfma_util     = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): busy cycles are summed over the
                  chip's SIMDs, GUI_ACTIVE over its 8 XCDs (checked on the gated conv: 4 608 000 MFMAs x 32 cycles =
                  147 456 000 busy cycles exactly)

  python tools/pmc_combine.py gpurun_out/r02a profiles/r02_pmc.json"""
import csv
import json
import re
import sys

prefix, out = sys.argv[1:3]
table = {}


def short(name):
    m = re.search(r"glowtts::(\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else None


for group in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
    try:
        rows = list(csv.DictReader(open(f"{prefix}_pmc_{group}.csv")))
    except FileNotFoundError:
        continue
    for r in rows:
        k = short(r["Kernel"])
        if k is None:
            continue
        e = table.setdefault(f"{k} grid={r.get('GridSize', '?')}", {"dispatches": int(r["Dispatches"])})
        e[r["Counter"]] = round(float(r["MeanPerDispatch"]), 2)
        if r["Counter"] in ("FETCH_SIZE", "GRBM_GUI_ACTIVE") and r.get("MeanDurationNs"):
            e.setdefault("duration_us_under_pmc", round(float(r["MeanDurationNs"]) / 1e3, 2))
for k, e in table.items():
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["traffic_bytes"] = int(1024 * (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]))
    if e.get("SQ_VALU_MFMA_BUSY_CYCLES") and e.get("GRBM_GUI_ACTIVE"):
        e["mfma_util"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
json.dump(table, open(out, "w"), indent=1, sort_keys=True)
print(f"{out}: {len(table)} kernel shapes")
