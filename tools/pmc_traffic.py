#!/usr/bin/env python3
"""HBM-side bytes per launch from two rocprofv3 PMC passes (one counter each, --kernel-trace only):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out/fetch -o run -- python bench.py ... --serial-tasks --no-graph
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out/write -o run -- python bench.py ... --serial-tasks --no-graph
    python tools/pmc_traffic.py out/fetch/run_results.db out/write/run_results.db --dtype f16 > profiles/rNN_pmc_traffic.json

Counter unit: KB.  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (wide coalesced reads are
tallied at half their size).  Kernels are grouped by a short name (template arguments dropped)."""
import argparse
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd._lib import csrc_sha16  # noqa: E402


def per_kernel(db, counter):
    c = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else None
    if view is None:
        sys.exit(f"{db}: no counters_collection view (tables: {tabs[:12]}...)")
    cols = [d[1] for d in c.execute(f"pragma table_info({view})")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    out = {}
    for name, cname, val in c.execute(f"select {name_col}, counter_name, value from {view}"):
        if cname != counter:
            continue
        out.setdefault(name, []).append(float(val))
    return out


def short(name):
    m = re.search(r"m3::?(\w+?)_kernel", name) or re.search(r"N2m3\d+(\w+?)_kernel", name)
    return m.group(1) if m else None


ap = argparse.ArgumentParser()
ap.add_argument("fetch_db")
ap.add_argument("write_db")
ap.add_argument("--dtype", default="f16")
ap.add_argument("--note", default="")
ap.add_argument("--head", default="", help="commit the passes were taken at (bench.py prints it in roofline.traffic_source)")
ap.add_argument("--merge", action="append", default=[], help="NEW=a,b : launch-weighted average of kernels a and b")
a = ap.parse_args()
fe, wr = per_kernel(a.fetch_db, "FETCH_SIZE"), per_kernel(a.write_db, "WRITE_SIZE")
agg = {}
for src, key in ((fe, "f"), (wr, "w")):
    for name, vals in src.items():
        s = short(name)
        if s is None:
            continue
        d = agg.setdefault(s, {"f": [], "w": []})
        d[key] += vals
kernels = {}
for s, d in sorted(agg.items()):
    if not d["f"] or not d["w"]:
        continue
    f = 1024.0 * sum(d["f"]) / len(d["f"])
    w = 1024.0 * sum(d["w"]) / len(d["w"])
    kernels[s] = {"launches_sampled": len(d["f"]), "fetch_bytes_raw": f, "fetch_bytes_corrected_x2": 2 * f,
                  "write_bytes": w, "hbm_bytes_per_launch": 2 * f + w}
for spec in a.merge:
    new, parts = spec.split("=")
    parts = [k for k in parts.split(",") if k in kernels]
    n = sum(kernels[k]["launches_sampled"] for k in parts)
    kernels[new] = {"launches_sampled": n, "merged_from": parts}
    for f in ("fetch_bytes_raw", "fetch_bytes_corrected_x2", "write_bytes", "hbm_bytes_per_launch"):
        kernels[new][f] = sum(kernels[k][f] * kernels[k]["launches_sampled"] for k in parts) / n
print(json.dumps({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only); counter unit KB; "
                          "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads; other "
                          "widths uncalibrated); Infinity-Cache hits are included. " + a.note,
                  "dtype": a.dtype, "head": a.head, "csrc_sha16": csrc_sha16(), "kernels": kernels}, indent=1))
