#!/usr/bin/env python3
"""Per launch SHAPE comparison of two rocprofv3 `--kernel-trace` kernel_trace CSVs of the same bench command: dispatches
are grouped by (kernel, grid size, workgroup size) - the expert grouped GEMMs, the qkv / proj / fc1 / fc2 GEMMs, the
weight gradients of each layer type ... show up as separate rows.
    python tools/prof_by_launch.py a_kernel_trace.csv [b_kernel_trace.csv] --steps 14 [--match gemm]"""
import argparse
import csv
import re

ap = argparse.ArgumentParser()
ap.add_argument("a")
ap.add_argument("b", nargs="?")
ap.add_argument("--steps", type=float, default=14.0)
ap.add_argument("--match", default="")
ap.add_argument("--min-ms", type=float, default=0.02)
a = ap.parse_args()


def load(path):
    agg = {}
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name") or r.get("Name")
        m = re.search(r"m3::?(\w+?)_kernel", name) or re.search(r"N2m3\d+(\w+?)_kernel", name)
        k = m.group(1) if m else name[:32]
        tmpl = re.search(r"kernelI(\w+?)E[Ev]", name)
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)))
        key = (k, (tmpl.group(1) if tmpl else "")[:18], grid // max(wg, 1))
        d = agg.setdefault(key, [0, 0.0])
        d[0] += 1; d[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return agg


A = load(a.a)
B = load(a.b) if a.b else {}
keys = sorted(set(A) | set(B), key=lambda k: -max(A.get(k, [0, 0])[1], B.get(k, [0, 0])[1]))
print(f"{'kernel':22s} {'template':18s} {'WGs':>7s} | {'n/step':>6s} {'avg us A':>9s} {'ms A':>7s} | {'n/step':>6s} {'avg us B':>9s} {'ms B':>7s} | {'B - A':>7s}")
for k in keys:
    if a.match and a.match not in k[0]:
        continue
    ca, na = A.get(k, [0, 0.0]); cb, nb = B.get(k, [0, 0.0])
    ma, mb = na / 1e6 / a.steps, nb / 1e6 / a.steps
    if max(ma, mb) < a.min_ms:
        continue
    print(f"{k[0]:22s} {k[1]:18s} {k[2]:7d} | {ca / a.steps:6.1f} {na / max(ca, 1) / 1e3:9.1f} {ma:7.3f} | {cb / a.steps:6.1f} {nb / max(cb, 1) / 1e3:9.1f} {mb:7.3f} | {mb - ma:+7.3f}")
