#!/usr/bin/env python3
"""Per-kernel HBM rate from the committed evidence: PMC bytes per launch (profiles/rNN_pmc_traffic.json) divided by the
rocprofv3 average launch duration of the serial trace (profiles/rNN_bench_f16_serial_kernel_stats.csv).
    python tools/kernel_roofline.py [--round r01] [--steps 14]"""
import argparse
import csv
import json
import os
import re

ap = argparse.ArgumentParser()
ap.add_argument("--round", default="r01")
ap.add_argument("--steps", type=int, default=14, help="steps in the serial trace (for ms/step)")
a = ap.parse_args()
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
pmc = json.load(open(os.path.join(root, f"{a.round}_pmc_traffic.json")))["kernels"]
agg = {}
for r in csv.DictReader(open(os.path.join(root, f"{a.round}_bench_f16_serial_kernel_stats.csv"))):
    m = re.search(r"m3::?(\w+?)_kernel", r["Name"]) or re.search(r"N2m3\d+(\w+?)_kernel", r["Name"])
    if not m:
        continue
    d = agg.setdefault(m.group(1), [0, 0.0])
    d[0] += int(r["Calls"]); d[1] += float(r["TotalDurationNs"])
print("| kernel | launches/step | avg µs | ms/step | HBM MB/launch (fetch + write) | TB/s | of 8 TB/s |")
print("|---|---|---|---|---|---|---|")
for k, (calls, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if k not in pmc:
        continue
    f, w = pmc[k]["fetch_bytes_corrected_x2"], pmc[k]["write_bytes"]
    us = ns / calls / 1e3
    tbs = (f + w) / us / 1e6
    print(f"| {k} | {calls / a.steps:.0f} | {us:.1f} | {ns / 1e6 / a.steps:.2f} | {f / 1e6:.1f} + {w / 1e6:.1f} | {tbs:.2f} | {100 * tbs / 8:.0f} % |")
