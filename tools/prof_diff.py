#!/usr/bin/env python3
"""Side-by-side per-kernel comparison of two rocprofv3 `--kernel-trace --stats` kernel_stats CSVs of the same bench
command (same number of steps): calls per step, average launch, ms per step, difference.
    python tools/prof_diff.py a_kernel_stats.csv b_kernel_stats.csv --steps 14"""
import argparse
import csv
import re

ap = argparse.ArgumentParser()
ap.add_argument("a")
ap.add_argument("b")
ap.add_argument("--steps", type=float, default=14.0, help="steps in each trace (warm-up + timed + the instrumented extra ones)")
a = ap.parse_args()


def load(path):
    agg = {}
    for r in csv.DictReader(open(path)):
        m = re.search(r"m3::?(\w+?)_kernel", r["Name"]) or re.search(r"N2m3\d+(\w+?)_kernel", r["Name"])
        k = m.group(1) if m else r["Name"][:40]
        d = agg.setdefault(k, [0, 0.0])
        d[0] += int(r["Calls"]); d[1] += float(r["TotalDurationNs"])
    return agg


A, B = load(a.a), load(a.b)
keys = sorted(set(A) | set(B), key=lambda k: -max(A.get(k, [0, 0])[1], B.get(k, [0, 0])[1]))
ta = tb = 0.0
print(f"{'kernel':28s} {'n/step A':>9s} {'avg us A':>9s} {'ms A':>7s} | {'n/step B':>9s} {'avg us B':>9s} {'ms B':>7s} | {'B - A ms':>8s}")
for k in keys:
    ca, na = A.get(k, [0, 0.0]); cb, nb = B.get(k, [0, 0.0])
    ma, mb = na / 1e6 / a.steps, nb / 1e6 / a.steps
    ta += ma; tb += mb
    print(f"{k:28s} {ca / a.steps:9.1f} {na / max(ca, 1) / 1e3:9.1f} {ma:7.3f} | {cb / a.steps:9.1f} {nb / max(cb, 1) / 1e3:9.1f} {mb:7.3f} | {mb - ma:+8.3f}")
print(f"{'sum':28s} {'':9s} {'':9s} {ta:7.3f} | {'':9s} {'':9s} {tb:7.3f} | {tb - ta:+8.3f}")
