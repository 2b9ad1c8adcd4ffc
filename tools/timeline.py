#!/usr/bin/env python3
"""Concurrency of the two-stream step from a rocprofv3 kernel trace (rocpd database): per step-sized window, how long
0 / 1 / >= 2 kernels were in flight, and which kernels run while nothing else does.
    rocprofv3 --kernel-trace -d out -o run -- python bench.py --steps 8 --warmup 2 --no-f32 --no-cpu-baseline
    python tools/timeline.py out/.../run_results.db"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1]).cursor()
rows = list(c.execute("select name, start, end from kernels order by start"))
if not rows:
    sys.exit("no kernels")
# the timed region: the last 60 % of the trace (skips engine set-up, capture and warm-up)
t_lo = rows[0][1] + 0.4 * (rows[-1][2] - rows[0][1])
rows = [r for r in rows if r[1] >= t_lo]
ev = []
for i, (n, s, e) in enumerate(rows):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
depth, last = 0, ev[0][0]
hist = defaultdict(int)
alone = defaultdict(int)
active = set()
for t, d, i in ev:
    dt = t - last
    hist[min(depth, 3)] += dt
    if depth == 1:
        alone[rows[next(iter(active))][0]] += dt
    last = t
    depth += d
    (active.add if d > 0 else active.discard)(i)
span = ev[-1][0] - ev[0][0]
busy = sum(e - s for _, s, e in rows)
print(f"window {span / 1e6:.2f} ms, {len(rows)} launches, sum of kernel durations {busy / 1e6:.2f} ms ({busy / span:.2f} x the window)")
for k in sorted(hist):
    print(f"  {'>= 3' if k == 3 else k} kernel(s) in flight: {hist[k] / 1e6:8.2f} ms  {100 * hist[k] / span:5.1f} %")
print("time with exactly one kernel in flight, by kernel:")
for n, v in sorted(alone.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {v / 1e6:7.2f} ms  {100 * v / span:5.1f} %  {n[:90]}")
