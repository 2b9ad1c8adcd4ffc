#!/usr/bin/env python3
"""Per-kernel averages of whatever counters a rocprofv3 --pmc pass collected:  python tools/pmc_dump.py run_results.db [...]"""
import re
import sqlite3
import sys

for db in sys.argv[1:]:
    c = sqlite3.connect(db).cursor()
    cols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    acc = {}
    for name, cname, val in c.execute(f"select {name_col}, counter_name, value from counters_collection"):
        m = re.search(r"m3::?(\w+?)_kernel", name) or re.search(r"N2m3\d+(\w+?)_kernel", name)
        if not m:
            continue
        d = acc.setdefault((m.group(1), cname), [0.0, 0])
        d[0] += float(val)
        d[1] += 1
    for (k, cn), (s, n) in sorted(acc.items()):
        print(f"{k:24s} {cn:34s} avg {s / n:16.1f}  (n={n})")
