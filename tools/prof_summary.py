#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 rocpd database (run_results.db): launches and time per bench step.
    python tools/prof_summary.py gpurun_out/prof_b/run_results.db --steps 11 [--csv out.csv]
"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--steps", type=float, default=1.0, help="bench steps contained in the trace")
ap.add_argument("--top", type=int, default=40)
ap.add_argument("--csv", default="")
a = ap.parse_args()
c = sqlite3.connect(a.db).cursor()
rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot / a.steps / 1e6:.3f} ms/step, {sum(r[1] for r in rows) / a.steps:.0f} launches/step")
for r in rows[: a.top]:
    print(f"{r[0][:84]:84s} n/step={r[1] / a.steps:7.1f} ms/step={r[2] / a.steps / 1e6:7.3f} avg_us={r[3] / 1e3:8.1f}")
if a.csv:
    with open(a.csv, "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            f.write(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.1f},{100.0 * r[2] / tot:.2f},{r[4]},{r[5]}\n")
