#!/usr/bin/env python3
"""Condensed view of a bench.py JSON line: python tools/show_bench.py <file>"""
import json
import sys

d = json.load(open(sys.argv[1]))
keep = ("value", "ms_per_step", "model_tflops", "ratio_to_value")
print({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk in keep}) for k, v in d.items()
       if k not in ("config", "roofline")})
if "roofline" in d:
    print("roofline", d["roofline"])
for k in ("f32", "configs3", "configs4"):
    if k in d and "roofline" in d[k]:
        r = d[k]["roofline"]
        print(k, {kk: r.get(kk) for kk in ("bound", "frac", "achieved", "mfma_tflops", "expert_grouped_gemm_tflops", "avg_launch_us")})
