#!/usr/bin/env python3
"""NT GEMM with operands STREAMED from HBM: every launch uses a different (A, C) pair out of a ring much larger
than the 256 MB Infinity Cache, as inside a training step.  (tools/gemm_bench.py re-uses one pair: cache-resident.)
    python tools/gemm_stream_bench.py [--ring 12]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ring", type=int, default=12)
ap.add_argument("--iters", type=int, default=48)
ap.add_argument("--only", default="")
a = ap.parse_args()
dev = torch.device("cuda:0")
M = 128 * 197


def timeit(fn):
    for i in range(a.ring):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(a.iters):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / a.iters


for name, N, K, two_out in (("qkv", 1152, 384, False), ("proj", 384, 384, False), ("fc1+gelu+pre", 1536, 384, True),
                            ("dpre (gelu')", 1536, 384, "gpre"), ("fc2", 384, 1536, False)):
    if a.only and name != a.only:
        continue
    As = [torch.randn(M, K, device=dev).half() for _ in range(a.ring)]
    B = (torch.randn(N, K, device=dev) * 0.05).half()
    Cs = [torch.empty(M, N, dtype=torch.float16, device=dev) for _ in range(a.ring)]
    Ps = [torch.randn(M, N, device=dev).half() for _ in range(a.ring)] if two_out else None
    bias = torch.zeros(N, device=dev)
    if two_out == "gpre":           # d pre = (d y W2) * GELU'(pre): reads a second [M, N] fp16 operand instead of writing one
        fn = lambda i: ops.gemm_nt(As[i % a.ring], B, Cs[i % a.ring], gelu_grad_pre=Ps[i % a.ring])  # noqa: E731
    elif two_out:
        fn = lambda i: ops.gemm_nt(As[i % a.ring], B, Cs[i % a.ring], bias=bias, act=ops.M3_ACT_GELU, pre_out=Ps[i % a.ring])  # noqa: E731
    else:
        fn = lambda i: ops.gemm_nt(As[i % a.ring], B, Cs[i % a.ring])  # noqa: E731
    us_res = timeit(lambda i: fn(0))
    us = timeit(fn)
    byts = M * K * 2 + M * N * 2 * (2 if two_out else 1)
    print(f"{name:14s} resident {us_res:6.1f} us | streamed {us:6.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TFLOP/s  "
          f"{byts / us / 1e6:5.2f} TB/s algorithmic", flush=True)
    del As, Cs, Ps
