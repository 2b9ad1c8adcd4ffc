#!/usr/bin/env python3
"""Micro-benchmark of the NT / TN GEMM kernels on the BASELINE config-2 shapes (HIP events).
    python tools/gemm_bench.py [--dtype f16|f32] [--iters 20] [--only NAME]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f16")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="")
args = ap.parse_args()
dt = torch.float16 if args.dtype == "f16" else torch.float32
dev = torch.device("cuda:0")
T, D, E, k = 128 * 197, 384, 16, 4
R = T * k
g = torch.Generator().manual_seed(0)


def rnd(*s, dtype=dt, scale=1.0):
    return (torch.randn(*s, generator=g) * scale).to(dtype).to(dev)


def timeit(fn, flops, name):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / args.iters
    print(f"{name:34s} {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s", flush=True)


x = rnd(T, D)
idx = torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(T)]).to(torch.int32).to(dev)
route = ops.route_build(idx, E)
cases = []
for name, (M, N, K) in {"qkv": (T, 1152, 384), "proj": (T, 384, 384), "fc1": (T, 1536, 384), "fc2": (T, 384, 1536)}.items():
    A, B = rnd(M, K), rnd(N, K, scale=0.05)
    C = torch.empty(M, N, dtype=dt, device=dev)
    bias = rnd(N, dtype=torch.float32)
    cases.append((f"dense {name} plain", lambda A=A, B=B, C=C: ops.gemm_nt(A, B, C), 2.0 * M * N * K))
    if name == "fc1":
        pre = torch.empty_like(C)
        cases.append((f"dense {name} bias+gelu+pre", lambda A=A, B=B, C=C, bias=bias, pre=pre: ops.gemm_nt(A, B, C, bias=bias, act=ops.M3_ACT_GELU, pre_out=pre), 2.0 * M * N * K))
W1, b1 = rnd(E, D, D, scale=0.05), rnd(E, D, dtype=torch.float32)
hid, pre = torch.empty(R, D, dtype=dt, device=dev), torch.empty(R, D, dtype=dt, device=dev)
y = torch.empty(R, D, dtype=dt, device=dev)
cases.append(("grouped FC1 gather+gelu+pre", lambda: ops.gemm_nt(x, W1, hid, M=R, bias=b1, act=ops.M3_ACT_GELU, pre_out=pre, a_row_idx=route.row_of_slot, a_row_div=k, group_offsets=route.offsets, tile_starts=route.tile_starts), 2.0 * R * D * D))
cases.append(("grouped FC1 gather plain", lambda: ops.gemm_nt(x, W1, hid, M=R, a_row_idx=route.row_of_slot, a_row_div=k, group_offsets=route.offsets, tile_starts=route.tile_starts), 2.0 * R * D * D))
cases.append(("grouped FC2 scatter", lambda: ops.gemm_nt(hid, W1, y, M=R, bias=b1, c_row_idx=route.row_of_slot, group_offsets=route.offsets, tile_starts=route.tile_starts), 2.0 * R * D * D))
cases.append(("grouped plain (no gather)", lambda: ops.gemm_nt(hid, W1, y, M=R, group_offsets=route.offsets, tile_starts=route.tile_starts), 2.0 * R * D * D))
dW = torch.empty(1536, D, device=dev)
dC = rnd(T, 1536)
cases.append(("wgrad dense fc1 (N=1536,K=384)", lambda: ops.wgrad_tn(dC, x, dW), 2.0 * T * 1536 * D))
dWe = torch.empty(E, D, D, device=dev)
cases.append(("wgrad grouped (E=16,384x384)", lambda: ops.wgrad_tn(hid, pre, dWe, M=R, group_offsets=route.offsets), 2.0 * R * D * D))
for name, fn, fl in cases:
    if args.only and args.only not in name:
        continue
    timeit(fn, fl, name)
