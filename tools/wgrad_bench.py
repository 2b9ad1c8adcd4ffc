#!/usr/bin/env python3
"""Weight-gradient (TN) GEMM per shape of the BASELINE step, operands STREAMED (a ring of (dC, A) pairs larger than the
Infinity Cache) and resident: time of m3_wgrad_tn + its slab reduction, algorithmic bytes (both operands once + the
fp32 slabs written and read once) over that time.
    python tools/wgrad_bench.py [--ring 10] [--only fc1]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ring", type=int, default=6)
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--only", default="")
ap.add_argument("--splits", type=int, default=0, help="override the default row splits")
a = ap.parse_args()
dev = torch.device("cuda:0")
M, E, k = 128 * 197, 16, 4


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


# (name, N, K, grouped)
# (the experts of configs[1] are 384 -> 384 -> 384: moe_mlp_ratio 1)
for name, N, K, grouped in (("qkv", 1152, 384, False), ("proj", 384, 384, False), ("fc1", 1536, 384, False),
                            ("fc2", 384, 1536, False), ("expert fc1", 384, 384, True), ("expert fc2", 384, 385, True)):
    if a.only and name != a.only:
        continue
    if not grouped:
        dCs = [torch.randn(M, N, device=dev).half() for _ in range(a.ring)]
        As = [torch.randn(M, K, device=dev).half() for _ in range(a.ring)]
        dW = torch.zeros(N, K, device=dev)
        db = torch.zeros(N, device=dev)
        splits = a.splits or ops.default_wgrad_splits(M, N, K, 1, torch.float16)
        ws = torch.empty(splits * N * (K + 1), device=dev)
        fn = lambda i: ops.wgrad_tn(dCs[i % a.ring], As[i % a.ring], dW, ws=ws, db=db, splits=splits)  # noqa: E731
        rows = M
        slab = splits * N * K * 4
    else:
        # routed rows: T tokens x k slots, expert-major order; FC1's A operand is gathered from the token rows
        T = M
        R = T * k                           # the step routes each task's tokens: R = T * k rows
        g = torch.Generator().manual_seed(0)
        idx = torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(R // k)]).to(torch.int32).to(dev)
        r = ops.route_build(idx, E)
        rows = R
        gather_c = K == 385                  # "expert fc2": dC rows gathered from the token gradient with the gate score
        K = 384
        if not gather_c:                     # expert FC1: A = token rows gathered through the routing, dC expert-major
            dCs = [torch.randn(R, N, device=dev).half() for _ in range(a.ring)]
            As = [torch.randn(R // k, K, device=dev).half() for _ in range(a.ring)]
            kw = dict(a_row_idx=r.row_of_slot, a_row_div=k)
        else:
            dCs = [torch.randn(R // k, N, device=dev).half() for _ in range(a.ring)]
            As = [torch.randn(R, K, device=dev).half() for _ in range(a.ring)]
            kw = dict(c_row_idx=r.row_of_slot, c_row_div=k, c_row_scale=torch.rand(R // k, k, device=dev))
        dW = torch.zeros(E, N, K, device=dev)
        db = torch.zeros(E, N, device=dev)
        splits = a.splits or ops.default_wgrad_splits(R, N, K, E, torch.float16)
        ws = torch.empty(ops.wgrad_ws_elems(R, N, K, E, grouped=True, dtype=torch.float16), device=dev)
        fn = lambda i: ops.wgrad_tn(dCs[i % a.ring], As[i % a.ring], dW, M=R, ws=ws, db=db, group_offsets=r.offsets, splits=splits, **kw)  # noqa: E731
        slab = ws.numel() * 4
    us_res = timeit(lambda i: fn(0))
    us = timeit(fn)
    byts = rows * (N + K) * 2 + 2 * slab
    print(f"{name:11s} rows {rows:6d} splits {splits:2d}  resident {us_res:6.1f} us | streamed {us:6.1f} us  {2.0 * rows * N * K / us / 1e6:6.0f} TFLOP/s  "
          f"{byts / us / 1e6:5.2f} TB/s algorithmic ({byts / 1e6:.0f} MB, slabs {slab / 1e6:.0f} MB)", flush=True)
    del dCs, As
