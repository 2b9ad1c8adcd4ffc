#!/usr/bin/env python3
"""A/B of the weight-gradient (TN) kernels on every weight-gradient shape of BASELINE configs[1] / [3] / [4] (one task pass,
single-GPU forms): register-staged (M3 dma = 0) against the LDS-DMA kernel (dma = 1), variants interleaved in ONE process,
random operands, HIP events around m3_wgrad_tn + its slab reduction (as the engine's queue issues them: the reduce rides on
the next launch, so a pair of calls is timed and halved).
    python tools/wgrad_ab_bench.py [--config 1|3|4] [--iters 20] [--only NAME]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=0, help="0 = all three")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--only", default="")
args = ap.parse_args()
dt = torch.float16
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def rnd(*s, dtype=dt):
    return torch.randn(*s, generator=g).to(dtype).to(dev)


def time_us(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / args.iters


def run(name, flops, mk):
    if args.only and args.only not in name:
        return
    best = {}
    import m3vit_amd.ops as O
    fns = {"staged": (0, False, 512, mk()), "dma": (1, False, 512, mk()), "dma/1024": (1, False, 1024, mk()),
           "dma+direct": (1, True, 512, mk())}
    for _ in range(args.rounds):
        for k_, (mode, direct, slots, fn) in fns.items():
            ops.wgrad_set_dma(mode)
            O._WGRAD_DIRECT, O._WGRAD_SLOTS = direct, slots
            best[k_] = min(best.get(k_, 1e30), time_us(fn))
    O._WGRAD_DIRECT, O._WGRAD_SLOTS = True, 512
    print(f"{name:50s} " + "  ".join(f"{k_} {v:7.1f} us {flops / v / 1e6:6.1f} TF" for k_, v in best.items()), flush=True)


def dense(name, M, N, K):
    dC, A = rnd(M, N), rnd(M, K)
    dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    q = ops.WgradQueue(3 * ops.wgrad_ws_elems(M, N, K, 1, grouped=False, dtype=dt), dev)

    def mk():
        return lambda: ops.wgrad_tn(dC, A, dW, beta=1, db=db, queue=q)
    run(f"dense {name} M={M} N={N} K={K}", 2.0 * M * N * K, mk)


def experts(name, T, E, k, D, H):
    R = T * k
    idx = torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(T)]).to(torch.int32).to(dev)
    r = ops.route_build(idx, E)
    x, dhp, hid, dy = rnd(T, D), rnd(R, H), rnd(R, H), rnd(R, D)
    dW1, db1 = torch.zeros(E, H, D, device=dev), torch.zeros(E, H, device=dev)
    dW2, db2 = torch.zeros(E, D, H, device=dev), torch.zeros(E, D, device=dev)
    q = ops.WgradQueue(3 * max(ops.wgrad_ws_elems(R, H, D, E, grouped=True, dtype=dt), ops.wgrad_ws_elems(R, D, H, E, grouped=True, dtype=dt)), dev)

    def mk1():
        return lambda: ops.wgrad_tn(dhp, x, dW1, M=R, beta=1, a_row_idx=r.row_of_slot, a_row_div=k, group_offsets=r.offsets, db=db1, queue=q)

    def mk2():
        return lambda: ops.wgrad_tn(dy, hid, dW2, M=R, beta=1, c_row_idx=r.row_of_slot, group_offsets=r.offsets, db=db2, queue=q)
    run(f"expert FC1 (gathered A) {name}", 2.0 * R * D * H, mk1)
    run(f"expert FC2 (gathered dC, no score) {name}", 2.0 * R * D * H, mk2)


if args.config in (0, 1):
    T = 128 * 197
    for n, N, K in (("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536)):
        dense("cfg1 " + n, T, N, K)
    experts("cfg1 E=16 D=H=384", T, 16, 4, 384, 384)
if args.config in (0, 3):
    T = 128 * 197
    for n, N, K in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
        dense("cfg3 " + n, T, N, K)
    experts("cfg3 E=64 D=H=768", T, 64, 4, 768, 768)
if args.config in (0, 4):
    T = 8 * 1201
    for n, N, K in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
        dense("cfg4 " + n, T, N, K)
    experts("cfg4 E=16 D=768 H=3072", T, 16, 4, 768, 3072)
ops.wgrad_set_dma(-1)
