#!/usr/bin/env python3
"""A/B of the weight-gradient (TN) kernels on every weight-gradient shape of BASELINE configs[1] / [3] / [4] (one task pass,
single-GPU forms): round 4's path (register-staged kernel, row splits for 512 slots, slabs + reduction always) against the
LDS-DMA kernel with the splits sized for 512 / 1024 slots and direct accumulation where a launch has one part per group,
and the library's default (which adds the 256 x 256 tiles for the 16-bit ViT-Base shapes).
Variants interleaved in ONE process, random operands STREAMED - a ring of (dC, A) sets larger than the 256 MiB Infinity
Cache, as inside the training step, where a weight gradient's operands were written many launches earlier (the first
version of this tool re-read one resident set and over-stated every variant, the DMA kernel most) - HIP events around
m3_wgrad_tn with the slab reduction riding on the next launch, as the engine's queue issues them.
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
    # (mode of m3_wgrad_set_dma, direct accumulation, slots the splits are sized for, 256 x 256 tiles)
    fns = {"r4": (0, False, 512, 0, mk()), "dma/512": (2, True, 512, 0, mk()), "dma/1024": (2, True, 1024, 0, mk()),
           "default": (1, True, 0, 1, mk())}
    for _ in range(args.rounds):
        for k_, (mode, direct, slots, big, fn) in fns.items():
            ops.wgrad_set_dma(mode)
            ops.wgrad_set_big(big)
            O._WGRAD_DIRECT, O._WGRAD_SLOTS = direct, slots
            best[k_] = min(best.get(k_, 1e30), time_us(fn))
    O._WGRAD_DIRECT, O._WGRAD_SLOTS = True, 0
    ops.wgrad_set_big(-1)
    print(f"{name:50s} " + "  ".join(f"{k_} {v:7.1f} us {flops / v / 1e6:6.1f} TF" for k_, v in best.items()), flush=True)


RING_BYTES = 320e6


def ring_of(nbytes):
    return int(max(2, min(8, -(-RING_BYTES // nbytes))))


def dense(name, M, N, K, dtype=dt):
    es = 2 if dtype != torch.float32 else 4
    n = ring_of(M * (N + K) * es)
    sets = [(rnd(M, N, dtype=dtype), rnd(M, K, dtype=dtype)) for _ in range(n)]
    dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    q = ops.WgradQueue(3 * ops.wgrad_ws_elems(M, N, K, 1, grouped=False, dtype=dtype), dev)

    def mk():
        st = [0]

        def fn():
            dC, A = sets[st[0] % n]
            st[0] += 1
            ops.wgrad_tn(dC, A, dW, beta=1, db=db, queue=q)
        return fn
    run(f"dense {name} M={M} N={N} K={K}{' f32' if es == 4 else ''} (ring {n})", 2.0 * M * N * K, mk)


def experts(name, T, E, k, D, H, dtype=dt):
    R = T * k
    es = 2 if dtype != torch.float32 else 4
    idx = torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(T)]).to(torch.int32).to(dev)
    r = ops.route_build(idx, E)
    n = ring_of(R * (H + D) * es)
    sets = [(rnd(T, D, dtype=dtype), rnd(R, H, dtype=dtype), rnd(R, H, dtype=dtype), rnd(T, D, dtype=dtype)) for _ in range(n)]
    score = torch.rand(R, generator=g).to(dev)
    dW1, db1 = torch.zeros(E, H, D, device=dev), torch.zeros(E, H, device=dev)
    dW2, db2 = torch.zeros(E, D, H, device=dev), torch.zeros(E, D, device=dev)
    q = ops.WgradQueue(3 * max(ops.wgrad_ws_elems(R, H, D, E, grouped=True, dtype=dtype), ops.wgrad_ws_elems(R, D, H, E, grouped=True, dtype=dtype)), dev)

    def mk1():
        st = [0]

        def fn():
            x, dhp, _, _ = sets[st[0] % n]
            st[0] += 1
            ops.wgrad_tn(dhp, x, dW1, M=R, beta=1, a_row_idx=r.row_of_slot, a_row_div=k, group_offsets=r.offsets, db=db1, queue=q)
        return fn

    def mk2():
        st = [0]

        def fn():
            _, _, hid, dout = sets[st[0] % n]
            st[0] += 1
            ops.wgrad_tn(dout, hid, dW2, M=R, beta=1, c_row_idx=r.row_of_slot, c_row_div=k, c_row_scale=score,
                         group_offsets=r.offsets, db=db2, queue=q)
        return fn
    tag = f"{name}{' f32' if es == 4 else ''} (ring {n})"
    run(f"expert FC1 (gathered A) {tag}", 2.0 * R * D * H, mk1)
    run(f"expert FC2 (d out through the gate score) {tag}", 2.0 * R * D * H, mk2)


if args.config in (0, 1):
    T = 128 * 197
    for n, N, K in (("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536)):
        dense("cfg1 " + n, T, N, K)
    experts("cfg1 E=16 D=H=384", T, 16, 4, 384, 384)
    for n, N, K in (("qkv", 1152, 384), ("fc1", 1536, 384), ("fc2", 384, 1536)):
        dense("cfg1 " + n, T, N, K, dtype=torch.float32)
    experts("cfg1 E=16 D=H=384", T, 16, 4, 384, 384, dtype=torch.float32)
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
