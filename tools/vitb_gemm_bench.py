#!/usr/bin/env python3
"""Micro-benchmark of m3_gemm_nt at the ViT-Base shapes of BASELINE configs[3] / configs[4] (K = 768 / 3072), 128 x 128
tiles (M3 big = 0) against the 256 x 256-tile kernel (big = 1), with torch.mm (hipBLASLt) on the same operands as the
yardstick for the dense and the per-expert products.  Random operands, HIP events, variants interleaved in one process.
    python tools/vitb_gemm_bench.py [--iters 20] [--only NAME] [--no-mm]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--only", default="")
ap.add_argument("--no-mm", action="store_true")
args = ap.parse_args()
dt = torch.float16
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def rnd(*s, dtype=dt, scale=1.0):
    return (torch.randn(*s, generator=g) * scale).to(dtype).to(dev)


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


def run(name, flops, variants):
    if args.only and args.only not in name:
        return
    best = {k: 1e30 for k in variants}
    for _ in range(args.rounds):
        for k, fn in variants.items():
            best[k] = min(best[k], time_us(fn))
    print(f"{name:44s} " + "  ".join(f"{k} {v:8.1f} us {flops / v / 1e6:7.1f} TF" for k, v in best.items()), flush=True)


def with_big(mode, fn):
    def f():
        ops.gemm_set_big(mode)
        fn()
    return f


def dense_case(name, M, N, K, **kw):
    A, B = rnd(M, K), rnd(N, K, scale=0.05)
    C = torch.empty(M, N, dtype=dt, device=dev)
    extra = {}
    if kw.get("gelu"):
        extra = dict(bias=rnd(N, dtype=torch.float32), act=ops.M3_ACT_GELU, pre_out=torch.empty_like(C))
    v = {"t128": with_big(0, lambda: ops.gemm_nt(A, B, C, **extra)), "t256": with_big(1, lambda: ops.gemm_nt(A, B, C, **extra))}
    if not args.no_mm:
        Bt = B.t()
        v["mm"] = lambda: torch.mm(A, Bt, out=C)
    run(f"dense {name} M={M} N={N} K={K}", 2.0 * M * N * K, v)


def grouped_case(name, T, E, k, D, H):
    R = T * k
    x = rnd(T, D)
    idx = torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(T)]).to(torch.int32).to(dev)
    r = ops.route_build(idx, E)
    W1, b1 = rnd(E, H, D, scale=0.05), rnd(E, H, dtype=torch.float32)
    W2, b2 = rnd(E, D, H, scale=0.05), rnd(E, D, dtype=torch.float32)
    hid, pre = torch.empty(R, H, dtype=dt, device=dev), torch.empty(R, H, dtype=dt, device=dev)
    y = torch.empty(R, D, dtype=dt, device=dev)

    def fc1():
        ops.gemm_nt(x, W1, hid, M=R, bias=b1, act=ops.M3_ACT_GELU, pre_out=pre, a_row_idx=r.row_of_slot, a_row_div=k,
                    group_offsets=r.offsets, tile_starts=r.tile_starts)

    def fc2():
        ops.gemm_nt(hid, W2, y, M=R, bias=b2, c_row_idx=r.row_of_slot, group_offsets=r.offsets, tile_starts=r.tile_starts)

    def dgrad2():
        ops.gemm_nt(y, W1, pre, M=R, gelu_grad_pre=hid, a_row_idx=r.row_of_slot, a_row_div=1, group_offsets=r.offsets,
                    tile_starts=r.tile_starts)
    off = r.offsets.cpu().tolist()
    v1 = {"t128": with_big(0, fc1), "t256": with_big(1, fc1)}
    v2 = {"t128": with_big(0, fc2), "t256": with_big(1, fc2)}
    v3 = {"t128": with_big(0, dgrad2), "t256": with_big(1, dgrad2)}
    if not args.no_mm:
        xs = rnd(R, D)
        W1t, W2t = W1.transpose(1, 2), W2.transpose(1, 2)

        def mm1():
            for e in range(E):
                torch.mm(xs[off[e]:off[e + 1]], W1t[e], out=hid[off[e]:off[e + 1]])

        def mm2():
            for e in range(E):
                torch.mm(hid[off[e]:off[e + 1]], W2t[e], out=y[off[e]:off[e + 1]])
        v1["mm/expert"] = mm1
        v2["mm/expert"] = mm2
    run(f"grouped FC1 gather+gelu+pre {name}", 2.0 * R * D * H, v1)
    run(f"grouped FC2 scatter {name}", 2.0 * R * D * H, v2)
    run(f"grouped FC2 dgrad gather+gelu' {name}", 2.0 * R * D * H, v3)


T4 = 8 * 1201
T3 = 128 * 197
dense_case("cfg4 qkv", T4, 2304, 768)
dense_case("cfg4 proj", T4, 768, 768)
dense_case("cfg4 fc1", T4, 3072, 768, gelu=True)
dense_case("cfg4 fc2", T4, 768, 3072)
dense_case("cfg3 qkv", T3, 2304, 768)
dense_case("cfg3 proj", T3, 768, 768)
dense_case("cfg3 fc1", T3, 3072, 768, gelu=True)
dense_case("cfg3 fc2", T3, 768, 3072)
dense_case("4096^3", 4096, 4096, 4096)
grouped_case("cfg4 E=16 D=768 H=3072", T4, 16, 4, 768, 3072)
grouped_case("cfg3 E=64 D=768 H=768", T3, 64, 4, 768, 768)
ops.gemm_set_big(-1)
