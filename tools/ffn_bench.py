#!/usr/bin/env python3
"""Fused FFN forward (m3_ffn_fwd) against the unfused m3_gemm_nt pair at the configs[1] shapes, operands streamed
from HBM (a ring of (X, Y) pairs larger than the Infinity Cache), interleaved rounds in ONE process.
    python tools/ffn_bench.py [--ring 8] [--iters 24] [--rounds 3]
expert: R = 100864 routed rows, E = 16, D = H = 384 (gather by row_of_slot / k, token-major scatter)
dense : T = 25216 rows, D = 384, H = 1536, fp32 output + fp32 residual
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ring", type=int, default=8)
ap.add_argument("--iters", type=int, default=24)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--only", default="")
a = ap.parse_args()
dev = torch.device("cuda:0")
T, D, E, k = 128 * 197, 384, 16, 4


def perm32(n):
    p = torch.arange(n)
    w = p % 32
    return (p - w) + 16 * ((w & 7) >> 2) + 4 * (w >> 3) + (w & 3)


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


def report(name, flops, variants):
    best = {n: 1e30 for n in variants}
    for _ in range(a.rounds):
        for n, fn in variants.items():
            best[n] = min(best[n], timeit(fn))
    for n, us in best.items():
        print(f"{name:8s} {n:10s} {us:7.1f} us  {flops / us / 1e6:6.0f} TFLOP/s  ({flops / us / 1e6 / 2500:.3f} of 2.5 PF)", flush=True)


if a.only in ("", "expert"):
    H = 384
    R = T * k
    Xs = [torch.randn(T, D, device=dev).half() for _ in range(a.ring)]
    Ys = [torch.empty(R, D, dtype=torch.float16, device=dev) for _ in range(a.ring)]
    hid = [torch.empty(R, H, dtype=torch.float16, device=dev) for _ in range(2)]
    pre = [torch.empty(R, H, dtype=torch.float16, device=dev) for _ in range(2)]
    w1 = (torch.randn(E, H, D, device=dev) * 0.05).half()
    w2 = (torch.randn(E, D, H, device=dev) * 0.05).half()
    w2p = w2[..., perm32(H).to(dev)].contiguous()
    b1, b2 = torch.zeros(E, H, device=dev), torch.zeros(E, D, device=dev)
    idx = torch.stack([torch.randperm(E)[:k] for _ in range(T)]).to(torch.int32).to(dev)
    r = ops.route_build(idx, E)

    def fused(i):
        ops.ffn_fwd(Xs[i % a.ring], w1, w2p, Ys[i % a.ring], b1=b1, b2=b2, M=R, x_row_idx=r.row_of_slot, x_row_div=k,
                    y_row_idx=r.row_of_slot, group_offsets=r.offsets)

    def unfused(i):
        ops.gemm_nt(Xs[i % a.ring], w1, hid[i % 2], M=R, bias=b1, act=ops.M3_ACT_GELU, pre_out=pre[i % 2],
                    a_row_idx=r.row_of_slot, a_row_div=k, group_offsets=r.offsets, tile_starts=r.tile_starts)
        ops.gemm_nt(hid[i % 2], w2, Ys[i % a.ring], M=R, bias=b2, c_row_idx=r.row_of_slot, group_offsets=r.offsets,
                    tile_starts=r.tile_starts)

    fused(0); unfused(1)
    torch.cuda.synchronize()
    y_f = Ys[0].float().clone()
    ops.gemm_nt(Xs[0], w1, hid[0], M=R, bias=b1, act=ops.M3_ACT_GELU, a_row_idx=r.row_of_slot, a_row_div=k,
                group_offsets=r.offsets, tile_starts=r.tile_starts)
    ops.gemm_nt(hid[0], w2, Ys[1], M=R, bias=b2, c_row_idx=r.row_of_slot, group_offsets=r.offsets, tile_starts=r.tile_starts)
    torch.cuda.synchronize()
    print(f"expert fused vs unfused rel diff {float((y_f - Ys[1].float()).norm() / Ys[1].float().norm()):.2e}", flush=True)
    report("expert", 4.0 * R * D * H, {"fused": fused, "unfused": unfused})
    del Xs, Ys, hid, pre

if a.only in ("", "dense"):
    H = 1536
    Xs = [torch.randn(T, D, device=dev).half() for _ in range(a.ring)]
    Rs = [torch.randn(T, D, device=dev) for _ in range(a.ring)]
    Ys = [torch.empty(T, D, device=dev) for _ in range(a.ring)]
    hid = [torch.empty(T, H, dtype=torch.float16, device=dev) for _ in range(2)]
    pre = [torch.empty(T, H, dtype=torch.float16, device=dev) for _ in range(2)]
    w1 = (torch.randn(H, D, device=dev) * 0.05).half()
    w2 = (torch.randn(D, H, device=dev) * 0.05).half()
    w2p = w2[..., perm32(H).to(dev)].contiguous()
    b1, b2 = torch.zeros(H, device=dev), torch.zeros(D, device=dev)

    def fused_d(i):
        ops.ffn_fwd(Xs[i % a.ring], w1, w2p, Ys[i % a.ring], b1=b1, b2=b2, residual=Rs[i % a.ring])

    def unfused_d(i):
        ops.gemm_nt(Xs[i % a.ring], w1, hid[i % 2], bias=b1, act=ops.M3_ACT_GELU, pre_out=pre[i % 2])
        ops.gemm_nt(hid[i % 2], w2, Ys[i % a.ring], bias=b2, residual=Rs[i % a.ring])

    fused_d(0)
    torch.cuda.synchronize()
    y_f = Ys[0].clone()
    unfused_d(0)
    torch.cuda.synchronize()
    print(f"dense fused vs unfused rel diff {float((y_f - Ys[0]).norm() / Ys[0].norm()):.2e}", flush=True)
    report("dense", 4.0 * T * D * H, {"fused": fused_d, "unfused": unfused_d})
