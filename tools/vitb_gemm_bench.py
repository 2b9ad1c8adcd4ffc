#!/usr/bin/env python3
"""Micro-benchmark of m3_gemm_nt at the ViT-Base shapes of BASELINE configs[3] / configs[4] (K = 768 / 3072), 128 x 128
tiles (M3 big = 0) against the 256 x 256-tile kernel (big = 1), with torch.mm (hipBLASLt) on the same operands as the
yardstick for the dense and the per-expert products.  Random operands STREAMED (a ring of activation sets larger than the
256 MiB Infinity Cache: inside the training step a GEMM's input was written launches earlier; --resident re-reads one set),
HIP events, variants interleaved in one process.
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
ap.add_argument("--resident", action="store_true")
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


def ring_of(nbytes):
    return 1 if args.resident else int(max(2, min(8, -(-320e6 // nbytes))))


def dense_case(name, M, N, K, **kw):
    n = ring_of(M * (K + N) * 2)
    As = [rnd(M, K) for _ in range(n)]
    B = rnd(N, K, scale=0.05)
    Cs = [torch.empty(M, N, dtype=dt, device=dev) for _ in range(n)]
    extra = {}
    if kw.get("gelu"):
        extra = dict(bias=rnd(N, dtype=torch.float32), act=ops.M3_ACT_GELU)
        pres = [torch.empty(M, N, dtype=dt, device=dev) for _ in range(n)]
    st = [0]

    def go(mode):
        def f():
            i = st[0] % n
            st[0] += 1
            ops.gemm_set_big(mode)
            ops.gemm_nt(As[i], B, Cs[i], **(dict(extra, pre_out=pres[i]) if extra else {}))
        return f
    v = {"t128": go(0), "t256": go(1)}
    if not args.no_mm:
        Bt = B.t()

        def mm():
            i = st[0] % n
            st[0] += 1
            torch.mm(As[i], Bt, out=Cs[i])
        v["mm"] = mm
    run(f"dense {name} M={M} N={N} K={K} (ring {n})", 2.0 * M * N * K, v)


def grouped_case(name, T, E, k, D, H):
    R = T * k
    idx = torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(T)]).to(torch.int32).to(dev)
    r = ops.route_build(idx, E)
    W1, b1 = rnd(E, H, D, scale=0.05), rnd(E, H, dtype=torch.float32)
    W2, b2 = rnd(E, D, H, scale=0.05), rnd(E, D, dtype=torch.float32)
    n = ring_of(R * (H + D) * 2)
    xs = [rnd(T, D) for _ in range(n)]
    hids = [rnd(R, H) for _ in range(n)]
    pres = [torch.empty(R, H, dtype=dt, device=dev) for _ in range(n)]
    ys = [rnd(R, D) for _ in range(n)]
    st = [0]

    def nxt():
        i = st[0] % n
        st[0] += 1
        return i

    def fc1(mode):
        def f():
            i = nxt()
            ops.gemm_set_big(mode)
            ops.gemm_nt(xs[i], W1, hids[i], M=R, bias=b1, act=ops.M3_ACT_GELU, pre_out=pres[i], a_row_idx=r.row_of_slot, a_row_div=k,
                        group_offsets=r.offsets, tile_starts=r.tile_starts)
        return f

    def fc2(mode):
        def f():
            i = nxt()
            ops.gemm_set_big(mode)
            ops.gemm_nt(hids[i], W2, ys[i], M=R, bias=b2, c_row_idx=r.row_of_slot, group_offsets=r.offsets, tile_starts=r.tile_starts)
        return f

    def dgrad2(mode):
        def f():
            i = nxt()
            ops.gemm_set_big(mode)
            ops.gemm_nt(ys[i], W1, pres[i], M=R, gelu_grad_pre=hids[i], a_row_idx=r.row_of_slot, a_row_div=1, group_offsets=r.offsets,
                        tile_starts=r.tile_starts)
        return f
    run(f"grouped FC1 gather+gelu+pre {name} (ring {n})", 2.0 * R * D * H, {"t128": fc1(0), "t256": fc1(1)})
    run(f"grouped FC2 scatter {name} (ring {n})", 2.0 * R * D * H, {"t128": fc2(0), "t256": fc2(1)})
    run(f"grouped FC2 dgrad gather+gelu' {name} (ring {n})", 2.0 * R * D * H, {"t128": dgrad2(0), "t256": dgrad2(1)})


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
