"""The weight-gradient calls of ONE dense block's backward in the order the engine issues them (fc2, fc1, proj, qkv; one
shared queue: every launch carries the previous call's slab reduction), each call timed by its own HIP-event pair, operands
streamed through a ring.  Looks for what a per-shape A/B (tools/wgrad_ab_bench.py: every shape rides its OWN reduction) cannot
see.   python tools/wgrad_seq_probe.py [--M 9608] [--D 768] [--H 3072] [--dtype f16] [--no-bias] [--no-queue]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=9608)
ap.add_argument("--D", type=int, default=768)
ap.add_argument("--H", type=int, default=3072)
ap.add_argument("--dtype", default="f16")
ap.add_argument("--no-bias", action="store_true")
ap.add_argument("--no-queue", action="store_true")
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[args.dtype]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, D, H = args.M, args.D, args.H
shapes = [("fc2", D, H), ("fc1", H, D), ("proj", D, D), ("qkv", 3 * D, D)]
es = 4 if dt == torch.float32 else 2
ring = int(max(2, min(8, -(-320e6 // (M * (H + D) * es)))))
ops_ = {}
for name, N, K in shapes:
    ops_[name] = ([(torch.randn(M, N, generator=g).to(dt).to(dev), torch.randn(M, K, generator=g).to(dt).to(dev)) for _ in range(ring)],
                  torch.zeros(N, K, device=dev), None if args.no_bias else torch.zeros(N, device=dev))
need = max(ops.wgrad_ws_elems(M, N, K, 1, grouped=False, dtype=dt) for _, N, K in shapes)
q = None if args.no_queue else ops.WgradQueue(need, dev)
ws = torch.empty(need, dtype=torch.float32, device=dev)
times = {n: [] for n, _, _ in shapes}
for it in range(args.iters + 2):
    evs = []
    for name, N, K in shapes:
        sets, dW, db = ops_[name]
        dC, A = sets[it % ring]
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.wgrad_tn(dC, A, dW, beta=1, db=db, ws=ws, queue=q)
        e.record()
        evs.append((name, s, e))
    torch.cuda.synchronize()
    if it >= 2:
        for name, s, e in evs:
            times[name].append(s.elapsed_time(e) * 1e3)
print(f"# M={M} D={D} H={H} {args.dtype} bias={not args.no_bias} queue={not args.no_queue} ring={ring}")
for name, N, K in shapes:
    t = sorted(times[name])[len(times[name]) // 2]
    sp = ops.default_wgrad_splits(M, N, K, 1, dt)
    print(f"{name:5s} N={N:5d} K={K:5d} splits={sp:2d} median {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TFLOP/s")
