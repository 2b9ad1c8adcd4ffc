#!/usr/bin/env python3
"""Sweep K (and N) of the plain dense NT GEMM to separate per-tile fixed cost from per-K-step cost."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import ops
dev = torch.device("cuda:0")
M = 128 * 197
for dt in (torch.float16,):
    for N in (384, 1152):
        for K in (64, 128, 384, 768, 1536, 3072):
            A = torch.randn(M, K, device=dev).to(dt); B = (torch.randn(N, K, device=dev) * 0.05).to(dt)
            C = torch.empty(M, N, dtype=dt, device=dev)
            for _ in range(3): ops.gemm_nt(A, B, C)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): ops.gemm_nt(A, B, C)
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) * 1e3 / 20
            tiles = ((M + 127) // 128) * ((N + 127) // 128)
            print(f"N={N:5d} K={K:5d} nk={K*2//128:3d} tiles={tiles:5d} {us:8.1f} us {2.0*M*N*K/us/1e6:8.1f} TF  per-tile-slot {us*512/tiles:6.2f} us", flush=True)
