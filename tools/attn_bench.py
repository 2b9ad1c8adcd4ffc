#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels at the BASELINE config-2 shape (B=128, N=197, 12 heads, dh=32)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import ops
dev = torch.device("cuda:0")
shapes = [(128, 197, 12, 32), (128, 197, 6, 64)] if len(sys.argv) < 2 else [tuple(int(v) for v in sys.argv[1].split(","))]
for B, N, h, dh in shapes:
  print(f"B={B} N={N} heads={h} dh={dh}", flush=True)
  for dt in (torch.float16, torch.float32):
      C = h * dh
      qkv = torch.randn(B * N, 3 * C, device=dev).to(dt)
      o = torch.empty(B * N, C, dtype=dt, device=dev); lse = torch.empty(B, h, N, device=dev)
      do = torch.randn(B * N, C, device=dev).to(dt); dqkv = torch.empty_like(qkv)
      def t(fn, fl, name):
          for _ in range(3): fn()
          torch.cuda.synchronize()
          s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          s.record()
          for _ in range(20): fn()
          e.record(); torch.cuda.synchronize()
          us = s.elapsed_time(e) * 1e3 / 20
          print(f"{str(dt):14s} {name:10s} {us:8.1f} us {fl/us/1e6:7.1f} TF", flush=True)
      t(lambda: ops.attention_fwd(qkv, B, N, h, dh, o, lse), 4.0 * B * h * N * N * dh, "fwd")
      t(lambda: ops.attention_bwd(qkv, o, do, lse, B, N, h, dh, dqkv), 10.0 * B * h * N * N * dh, "bwd")
