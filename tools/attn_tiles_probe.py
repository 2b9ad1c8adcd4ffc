#!/usr/bin/env python3
"""What do the padded key tiles / query rows of N = 197 cost the short-sequence attention kernels?  Same (B, heads, dh),
sequence lengths around the tile boundaries: 192 = 12 tiles (backward: 3 per wave), 197 = 13 of 16 (4 per wave, three of
the waves carry a dead tile), 208 = 13 full tiles, 224 = 14, 256 = 16 (no padding at all).
    python tools/attn_tiles_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3vit_amd import ops  # noqa: E402

B, heads, dh = 128, 12, 32
C = heads * dh
for N in (176, 192, 197, 208, 224, 256):
    qkv = (torch.randn(B * N, 3 * C, device="cuda") * 0.5).half()
    o = torch.empty(B * N, C, dtype=torch.float16, device="cuda")
    lse = torch.empty(B, heads, N, device="cuda")
    do = torch.randn(B * N, C, device="cuda").half()
    dqkv = torch.empty_like(qkv)
    ops.attention_fwd(qkv, B, N, heads, dh, o, lse)
    ops.attention_bwd(qkv, o, do, lse, B, N, heads, dh, dqkv)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    n = 30
    ev[0].record()
    for _ in range(n):
        ops.attention_fwd(qkv, B, N, heads, dh, o, lse)
    ev[1].record()
    for _ in range(n):
        ops.attention_bwd(qkv, o, do, lse, B, N, heads, dh, dqkv)
    ev[2].record()
    torch.cuda.synchronize()
    f, b = ev[0].elapsed_time(ev[1]) / n * 1e3, ev[1].elapsed_time(ev[2]) / n * 1e3
    print(f"N = {N:3d} ({(N + 15) // 16:2d} key tiles): forward {f:6.1f} us ({f / N / N * 1e3:.3f} ns per query x key)   "
          f"backward {b:6.1f} us ({b / N / N * 1e3:.3f})", flush=True)
